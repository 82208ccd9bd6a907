// conv_pool.hip — first-layer Conv2D (layers.py:27-32, Cin = 7 or 10) whose epilogue also reduces each
// (5,4) max-pool window of the PRE-normalisation output, so MaxPooling2D(BN+ReLU(z)) never re-reads z.
//
// BatchNorm + ReLU is y = max(0, fmaf(z, scale, shift)) with scale = gamma * invstd, and a correctly
// rounded fma is monotone in z, so over a pooling window
//     max_w y(z_w) = y(max_w z_w)   if gamma >= 0          max_w y(z_w) = y(min_w z_w)   if gamma < 0
// bit for bit.  gamma is known before the batch statistics are, so the convolution writes one extreme
// value per window and channel (zext, 1/20 of z) next to the statistics partials, and after
// bn_finalize the pooled activation is the elementwise bn_relu_ext over zext (79 MB instead of 1.57 GB).
// z itself is still written in training (the backward pass reads it) and skipped in inference.
//
// Tile = 10 image rows x 64 frequency bins = two pooling rows.  Wave w owns pooling row w>>1 and the
// 32-bin strip w&1: 5 MFMA row tiles x 2 channel halves = 10 independent accumulator chains that share
// the two weight operands of a k-step (7 LDS reads per 10 MFMAs).  In the 32x32 accumulator layout a
// lane holds bins 4*hi + 8*q + (0..3), i.e. whole pooling windows, so the (5,4) reduction is register
// only.
#include "common.h"

#define CPOOL_MAX_PERSISTENT 512
int conv_pool_stat_capacity() { return CPOOL_MAX_PERSISTENT; }

template <int CIN>
struct PoolGeom {
    static constexpr int K = 9 * CIN;
    static constexpr int KPAD = (K + 2) & ~1;      // + bias row, even
    static constexpr int ROWF = 66 * CIN;          // floats per patch row (64 bins + 2 halo columns)
    static constexpr int ROWS = 12;                // 10 image rows + 2 halo rows
    static constexpr int PATCH = ROWS * ROWF;
    static constexpr int NV = 64 * CIN / 4;        // float4 per image row
    static constexpr int SLOTS = ROWS * NV;
    static constexpr int PER_THREAD = (SLOTS + 255) / 256;
    static constexpr size_t SMEM = (size_t)(KPAD * 64 + 2 * PATCH + 512) * sizeof(float);
};

template <int CIN, bool WRITE_Z, bool WRITE_AMAX>
__global__ __launch_bounds__(256, 2) void conv_first_fwd_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias,
                                                                     const float* __restrict__ gamma, float* __restrict__ z,
                                                                     float* __restrict__ zext,
                                                                     unsigned char* __restrict__ amax,
                                                                     float* __restrict__ stat_partial, int B, int H) {
    using G = PoolGeom<CIN>;
    constexpr int K = G::K, KPAD = G::KPAD, ROWF = G::ROWF, NS = KPAD / 2, NV = G::NV, SLOTS = G::SLOTS, PER = G::PER_THREAD;
    extern __shared__ __attribute__((aligned(16))) float cp_smem[];
    float* Wl = cp_smem;                       // [KPAD][64]
    float* patch0 = Wl + KPAD * 64;            // [2][PATCH]
    float* red = patch0 + 2 * G::PATCH;        // [4][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // uniform: row/strip offsets stay scalar
    const int hi = lane >> 5, li = lane & 31;
    const int grp = wave >> 1, strip = wave & 1;
    for (int idx = tid; idx < KPAD * 64; idx += 256) {
        const int k = idx >> 6, co = idx & 63;
        Wl[idx] = (k < K) ? w[idx] : (k == K ? (bias ? bias[co] : 0.f) : 0.f);
    }
    for (int idx = tid; idx < 2 * G::PATCH; idx += 256) patch0[idx] = 0.f;   // halo columns stay zero
    // gamma < 0: the window extreme that survives BN+ReLU+MaxPool is the minimum of z -> flip the sign, take the maximum
    const unsigned smask[2] = {gamma[li] < 0.f ? 0x80000000u : 0u, gamma[32 + li] < 0.f ? 0x80000000u : 0u};
    const int tiles_per_img = (H + 9) / 10;
    const int ntiles = B * tiles_per_img;
    const int HP = H / 5;
    float4 stg[PER];
    // Loads are unconditional (clamped pointer + select) so that no branch separates them, and the commit
    // always runs (a conditional commit lets the compiler sink the loads behind the condition).
#define CP_ISSUE(tile_)                                                                                 \
    {                                                                                                   \
        const int ib_ = (tile_) / tiles_per_img, it0_ = ((tile_) - ib_ * tiles_per_img) * 10;           \
        _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                               \
            const int idx = tid + 256 * u;                                                              \
            const int r = idx / NV, c4 = idx - r * NV;                                                  \
            const int t = it0_ - 1 + r;                                                                 \
            const bool ok = idx < SLOTS && t >= 0 && t < H;                                             \
            const float4 v = *reinterpret_cast<const float4*>(ok ? x + ((size_t)(ib_ * H + t) * 64 * CIN + 4 * c4) : x); \
            stg[u] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);                                          \
        }                                                                                               \
    }
#define CP_COMMIT(dst_)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                                   \
        const int idx = tid + 256 * u;                                                                  \
        if (idx < SLOTS) {                                                                              \
            const int r = idx / NV, c4 = idx - r * NV;                                                  \
            float* d = (dst_) + r * ROWF + CIN + 4 * c4;      /* column 1: not 16-B aligned */          \
            d[0] = stg[u].x; d[1] = stg[u].y; d[2] = stg[u].z; d[3] = stg[u].w;                         \
        }                                                                                               \
    }
    int tile = blockIdx.x;
    __syncthreads();
    if (tile < ntiles) {
        CP_ISSUE(tile)
        CP_COMMIT(patch0)
    }
    __syncthreads();
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int base = ((5 * grp) * 66 + 32 * strip + li) * CIN;     // pixel (row 5*grp, bin 32*strip+li) at tap (0,0)
    const float* wlh = Wl + hi * 64 + li;
    const int lane_z = (4 * hi) * 64 + li, lane_e = hi * 64 + li;
    int cur = 0;
#define CP_F(k_) (((k_) / (3 * CIN)) * ROWF + ((k_) % (3 * CIN)))
#define CP_A(s_, j_) ((CP_F(2 * (s_) + 1) - CP_F(2 * (s_)) == 1) ? pth[CP_F(2 * (s_)) + (j_) * ROWF]                   \
                                                                 : pt[CP_F(2 * (s_)) + hi * (CP_F(2 * (s_) + 1) - CP_F(2 * (s_))) + (j_) * ROWF])
    // operands of k-step s for channel half c_: k = 2s + hi.  The patch offset of k is f(k) = (k / 3CIN) * ROWF +
    // k % 3CIN, written as f(2s) + hi * (f(2s+1) - f(2s)) with compile-time f so that every LDS read is one base
    // register (pt + hi, or pt when the pair straddles a kernel row) plus an immediate offset.  The last
    // step(s) carry the bias row (A = 1) and zero padding.
#define CP_LD(s_, c_, A0, A1, A2, A3, A4, B0)                                                           \
    {                                                                                                   \
        if (2 * (s_) + 1 < K) {                                                                         \
            A0 = CP_A(s_, 0); A1 = CP_A(s_, 1); A2 = CP_A(s_, 2); A3 = CP_A(s_, 3); A4 = CP_A(s_, 4);   \
        } else if (2 * (s_) < K) {         /* hi = 0: last real row; hi = 1: k = K, the bias row */     \
            A0 = hi ? 1.f : pt[CP_F(2 * (s_))]; A1 = hi ? 1.f : pt[CP_F(2 * (s_)) + ROWF];              \
            A2 = hi ? 1.f : pt[CP_F(2 * (s_)) + 2 * ROWF]; A3 = hi ? 1.f : pt[CP_F(2 * (s_)) + 3 * ROWF]; \
            A4 = hi ? 1.f : pt[CP_F(2 * (s_)) + 4 * ROWF];                                              \
        } else {                                                                                        \
            A0 = A1 = A2 = A3 = A4 = (2 * (s_) + hi == K) ? 1.f : 0.f;                                  \
        }                                                                                               \
        B0 = wlh[128 * (s_) + 32 * (c_)];                                                               \
    }
    // drain accumulator register r_ of one channel half (Y0..Y4 = the 5 image rows): 5 z stores, statistics,
    // running window extreme (and its position); the window's zext / amax go out with its fourth register
#define CP_DRAIN(r_, c_, Y0, Y1, Y2, Y3, Y4)                                                            \
    {                                                                                                   \
        const float v0 = Y0[r_], v1 = Y1[r_], v2 = Y2[r_], v3 = Y3[r_], v4 = Y4[r_];                    \
        if (WRITE_Z) {                                                                                  \
            float* zp = z + (zr + (size_t)((8 * ((r_) >> 2) + ((r_) & 3)) * 64 + 32 * (c_)));          \
            zp[lane_z] = v0; zp[lane_z + 4096] = v1; zp[lane_z + 2 * 4096] = v2;                        \
            zp[lane_z + 3 * 4096] = v3; zp[lane_z + 4 * 4096] = v4;                                     \
        }                                                                                               \
        s1[c_] += (v0 + v1) + (v2 + v3) + v4;                                                           \
        s2[c_] = fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, fmaf(v3, v3, fmaf(v4, v4, s2[c_])))));          \
        const float w0 = __uint_as_float(__float_as_uint(v0) ^ smask[c_]), w1 = __uint_as_float(__float_as_uint(v1) ^ smask[c_]); \
        const float w2 = __uint_as_float(__float_as_uint(v2) ^ smask[c_]), w3 = __uint_as_float(__float_as_uint(v3) ^ smask[c_]); \
        const float w4 = __uint_as_float(__float_as_uint(v4) ^ smask[c_]);                              \
        const float hi5 = fmaxf(fmaxf(fmaxf(w0, w1), fmaxf(w2, w3)), w4);                               \
        const bool take_ = ((r_) & 3) == 0 || hi5 > best;     /* strict: the first extreme in scan order wins ties */ \
        if (WRITE_AMAX) {    /* training: remember WHERE the extreme is: position row * 4 + column of the window */ \
            int row_ = 4;                                                                               \
            row_ = (w3 == hi5) ? 3 : row_; row_ = (w2 == hi5) ? 2 : row_;                               \
            row_ = (w1 == hi5) ? 1 : row_; row_ = (w0 == hi5) ? 0 : row_;                               \
            bpos = take_ ? row_ * 4 + ((r_) & 3) : bpos;                                                \
        }                                                                                               \
        best = take_ ? hi5 : best;                                                                      \
        if (((r_) & 3) == 3) {                                                                          \
            (zext + (er + (size_t)((2 * ((r_) >> 2)) * 64 + 32 * (c_))))[lane_e] = __uint_as_float(__float_as_uint(best) ^ smask[c_]); \
            if (WRITE_AMAX) (amax + (er + (size_t)((2 * ((r_) >> 2)) * 64 + 32 * (c_))))[lane_e] = (unsigned char)bpos; \
        }                                                                                               \
    }
    static_assert(NS >= 32, "the drain of half a tile is spread over 32 k-steps");
#ifdef CPOOL_TIMING
    long long tm_top = 0, tm_a = 0, tm_b = 0, tm_commit = 0, tm_epi = 0, tm_bar = 0;
#endif
    for (; tile < ntiles; tile += gridDim.x) {
#ifdef CPOOL_TIMING
        const long long c0 = clock64();
#endif
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * 10;
        const int nxt = tile + gridDim.x;
        CP_ISSUE(nxt < ntiles ? nxt : tile)
        __builtin_amdgcn_sched_barrier(0);
        const float* pt = patch0 + cur * G::PATCH + base;
        const float* pth = pt + hi;
        const int tg = t0 + 5 * grp;             // first image row of this wave's pooling row
        const bool live = tg < H;                // H % 5 == 0: a pooling row is entirely inside or outside
        const size_t zr = ((size_t)(b * H + tg) * 64 + 32 * strip) * 64;
        const size_t er = ((size_t)(b * HP + tg / 5) * 16 + 8 * strip) * 64;
        float best = 0.f;
        int bpos = 0;
#ifdef CPOOL_TIMING
        const long long c1 = clock64();
#endif
        f32x16 accA0 = zero16(), accA1 = zero16(), accA2 = zero16(), accA3 = zero16(), accA4 = zero16();
        f32x16 accB0 = zero16(), accB1 = zero16(), accB2 = zero16(), accB3 = zero16(), accB4 = zero16();
        if constexpr (!WRITE_Z) {
            // No z to store (inference, or training through the patch Gram matrix): nothing to spread, so ONE pass over k
            // with all 10 accumulator chains — 7 LDS reads per 10 MFMAs instead of 12 — and the window reduction after it.
            float a0, a1, a2, a3, a4, b0, b1;
            CP_LD(0, 0, a0, a1, a2, a3, a4, b0)
            b1 = wlh[32];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f, n4 = 0.f, m0 = 0.f, m1 = 0.f;
                if (s + 1 < NS) {
                    CP_LD(s + 1, 0, n0, n1, n2, n3, n4, m0)
                    m1 = wlh[128 * (s + 1) + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
                accA0 = MFMA_F32_32x32x2(a0, b0, accA0);
                accB0 = MFMA_F32_32x32x2(a0, b1, accB0);
                accA1 = MFMA_F32_32x32x2(a1, b0, accA1);
                accB1 = MFMA_F32_32x32x2(a1, b1, accB1);
                accA2 = MFMA_F32_32x32x2(a2, b0, accA2);
                accB2 = MFMA_F32_32x32x2(a2, b1, accB2);
                accA3 = MFMA_F32_32x32x2(a3, b0, accA3);
                accB3 = MFMA_F32_32x32x2(a3, b1, accB3);
                accA4 = MFMA_F32_32x32x2(a4, b0, accA4);
                accB4 = MFMA_F32_32x32x2(a4, b1, accB4);
                __builtin_amdgcn_sched_barrier(0);
                a0 = n0; a1 = n1; a2 = n2; a3 = n3; a4 = n4; b0 = m0; b1 = m1;
            }
            if (live) {
#pragma unroll
                for (int r = 0; r < 16; ++r) CP_DRAIN(r, 0, accA0, accA1, accA2, accA3, accA4)
            }
        } else {
        // ---- phase A: channels 0..31 of the wave's 5 row tiles
        {
            float a0, a1, a2, a3, a4, b0;
            CP_LD(0, 0, a0, a1, a2, a3, a4, b0)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f, n4 = 0.f, m0 = 0.f;
                if (s + 1 < NS) CP_LD(s + 1, 0, n0, n1, n2, n3, n4, m0)
                __builtin_amdgcn_sched_barrier(0);
                accA0 = MFMA_F32_32x32x2(a0, b0, accA0);
                accA1 = MFMA_F32_32x32x2(a1, b0, accA1);
                accA2 = MFMA_F32_32x32x2(a2, b0, accA2);
                accA3 = MFMA_F32_32x32x2(a3, b0, accA3);
                accA4 = MFMA_F32_32x32x2(a4, b0, accA4);
                __builtin_amdgcn_sched_barrier(0);
                a0 = n0; a1 = n1; a2 = n2; a3 = n3; a4 = n4; b0 = m0;
            }
        }
#ifdef CPOOL_TIMING
        const long long c2 = clock64();
#endif
        // ---- phase B: channels 32..63, and under its MFMAs the results of phase A trickle out: one accumulator
        // register (5 stores) every second k-step.  A wave can have 64 vector-memory operations in flight
        // (vmcnt); the 160 stores of a whole tile issued in one burst stalled on HBM write acknowledgements
        // for ~19k cycles per tile (measured with -DCPOOL_TIMING), longer than the other block's k-loop.
        {
            float a0, a1, a2, a3, a4, b0;
            CP_LD(0, 1, a0, a1, a2, a3, a4, b0)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f, n4 = 0.f, m0 = 0.f;
                if (s + 1 < NS) CP_LD(s + 1, 1, n0, n1, n2, n3, n4, m0)
                __builtin_amdgcn_sched_barrier(0);
                accB0 = MFMA_F32_32x32x2(a0, b0, accB0);
                accB1 = MFMA_F32_32x32x2(a1, b0, accB1);
                accB2 = MFMA_F32_32x32x2(a2, b0, accB2);
                accB3 = MFMA_F32_32x32x2(a3, b0, accB3);
                accB4 = MFMA_F32_32x32x2(a4, b0, accB4);
                if (s < 32 && (s & 1) == 0 && live) CP_DRAIN(s >> 1, 0, accA0, accA1, accA2, accA3, accA4)
                __builtin_amdgcn_sched_barrier(0);
                a0 = n0; a1 = n1; a2 = n2; a3 = n3; a4 = n4; b0 = m0;
            }
        }
        }
#ifdef CPOOL_TIMING
        const long long c3 = clock64();
#endif
        // Commit the prefetched patch BEFORE the remaining z stores: vmcnt counts loads and stores in one
        // in-order queue, so waiting for loads issued ahead of stores would wait for those stores too.
        CP_COMMIT(patch0 + (cur ^ 1) * G::PATCH)
        __builtin_amdgcn_sched_barrier(0);
#ifdef CPOOL_TIMING
        const long long c4 = clock64();
#endif
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) CP_DRAIN(r, 1, accB0, accB1, accB2, accB3, accB4)
        }
#ifdef CPOOL_TIMING
        const long long c5 = clock64();
#endif
        lds_barrier();
        cur ^= 1;
#ifdef CPOOL_TIMING
        const long long c6 = clock64();
        tm_top += c1 - c0; tm_a += c2 - c1; tm_b += c3 - c2; tm_commit += c4 - c3; tm_epi += c5 - c4; tm_bar += c6 - c5;
#endif
    }
#undef CP_DRAIN
#undef CP_LD
#undef CP_A
#undef CP_F
#undef CP_ISSUE
#undef CP_COMMIT
    if (stat_partial) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            s1[c] += __shfl_xor(s1[c], 32);
            s2[c] += __shfl_xor(s2[c], 32);
        }
        if (hi == 0) {
            red[wave * 128 + li] = s1[0];
            red[wave * 128 + 32 + li] = s1[1];
            red[wave * 128 + 64 + li] = s2[0];
            red[wave * 128 + 96 + li] = s2[1];
        }
        __syncthreads();
        if (tid < 128) stat_partial[(size_t)blockIdx.x * 128 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
#ifdef CPOOL_TIMING
        __syncthreads();
        if (tid == 0) {      // debug build: the first 6 statistics slots carry wave 0's phase cycles instead
            float* o = stat_partial + (size_t)blockIdx.x * 128;
            o[0] = (float)tm_top; o[1] = (float)tm_a; o[2] = (float)tm_b; o[3] = (float)tm_commit; o[4] = (float)tm_epi; o[5] = (float)tm_bar;
        }
#endif
    }
}

template <int CIN>
static int launch_cpool(hipStream_t st, const float* x, const float* w, const float* bias, const float* gamma, float* z,
                        float* zext, unsigned char* amax, float* stat_partial, int* n_partial, int B, int H) {
    using G = PoolGeom<CIN>;
    const int ntiles = B * ((H + 9) / 10);
    const int grid = ntiles < CPOOL_MAX_PERSISTENT ? ntiles : CPOOL_MAX_PERSISTENT;
#define CPOOL_GO(Z_, A_)                                                                                              \
    {                                                                                                                 \
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_fwd_pool_kernel<CIN, Z_, A_>),                   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM);                                \
        hipLaunchKernelGGL((conv_first_fwd_pool_kernel<CIN, Z_, A_>), dim3(grid), dim3(256), G::SMEM, st, x, w, bias, \
                           gamma, z, zext, amax, stat_partial, B, H);                                                 \
    }
    if (z && amax) CPOOL_GO(true, true)
    else if (amax) CPOOL_GO(false, true)        // training without z: the Gram-matrix backward (conv_gram.hip)
    else CPOOL_GO(false, false)                 // inference
#undef CPOOL_GO
    if (n_partial) *n_partial = grid;
    return 0;
}

// x [B,H,64,Cin] -> z [B,H,64,64] and amax [B,H/5,16,64] (bytes: position row*4+col of the window's extreme, the first
// in column-then-row scan order on ties) — both optional, stored together for the backward pass —, zext
// [B,H/5,16,64] (per (5,4) window the max of z where gamma >= 0, the min where gamma < 0), BN statistics partials
// [n_partial][128].
int launch_conv_first_fwd_pool(hipStream_t st, const float* x, const float* w, const float* bias, const float* gamma,
                               float* z, float* zext, unsigned char* amax, float* stat_partial, int* n_partial, int B, int H,
                               int Cin, int split_bf16) {
    if (H % 5 || H <= 0 || B <= 0 || (z != nullptr && amax == nullptr)) return -2;   // z is stored for a backward pass, which needs amax
    // without a z to store the layer runs on the bf16 matrix cores with exactly split operands (conv_pool_sb.hip)
    if (!z && split_bf16) return launch_conv_first_fwd_pool_sb(st, x, w, bias, gamma, zext, amax, stat_partial, n_partial, B, H, Cin);
    if (Cin == 7) return launch_cpool<7>(st, x, w, bias, gamma, z, zext, amax, stat_partial, n_partial, B, H);
    if (Cin == 10) return launch_cpool<10>(st, x, w, bias, gamma, z, zext, amax, stat_partial, n_partial, B, H);
    return -2;
}

// p = max(0, fmaf(zext, scale[c], shift[c])), channels innermost (64)
__global__ __launch_bounds__(256) void bn_relu_ext_kernel(const float* zext, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* p, int64_t n4) {   // p may alias zext
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int g = (int)(gid & 15);
    const float4 sc = reinterpret_cast<const float4*>(scale)[g];
    const float4 sh = reinterpret_cast<const float4*>(shift)[g];
    const float4 v = reinterpret_cast<const float4*>(zext)[gid];
    float4 o;
    o.x = fmaxf(0.f, fmaf(v.x, sc.x, sh.x));
    o.y = fmaxf(0.f, fmaf(v.y, sc.y, sh.y));
    o.z = fmaxf(0.f, fmaf(v.z, sc.z, sh.z));
    o.w = fmaxf(0.f, fmaf(v.w, sc.w, sh.w));
    reinterpret_cast<float4*>(p)[gid] = o;
}

int launch_bn_relu_ext(hipStream_t st, const float* zext, const float* scale, const float* shift, float* p, int64_t n) {
    if (n % 64) return -2;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(bn_relu_ext_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, zext, scale, shift, p, n4);
    return 0;
}

// Position (row * PF + col) of each pooling window's extreme of z — max where gamma >= 0, min where gamma < 0, the
// first in column-then-row scan order on ties — for forward paths that did not produce it (unfused first block,
// per-kernel test entry).  Same rule as the epilogue of conv_first_fwd_pool_kernel.
__global__ __launch_bounds__(256) void pool_argext_kernel(const float* __restrict__ z, const float* __restrict__ gamma,
                                                          unsigned char* __restrict__ amax, int64_t npool, int H, int W, int PT,
                                                          int PF) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npool * 64) return;
    const int c = (int)(gid & 63);
    const int64_t pp = gid >> 6;
    const int Wp = W / PF, Hp = H / PT;
    const int fp = (int)(pp % Wp);
    const int tp = (int)((pp / Wp) % Hp);
    const int b = (int)(pp / ((int64_t)Wp * Hp));
    const unsigned sm = gamma[c] < 0.f ? 0x80000000u : 0u;
    const float* base = z + (((size_t)b * H + (size_t)tp * PT) * W + (size_t)fp * PF) * 64 + c;
    float best = 0.f;
    int bpos = 0;
    for (int j = 0; j < PF; ++j)
        for (int i = 0; i < PT; ++i) {
            const float w = __uint_as_float(__float_as_uint(base[((size_t)i * W + j) * 64]) ^ sm);
            if ((i == 0 && j == 0) || w > best) { best = w; bpos = i * PF + j; }
        }
    amax[gid] = (unsigned char)bpos;
}

int launch_pool_argext(hipStream_t st, const float* z, const float* gamma, unsigned char* amax, int B, int H, int W, int pt,
                       int pf) {
    if (H % pt || W % pf || pt * pf > 255) return -2;
    const int64_t n = (int64_t)B * (H / pt) * (W / pf) * 64;
    hipLaunchKernelGGL(pool_argext_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, gamma, amax, n / 64, H, W, pt, pf);
    return 0;
}
