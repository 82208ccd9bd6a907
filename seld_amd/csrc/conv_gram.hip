// conv_gram.hip — the first conv block's kernel/bias gradient WITHOUT the pre-BN activations z.
//
// With P[px][k] the zero-padded im2col row of pixel px (k = (kh,kw,ci), plus a constant-1 column k = K for the
// bias), z = P W + b is linear in P, and the gradient that reaches z through BatchNorm(batch statistics) + ReLU +
// MaxPool is   dz[px][co] = ka[co] z[px][co] + kb[co] + hit(px,co) * scale[co] * dp[window(px)][co]
// (ka, kb from the two BN sums; hit = px is the recorded argmax of its window and the pooled value is > 0).  So
//     dW[k][co] = sum_px P[px][k] dz[px][co] = ka[co] * (G W + g (x) b)[k][co] + g[k] kb[co] + M[k][co]
//     G = P^T P  (Gram matrix of the input patches, (K+1)^2, depends on the INPUT only),  g[k] = sum_px P[px][k] = G[k][K]
//     M[k][co] = sum over windows w with p(w,co) > 0 of  scale[co] dp(w,co) P[argmax px of (w,co)][k]
// and z (1.57 GB at B=32, T=3000, written once and read once) is needed by nobody.
//   conv_first_gram_kernel   G on the fp32 MFMA: the wgrad kernel with the patch on BOTH operand sides (A and B
//                            fragments of a k-tile are the same registers), upper 32x32 tiles only.  Depends on x
//                            alone: it runs on the side stream under the GRU recurrence (which uses 64 of 256 CUs).
//   conv_first_msparse_kernel  M: one wave per pooling window, lane = output channel; each lane gathers the 63 patch
//                            values of ITS argmax pixel from the window's x rows in LDS (20 window positions ->
//                            20 distinct banks with a padded row stride) and accumulates 63 (+1 bias) sums in
//                            registers: 1/20 of the dense MACs, on the VALU.
//   conv_first_assemble_kernel  the 64 x 64 combination above, in double.
#include "common.h"

template <int CIN>
struct GramGeom {
    static constexpr int K = 9 * CIN;
    static constexpr int KP = 64 * ((K + 1 + 63) / 64);      // padded (K + ones column): 64 (CIN 7) or 128 (CIN 10)
    static constexpr int NKT = KP / 32;
    static constexpr int ROWF = 66 * CIN;
    static constexpr int PATCH = 6 * ROWF;
    static constexpr int NV = 64 * CIN / 4, SLOTS = 6 * NV, PER = (SLOTS + 255) / 256;
    static constexpr int NTILE = NKT * (NKT + 1) / 2;         // upper-triangular 32x32 tiles
};
int conv_gram_dim(int Cin) { return Cin == 7 ? 64 : 128; }
#define GRAM_MAX_BLOCKS 512
int conv_gram_slab_capacity() { return GRAM_MAX_BLOCKS; }

// G slab per block: [KP][KP] (only the upper tiles are written; the assembler mirrors them)
template <int CIN>
__global__ __launch_bounds__(256, 2) void conv_first_gram_kernel(const float* __restrict__ x, float* __restrict__ slab, int B, int H, int tile0, int tile1) {
    using G = GramGeom<CIN>;
    constexpr int K = G::K, KP = G::KP, NKT = G::NKT, ROWF = G::ROWF, NV = G::NV, SLOTS = G::SLOTS, PER = G::PER;
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float* patch = gsm;                   // [2][PATCH + 4]: double-buffered halo patch of a 4-row tile (+ the constants 1, 0)
    float* red = gsm + 2 * (G::PATCH + 4);   // [4][1024] combine buffer (used at the end)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, li = lane & 31;
    // operand of k-tile kt at k-step s (pixel f = 2s + hi of this wave's image row): patch[pa + s * sa]
    int pa[NKT], sa[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int k = kt * 32 + li;
        const int kh = k / (3 * CIN);
        const bool real = k < K;
        pa[kt] = real ? (wave + kh) * ROWF + (k - kh * 3 * CIN) + hi * CIN : (k == K ? G::PATCH : G::PATCH + 1);
        sa[kt] = real ? 2 * CIN : 0;
    }
    for (int i = tid; i < 2 * (G::PATCH + 4); i += 256) patch[i] = 0.f;
    __syncthreads();
    if (tid < 2) patch[tid * (G::PATCH + 4) + G::PATCH] = 1.f;       // the ones column (k = K); PATCH + 1 stays 0 (k > K)
    const int tiles_per_img = (H + 3) >> 2;
    const int ntiles = tile1;            // this launch covers tiles [tile0, tile1) of the B * tiles_per_img (a background launch may come in two parts)
    float4 stg[PER];
#define GR_ISSUE(tile_)                                                                                 \
    {                                                                                                   \
        const int ib_ = (tile_) / tiles_per_img, it0_ = ((tile_) - ib_ * tiles_per_img) * 4;            \
        _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                               \
            const int idx = tid + 256 * u;                                                              \
            const int r = idx / NV, c4 = idx - r * NV;                                                  \
            const int t = it0_ - 1 + r;                                                                 \
            const bool ok = idx < SLOTS && t >= 0 && t < H;                                             \
            const u32x4g v = *reinterpret_cast<const u32x4g*>(ok ? x + ((size_t)(ib_ * H + t) * 64 * CIN + 4 * c4) : x) & (ok ? 0xffffffffu : 0u); \
            stg[u] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); \
        }                                                                                               \
    }
#define GR_COMMIT(dst_)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                                   \
        const int idx = tid + 256 * u;                                                                  \
        if (idx < SLOTS) {                                                                              \
            const int r = idx / NV, c4 = idx - r * NV;                                                  \
            float* d = (dst_) + r * ROWF + CIN + 4 * c4;                                                \
            d[0] = stg[u].x; d[1] = stg[u].y; d[2] = stg[u].z; d[3] = stg[u].w;                         \
        }                                                                                               \
    }
    typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
    int tile = tile0 + blockIdx.x, cur = 0;
    if (tile < ntiles) {
        GR_ISSUE(tile)
        GR_COMMIT(patch)
    }
    __syncthreads();
    f32x16 acc[G::NTILE];
#pragma unroll
    for (int i = 0; i < G::NTILE; ++i) acc[i] = zero16();
    for (; tile < ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        GR_ISSUE(nxt < ntiles ? nxt : tile)
        __builtin_amdgcn_sched_barrier(0);
        const float* pt = patch + cur * (G::PATCH + 4);
        const int t = (tile - (tile / tiles_per_img) * tiles_per_img) * 4 + wave;     // this wave's image row
        if (t < H) {          // rows past H contribute nothing (their patch row is not even valid)
            float a[NKT];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) a[kt] = pt[pa[kt]];
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                float na[NKT];
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) na[kt] = (s + 1 < 32) ? pt[pa[kt] + (s + 1) * sa[kt]] : 0.f;
                __builtin_amdgcn_sched_barrier(0);
                int ti = 0;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int ct = kt; ct < NKT; ++ct, ++ti) acc[ti] = MFMA_F32_32x32x2(a[kt], a[ct], acc[ti]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) a[kt] = na[kt];
            }
        }
        GR_COMMIT(patch + (cur ^ 1) * (G::PATCH + 4))
        lds_barrier();
        cur ^= 1;
    }
#undef GR_ISSUE
#undef GR_COMMIT
    // combine the 4 waves' partial tiles (fixed order), one 32x32 tile at a time
    float* out = slab + (size_t)blockIdx.x * KP * KP;
    int ti = 0;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ct = kt; ct < NKT; ++ct, ++ti) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + mfma_row(r, hi) * 32 + li] = acc[ti][r];
            __syncthreads();
            for (int i = tid; i < 1024; i += 256) {
                const float v = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
                out[(size_t)(kt * 32 + (i >> 5)) * KP + ct * 32 + (i & 31)] = v;
            }
        }
}

// `background`: the launch shares the GPU with a latency-critical kernel on another stream (the GRU recurrence: one
// 512-thread block with 50 KB of LDS on 2B of the CUs).  Co-resident Gram blocks cost that kernel ~30 % (measured), so
// a background launch asks for 112 KB of LDS per block — it cannot land on a CU that runs a recurrence block — and for
// at most 128 blocks (swept 64..192: 128 disturbs the input-projection GEMMs and the second GRU layer least while still
// finishing well before the first block's backward needs G, ~2 ms later).
// Round 4: the background launch comes in two halves (api.hip, option "gram_parts" = 2), one under each of the first two GRU layers' forward
// recurrences, 192 blocks each: a half is done (~0.23 ms) before its recurrence is (0.29 ms), so the Gram product no longer shares the card with the second
// layer's input projection, the heads and the losses (the heads' product took 44 us beside it, ~15 alone).  Same box: 2.652 / 2.645 ms per step with one
// 128-block launch, 2.673 / 2.657 with two of 128, 2.631 / 2.633 with two of 192, 2.643 with two of 160.
int g_gram_bg_blocks = 192;   // workgroups of a background launch
// part / nparts: the launch covers the part-th of nparts equal shares of the tiles and writes its slabs behind those of the parts before it
// (*n_slab = this part's slab count; the caller sums them)
int launch_conv_first_gram(hipStream_t st, const float* x, float* slab, int* n_slab, int B, int H, int Cin, int background, int part, int nparts) {
    if ((Cin != 7 && Cin != 10) || B <= 0 || H <= 0 || nparts < 1 || part < 0 || part >= nparts) return -2;
    const int all = B * ((H + 3) / 4);
    const int tile0 = (int)((int64_t)all * part / nparts), tile1 = (int)((int64_t)all * (part + 1) / nparts);
    const int ntiles = tile1 - tile0;
    const int cap = background ? g_gram_bg_blocks : GRAM_MAX_BLOCKS;
    const int grid = ntiles < cap ? ntiles : cap;
    if (grid < 1) { *n_slab = 0; return 0; }
    const size_t lds_floor = background ? (size_t)112 * 1024 : 0;
    if (Cin == 7) {
        size_t smem = (size_t)(2 * (GramGeom<7>::PATCH + 4) + 4096) * sizeof(float);
        if (smem < lds_floor) smem = lds_floor;
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_gram_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(conv_first_gram_kernel<7>, dim3(grid), dim3(256), smem, st, x, slab, B, H, tile0, tile1);
    } else {
        size_t smem = (size_t)(2 * (GramGeom<10>::PATCH + 4) + 4096) * sizeof(float);
        if (smem < lds_floor) smem = lds_floor;
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_gram_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(conv_first_gram_kernel<10>, dim3(grid), dim3(256), smem, st, x, slab, B, H, tile0, tile1);
    }
    *n_slab = grid;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// M[k][co] = sum over pooling windows w of [p(w,co) > 0] scale[co] dp(w,co) P[argmax pixel of (w,co)][k]   (k = K: bias)
// Work unit = one pooled row (b, tp): image rows 5tp-1 .. 5tp+5 of x in LDS (row stride padded so that the 20 window
// positions fall into 20 different banks), 16 windows x 64 channels; wave w takes windows w, w+4, w+8, w+12.
template <int CIN>
struct MsGeom {
    static constexpr int K = 9 * CIN, K3 = 3 * CIN;
    static constexpr int RS = CIN == 7 ? 477 : 681;          // 66*CIN padded: RS % 64 = 29 (41) -> wr*RS + wc*CIN distinct mod 64
    static constexpr int ROWS = 7, REG = ROWS * RS;
    static constexpr int NV = 64 * CIN / 4, SLOTS = ROWS * NV, PER = (SLOTS + 255) / 256;
    static constexpr int KP = CIN == 7 ? 64 : 128;
};
#define MS_MAX_BLOCKS 1024
int conv_msparse_slab_capacity() { return MS_MAX_BLOCKS; }

template <int CIN>
__global__ __launch_bounds__(256, 2) void conv_first_msparse_kernel(const float* __restrict__ x, const float* __restrict__ p,
                                                                 const float* __restrict__ dp,
                                                                 const unsigned char* __restrict__ amax,
                                                                 const float* __restrict__ scale, float* __restrict__ slab, int B,
                                                                 int H) {
    using G = MsGeom<CIN>;
    constexpr int K = G::K, K3 = G::K3, RS = G::RS, NV = G::NV, SLOTS = G::SLOTS, PER = G::PER, KP = G::KP;
    extern __shared__ __attribute__((aligned(16))) float msm[];
    float* reg0 = msm;                    // [2][REG]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HP = H / 5;
    const int nrows = B * HP;             // pooled rows
    const float sc = scale[lane];
    float m[K + 1];
#pragma unroll
    for (int k = 0; k <= K; ++k) m[k] = 0.f;
    for (int i = tid; i < 2 * G::REG; i += 256) reg0[i] = 0.f;      // halo columns stay zero
    typedef unsigned u32x4m __attribute__((ext_vector_type(4)));
    float4 stg[PER];
#define MS_ISSUE(row_)                                                                                  \
    {                                                                                                   \
        const int ib_ = (row_) / HP, itp_ = (row_) - ib_ * HP;                                          \
        _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                               \
            const int idx = tid + 256 * u;                                                              \
            const int r = idx / NV, c4 = idx - r * NV;                                                  \
            const int t = 5 * itp_ - 1 + r;                                                             \
            const bool ok = idx < SLOTS && t >= 0 && t < H;                                             \
            const u32x4m v = *reinterpret_cast<const u32x4m*>(ok ? x + ((size_t)(ib_ * H + t) * 64 * CIN + 4 * c4) : x) & (ok ? 0xffffffffu : 0u); \
            stg[u] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); \
        }                                                                                               \
    }
#define MS_COMMIT(dst_)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < PER; ++u) {                                                   \
        const int idx = tid + 256 * u;                                                                  \
        if (idx < SLOTS) {                                                                              \
            const int r = idx / NV, c4 = idx - r * NV;                                                  \
            float* d = (dst_) + r * RS + CIN + 4 * c4;                                                  \
            d[0] = stg[u].x; d[1] = stg[u].y; d[2] = stg[u].z; d[3] = stg[u].w;                         \
        }                                                                                               \
    }
    int row = blockIdx.x, cur = 0;
    __syncthreads();
    if (row < nrows) {
        MS_ISSUE(row)
        MS_COMMIT(reg0)
    }
    __syncthreads();
    for (; row < nrows; row += gridDim.x) {
        const int nxt = row + gridDim.x;
        MS_ISSUE(nxt < nrows ? nxt : row)
        const float* rg = reg0 + cur * G::REG;
        const size_t wbase = (size_t)row * 16 * 64 + lane;      // (b*HP + tp)*16 windows * 64 channels
        // window j+1's pooled operands are loaded while window j's 63 gathers run (rolled loop: with the four windows
        // unrolled the register allocator kept the 64 accumulators in scratch memory)
        float pn = p[wbase + (size_t)wave * 64], dn = dp[wbase + (size_t)wave * 64];
        int an = amax[wbase + (size_t)wave * 64];
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int fp = wave + 4 * j;
            const float v = pn > 0.f ? sc * dn : 0.f;
            const float* q0 = rg + (an >> 2) * RS + (4 * fp + (an & 3)) * CIN;
            if (j < 3) {
                const size_t o = wbase + (size_t)(fp + 4) * 64;
                pn = p[o]; dn = dp[o]; an = amax[o];
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int i = 0; i < K3; ++i) m[kh * K3 + i] = fmaf(v, q0[kh * RS + i], m[kh * K3 + i]);
            m[K] += v;
        }
        MS_COMMIT(reg0 + (cur ^ 1) * G::REG)
        lds_barrier();
        cur ^= 1;
    }
#undef MS_ISSUE
#undef MS_COMMIT
    // combine the 4 waves (fixed order) through LDS, 16 rows of k at a time ([4][16][64] floats = 16 KB)
    float* red = msm;
    float* out = slab + (size_t)blockIdx.x * KP * 64;
#pragma unroll
    for (int k0 = 0; k0 <= K; k0 += 16) {
        __syncthreads();
#pragma unroll
        for (int k = k0; k < k0 + 16; ++k)
            if (k <= K) red[(wave * 16 + (k - k0)) * 64 + lane] = m[k];
        __syncthreads();
        for (int i = tid; i < 16 * 64; i += 256) {
            const int k = k0 + (i >> 6);
            if (k <= K) out[k * 64 + (i & 63)] = (red[i] + red[1024 + i]) + (red[2048 + i] + red[3072 + i]);
        }
    }
    for (int i = tid + (K + 1) * 64; i < KP * 64; i += 256) out[i] = 0.f;
}

int launch_conv_first_msparse(hipStream_t st, const float* x, const float* p, const float* dp, const unsigned char* amax,
                              const float* scale, float* slab, int* n_slab, int B, int H, int Cin) {
    if ((Cin != 7 && Cin != 10) || H % 5 || B <= 0) return -2;
    const int nrows = B * (H / 5);
    const int grid = nrows < MS_MAX_BLOCKS ? nrows : MS_MAX_BLOCKS;
    if (Cin == 7) {
        const size_t smem = (size_t)2 * MsGeom<7>::REG * sizeof(float);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_msparse_kernel<7>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(conv_first_msparse_kernel<7>, dim3(grid), dim3(256), smem, st, x, p, dp, amax, scale, slab, B, H);
    } else {
        const size_t smem = (size_t)2 * MsGeom<10>::REG * sizeof(float);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_msparse_kernel<10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(conv_first_msparse_kernel<10>, dim3(grid), dim3(256), smem, st, x, p, dp, amax, scale, slab, B, H);
    }
    *n_slab = grid;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// out[k][co] = ka[co] * ( sum_{k' < K} Gs[k][k'] W[k'][co] + Gs[k][K] b[co] ) + Gs[k][K] kb[co] + M[k][co],   k = 0..K
// (row K = the bias gradient: Gs[K][k'] = g[k'], Gs[K][K] = number of pixels).  Gs = the symmetric Gram matrix of which
// `G` holds the upper 32x32 tiles; coef = [mean | invstd | scale | shift | c1 | c2] x 64; all in double.
__global__ __launch_bounds__(256) void conv_first_assemble_kernel(const float* __restrict__ G, int KP, const float* __restrict__ M,
                                                                  const float* __restrict__ W, const float* __restrict__ bias,
                                                                  const float* __restrict__ coef, float* __restrict__ dW,
                                                                  float* __restrict__ db, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (K + 1) * 64) return;
    const int k = i >> 6, co = i & 63;
    const double mu = coef[co], is = coef[64 + co], sc = coef[128 + co], c1 = coef[256 + co], c2 = coef[320 + co];
    const double ka = -sc * c2 * is, kb = -sc * c1 - ka * mu;
    auto gs = [&](int r, int c) -> double { return (r >> 5) <= (c >> 5) ? G[(size_t)r * KP + c] : G[(size_t)c * KP + r]; };
    // four interleaved partial sums (fixed order): the 63 loads of a row no longer wait on one dependent FMA chain
    double t0 = gs(k, K) * (double)bias[co], t1 = 0.0, t2 = 0.0, t3 = 0.0;
    int k2 = 0;
#pragma unroll 2
    for (; k2 + 3 < K; k2 += 4) {
        t0 += gs(k, k2) * (double)W[k2 * 64 + co];
        t1 += gs(k, k2 + 1) * (double)W[(k2 + 1) * 64 + co];
        t2 += gs(k, k2 + 2) * (double)W[(k2 + 2) * 64 + co];
        t3 += gs(k, k2 + 3) * (double)W[(k2 + 3) * 64 + co];
    }
    for (; k2 < K; ++k2) t0 += gs(k, k2) * (double)W[k2 * 64 + co];
    const double t = (t0 + t1) + (t2 + t3);
    const float v = (float)(ka * t + gs(k, K) * kb + (double)M[k * 64 + co]);
    if (k < K) dW[k * 64 + co] = v;
    else db[co] = v;
}

int launch_conv_first_assemble(hipStream_t st, const float* G, const float* M, const float* W, const float* bias, const float* coef,
                               float* dW, float* db, int Cin) {
    if (Cin != 7 && Cin != 10) return -2;
    const int K = 9 * Cin, KP = conv_gram_dim(Cin);
    hipLaunchKernelGGL(conv_first_assemble_kernel, dim3(((K + 1) * 64 + 255) / 256), dim3(256), 0, st, G, KP, M, W, bias, coef, dW, db, K);
    return 0;
}
