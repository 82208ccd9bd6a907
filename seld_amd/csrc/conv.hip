// conv.hip — Conv2D(64, 3x3, 'same', use_bias) of layers.conv2d_bn (layers.py:27-32) on NHWC fp32,
// forward + kernel/bias gradient + input gradient, as implicit GEMMs on v_mfma_f32_32x32x2_f32.
//
//   conv_first_fwd<CIN>    first layer (Cin = 7): x [B,H,64,CIN] -> z [B,H,64,64]; K = 9*CIN (+1 bias row),
//                          one halo patch [6][66][CIN] in LDS per 4x64-pixel tile, persistent blocks,
//                          BatchNorm batch statistics (sum z, sum z^2) from the accumulators.
//   conv64_fwd<STATS>      Cin = Cout = 64 layers: 128-pixel tiles, 9 taps x K=64 through LDS.
//                          The same kernel computes the input gradient with flipped/transposed weights.
//   conv_first_wgrad<CIN>  dW[k][co] = sum_p col(x)[p][k] dz[p][co]; bias gradient = the "ones" row k = K.
//   conv64_wgrad<WLOG2>    all 9 taps per block from one halo region; 9 accumulator tiles per wave.
// Block partials go to slabs that reduce_slabs_kernel (gemm.hip) combines in a fixed order.
#include "common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CONV_MAX_PERSISTENT 768
int conv_stat_partial_capacity() { return CONV_MAX_PERSISTENT; }
#define WGRAD_MAX_BLOCKS 512
int conv_wgrad_slab_capacity() { return WGRAD_MAX_BLOCKS; }

// ================================================================================================
// first layer forward
// ================================================================================================
template <int CIN>
struct FirstGeom {
    static constexpr int K = 9 * CIN;              // im2col depth
    static constexpr int KPAD = (K + 2) & ~1;      // + bias row, even
    static constexpr int ROWF = 66 * CIN;          // floats per patch row
    static constexpr int PATCH = 6 * ROWF;
    static constexpr int NKT = (KPAD + 31) / 32;   // 32-row k-tiles of the weight gradient
    static constexpr int WG_SLAB = NKT * 32 * 64;  // floats per wgrad slab: rows 0..K-1 kernel, row K bias
};
int conv_first_wgrad_slab_stride(int Cin) { return (((9 * Cin + 2) & ~1) + 31) / 32 * 32 * 64; }

// The halo patch of a 4x64-pixel tile: patch[r][c][ci] for image rows t0-1..t0+4, cols -1..64.
// Global side: each image row is 64*CIN contiguous floats -> float4 loads (PatchStage::NV per row).
// LDS side: data lands at column 1 (offset CIN floats, not 16-B aligned) -> b32 writes; the halo
// columns 0 and 65 are zeroed once per kernel and never overwritten.
template <int CIN>
struct PatchStage {
    static constexpr int NV = 64 * CIN / 4;        // float4 per image row
    static constexpr int SLOTS = 6 * NV;           // float4 per patch
    static constexpr int PER_THREAD = (SLOTS + 255) / 256;
    float4 v[PER_THREAD];
    __device__ __forceinline__ void issue(const float* __restrict__ x, int b, int t0, int H, int tid) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int idx = tid + 256 * u;
            const int r = idx / NV, c4 = idx - r * NV;
            const int t = t0 - 1 + r;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < SLOTS && t >= 0 && t < H)
                v[u] = reinterpret_cast<const float4*>(x + (size_t)(b * H + t) * 64 * CIN)[c4];
        }
    }
    __device__ __forceinline__ void commit(float* patch, int tid) const {
        constexpr int ROWF = 66 * CIN;
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int idx = tid + 256 * u;
            if (idx < SLOTS) {
                const int r = idx / NV, c4 = idx - r * NV;
                float* d = patch + r * ROWF + CIN + 4 * c4;
                d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
            }
        }
    }
};

template <int CIN>
__device__ __forceinline__ void zero_patch(float* patch, int tid) {
    for (int idx = tid; idx < FirstGeom<CIN>::PATCH; idx += 256) patch[idx] = 0.f;
}

// Pipeline per tile: [issue next tile's patch loads -> registers] [MFMA over the current LDS patch]
// [epilogue: z stores, BN partial sums in registers] [commit the prefetched patch to the other LDS
// buffer] [one LDS-only barrier].
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ z,
                                                             float* __restrict__ stat_partial, int B, int H) {
    using G = FirstGeom<CIN>;
    constexpr int K = G::K, KPAD = G::KPAD, ROWF = G::ROWF, NS = KPAD / 2;
    __shared__ __attribute__((aligned(16))) float Wl[KPAD * 64];
    __shared__ __attribute__((aligned(16))) float patch[2][G::PATCH];
    __shared__ float red[4 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    for (int idx = tid; idx < KPAD * 64; idx += 256) {
        const int k = idx >> 6, co = idx & 63;
        Wl[idx] = (k < K) ? w[idx] : (k == K ? (bias ? bias[co] : 0.f) : 0.f);
    }
    zero_patch<CIN>(patch[0], tid);
    zero_patch<CIN>(patch[1], tid);
    const int tiles_per_img = (H + 3) >> 2;
    const int ntiles = B * tiles_per_img;
    PatchStage<CIN> stg;
    int tile = blockIdx.x;
    __syncthreads();
    if (tile < ntiles) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * 4;
        stg.issue(x, b, t0, H, tid);
        stg.commit(patch[0], tid);
    }
    __syncthreads();
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};  // running BN sums of this lane's 2 channels
    const int base0 = (wave * 66 + li) * CIN;      // pixel (row = wave, f = li) at tap (0,0)
    const int base1 = base0 + 32 * CIN;
    int cur = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * 4;
        const int nxt = tile + gridDim.x;
        const bool has_next = nxt < ntiles;
        if (has_next) {
            const int nb = nxt / tiles_per_img;
            stg.issue(x, nb, (nxt - nb * tiles_per_img) * 4, H, tid);
        }
        const float* pt = patch[cur];
        f32x16 acc[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[p][c] = zero16();
        // operands of step s: k = 2s + hi
        auto ld = [&](int s, float& a0, float& a1, float& b0, float& b1) {
            const int k = 2 * s + hi;
            const int kh = k / (3 * CIN);
            const int off = kh * ROWF + (k - kh * 3 * CIN);
            if (2 * s + 1 < K) {
                a0 = pt[base0 + off];
                a1 = pt[base1 + off];
            } else {  // last step(s): the bias row (A = 1) and zero padding
                const bool real = k < K;
                const float one = (k == K) ? 1.f : 0.f;
                a0 = real ? pt[base0 + (real ? off : 0)] : one;
                a1 = real ? pt[base1 + (real ? off : 0)] : one;
            }
            b0 = Wl[k * 64 + li];
            b1 = Wl[k * 64 + 32 + li];
        };
        // LDS operand pipeline, pinned with sched_barrier: the reads of step s+1 are in flight while the 4
        // MFMAs of step s (256 pipe cycles) issue; left alone, hipcc sinks each read next to its use and
        // waits lgkmcnt(0) in front of every MFMA group (measured: 41% MFMA-pipe utilisation).
        float a0, a1, b0, b1;
        ld(0, a0, a1, b0, b1);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
            if (s + 1 < NS) ld(s + 1, na0, na1, nb0, nb1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = MFMA_F32_32x32x2(a0, b0, acc[0][0]);
            acc[0][1] = MFMA_F32_32x32x2(a0, b1, acc[0][1]);
            acc[1][0] = MFMA_F32_32x32x2(a1, b0, acc[1][0]);
            acc[1][1] = MFMA_F32_32x32x2(a1, b1, acc[1][1]);
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
        const int t = t0 + wave;
        if (t < H) {
            float* zr = z + (size_t)(b * H + t) * 64 * 64;
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[p][c][r];
                        zr[(p * 32 + mfma_row(r, hi)) * 64 + c * 32 + li] = v;
                        s1[c] += v;
                        s2[c] = fmaf(v, v, s2[c]);
                    }
        }
        if (has_next) stg.commit(patch[cur ^ 1], tid);
        lds_barrier();
        cur ^= 1;
    }
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
        if (tid < 128) stat_partial[(size_t)blockIdx.x * 128 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
    }
}

int launch_conv_first_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* z,
                          float* stat_partial, int* n_partial, int B, int H, int Cin) {
    const int ntiles = B * ((H + 3) / 4);
    const int grid = ntiles < CONV_MAX_PERSISTENT ? ntiles : CONV_MAX_PERSISTENT;
    if (Cin == 7)
        hipLaunchKernelGGL(conv_first_fwd_kernel<7>, dim3(grid), dim3(256), 0, st, x, w, bias, z, stat_partial, B, H);
    else if (Cin == 10)
        hipLaunchKernelGGL(conv_first_fwd_kernel<10>, dim3(grid), dim3(256), 0, st, x, w, bias, z, stat_partial, B, H);
    else
        return -2;
    if (n_partial) *n_partial = grid;
    return 0;
}

// ================================================================================================
// Cin = Cout = 64 forward (and input gradient with flipped weights)
// ================================================================================================
#define C64_LDA 129  // A tile [ci][px], +1 pad

// Pipeline per tap: [issue tap+1's A tile and weights -> registers] [MFMA over tap's LDS tiles]
// [LDS barrier] [commit registers] [LDS barrier].  BN partial sums stay in registers across tiles.
template <bool STATS>
__global__ __launch_bounds__(256) void conv64_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w9,
                                                         const float* __restrict__ bias, float* __restrict__ z,
                                                         float* __restrict__ stat_partial, int npix, int H, int W) {
    __shared__ __attribute__((aligned(16))) float As[64 * C64_LDA];
    __shared__ __attribute__((aligned(16))) float Wt[64 * 64];
    __shared__ float red[4 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    const int ntiles = (npix + 127) >> 7;
    const int g4 = (tid & 15) * 4;   // channel group of this thread's staging loads
    const int pxs = tid >> 4;        // staging pixel slot (0..15), pixels pxs + 16u
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    float4 ar[8], wr[4];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p0 = tile << 7;
        // validity mask of the 9 taps for this thread's 8 staging pixels
        unsigned vmask[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + pxs + 16 * u;
            unsigned m = 0;
            if (p < npix) {
                const int f = p % W, t = (p / W) % H;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dt = tap / 3 - 1, df = tap % 3 - 1;
                    if (t + dt >= 0 && t + dt < H && f + df >= 0 && f + df < W) m |= 1u << tap;
                }
            }
            vmask[u] = m;
        }
        // (macros, not lambdas: arrays captured by a lambda were demoted to scratch memory)
#define C64_ISSUE(tap_)                                                                                         \
    {                                                                                                           \
        const int tp_ = (tap_);                                                                                 \
        const int shift_ = (tp_ / 3 - 1) * W + (tp_ % 3 - 1);                                                   \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                         \
            const bool ok_ = (vmask[u] >> tp_) & 1u;                                                            \
            const float4 v_ = *reinterpret_cast<const float4*>(x + (size_t)(ok_ ? p0 + pxs + 16 * u + shift_ : 0) * 64 + g4); \
            ar[u] = ok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                 \
        }                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                           \
            wr[u] = reinterpret_cast<const float4*>(w9 + (size_t)tp_ * 4096)[tid + 256 * u];                    \
    }
#define C64_COMMIT()                                                                       \
    {                                                                                      \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                    \
            const int px_ = pxs + 16 * u;                                                  \
            As[(g4 + 0) * C64_LDA + px_] = ar[u].x;                                        \
            As[(g4 + 1) * C64_LDA + px_] = ar[u].y;                                        \
            As[(g4 + 2) * C64_LDA + px_] = ar[u].z;                                        \
            As[(g4 + 3) * C64_LDA + px_] = ar[u].w;                                        \
        }                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) reinterpret_cast<float4*>(Wt)[tid + 256 * u] = wr[u]; \
    }
        C64_ISSUE(0)
        lds_barrier();   // previous tile's readers are done
        C64_COMMIT()
        lds_barrier();
        f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            C64_ISSUE(tap < 8 ? tap + 1 : 8)   // unconditional (tap 8 re-reads itself): no phi on the staged registers
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch loads ahead of the MFMA loop
            {
                // 2-step operand ring pinned with sched_barrier (each step = 2 MFMAs = 128 pipe cycles)
                float qa[2], qb0[2], qb1[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = 2 * u + hi;
                    qa[u] = As[k * C64_LDA + wave * 32 + li];
                    qb0[u] = Wt[k * 64 + li];
                    qb1[u] = Wt[k * 64 + 32 + li];
                }
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    const int u = s & 1;
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0] = MFMA_F32_32x32x2(qa[u], qb0[u], acc[0]);
                    acc[1] = MFMA_F32_32x32x2(qa[u], qb1[u], acc[1]);
                    if (s + 2 < 32) {
                        const int k = 2 * (s + 2) + hi;
                        qa[u] = As[k * C64_LDA + wave * 32 + li];
                        qb0[u] = Wt[k * 64 + li];
                        qb1[u] = Wt[k * 64 + 32 + li];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // unconditional commit (tap 8 re-commits itself): with the staged registers used only inside an
            // `if (tap < 8)` block the compiler sank the global loads behind the barrier
            lds_barrier();
            C64_COMMIT()
            lds_barrier();
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float bv = bias ? bias[c * 32 + li] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wave * 32 + mfma_row(r, hi);
                if (p < npix) {
                    const float v = acc[c][r] + bv;
                    z[(size_t)p * 64 + c * 32 + li] = v;
                    s1[c] += v;
                    s2[c] = fmaf(v, v, s2[c]);
                }
            }
        }
    }
    if (STATS) {
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
        if (tid < 128) stat_partial[(size_t)blockIdx.x * 128 + tid] = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
    }
}

int launch_conv64_fwd(hipStream_t st, const float* x, const float* w9, const float* bias, float* z,
                      float* stat_partial, int* n_partial, int B, int H, int W) {
    const int npix = B * H * W;
    const int ntiles = (npix + 127) / 128;
    const int grid = ntiles < CONV_MAX_PERSISTENT ? ntiles : CONV_MAX_PERSISTENT;
    if (stat_partial)
        hipLaunchKernelGGL(conv64_fwd_kernel<true>, dim3(grid), dim3(256), 0, st, x, w9, bias, z, stat_partial, npix, H, W);
    else
        hipLaunchKernelGGL(conv64_fwd_kernel<false>, dim3(grid), dim3(256), 0, st, x, w9, bias, z, stat_partial, npix, H, W);
    if (n_partial) *n_partial = grid;
    return 0;
}

// dgrad weights: wt[kh'][kw'][co][ci] = w[2-kh'][2-kw'][ci][co]
__global__ __launch_bounds__(256) void flip_weights_kernel(const float* __restrict__ w, float* __restrict__ wt) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 9 * 4096) return;
    const int tap = idx >> 12, rem = idx & 4095, co = rem >> 6, ci = rem & 63;
    wt[idx] = w[(8 - tap) * 4096 + ci * 64 + co];
}

int launch_flip_weights(hipStream_t st, const float* w, float* wt) {
    hipLaunchKernelGGL(flip_weights_kernel, dim3(9 * 4096 / 256), dim3(256), 0, st, w, wt);
    return 0;
}

// ================================================================================================
// first layer kernel/bias gradient
// ================================================================================================
// Pipeline per tile: [issue next tile's patch + dz loads -> registers] [MFMA over the current LDS tile]
// [barrier] [commit registers to LDS] [barrier].
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                               float* __restrict__ slab, int B, int H) {
    using G = FirstGeom<CIN>;
    constexpr int K = G::K, ROWF = G::ROWF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzl = smem;                 // [256 px][64 co]
    float* patch = smem + 256 * 64;    // [6][66][CIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    // wave w owns pixel row w of the tile (64 px) and all 4 output tiles (k 0-31 / 32-63) x (co 0-31 / 32-63):
    // 4 independent accumulator chains per wave (a single dependent chain ran the MFMA pipe at ~30%)
    constexpr int NKT = G::NKT;
    int koffs[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int k = kt * 32 + li;   // this lane's im2col column (A-operand row) in k-tile kt
        const int kh = k / (3 * CIN);
        // k == K is the bias row: its A operand is the constant 1.0 parked at patch[PATCH];
        // k > K (padding of the last k-tile) reads the 0.0 parked at patch[PATCH + 1]
        koffs[kt] = (k < K) ? wave * ROWF + kh * ROWF + (k - kh * 3 * CIN) : (k == K ? -1 : -2);
    }
    const int tiles_per_img = (H + 3) >> 2;
    const int ntiles = B * tiles_per_img;
    PatchStage<CIN> stg;
    float4 dzr[16];
    auto issue_dz = [&](int b, int t0) {
        const float4* src = reinterpret_cast<const float4*>(dz + (size_t)(b * H + t0) * 4096);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = tid + 256 * u;   // float4 index; 1024 per image row
            dzr[u] = (t0 + (idx >> 10) < H) ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit_dz = [&]() {
#pragma unroll
        for (int u = 0; u < 16; ++u) reinterpret_cast<float4*>(dzl)[tid + 256 * u] = dzr[u];
    };
    zero_patch<CIN>(patch, tid);
    if (tid == 0) { patch[G::PATCH] = 1.f; patch[G::PATCH + 1] = 0.f; }   // A operands of the bias row / k padding
    int tile = blockIdx.x;
    __syncthreads();
    if (tile < ntiles) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * 4;
        stg.issue(x, b, t0, H, tid);
        issue_dz(b, t0);
        stg.commit(patch, tid);
        commit_dz();
    }
    __syncthreads();
    f32x16 acc[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[kt][ct] = zero16();
    for (; tile < ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        const bool has_next = nxt < ntiles;
        if (has_next) {
            const int nb = nxt / tiles_per_img, nt0 = (nxt - nb * tiles_per_img) * 4;
            stg.issue(x, nb, nt0, H, tid);
            issue_dz(nb, nt0);
        }
        // per step (2 pixels): 2 A reads (k-tiles) + 2 B reads (co-tiles) feed 4 MFMAs; one-step pipeline
        auto ld = [&](int s, float (&a)[NKT], float& b0, float& b1) {
            const int f = 2 * s + hi;                       // pixel column in this wave's row
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) a[kt] = patch[koffs[kt] >= 0 ? koffs[kt] + f * CIN : G::PATCH - 1 - koffs[kt]];
            const float* bp = dzl + (wave * 64 + f) * 64 + li;
            b0 = bp[0];
            b1 = bp[32];
        };
        float a[NKT], b0, b1;
        ld(0, a, b0, b1);
#pragma unroll 8
        for (int s = 0; s < 32; ++s) {
            float na[NKT], nb0 = 0.f, nb1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) na[kt] = 0.f;
            if (s + 1 < 32) ld(s + 1, na, nb0, nb1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                acc[kt][0] = MFMA_F32_32x32x2(a[kt], b0, acc[kt][0]);
                acc[kt][1] = MFMA_F32_32x32x2(a[kt], b1, acc[kt][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) a[kt] = na[kt];
            b0 = nb0; b1 = nb1;
        }
        lds_barrier();          // every wave is done reading this tile
        if (has_next) {
            stg.commit(patch, tid);
            commit_dz();
        }
        lds_barrier();
    }
    // combine the 4 waves' partial sums through LDS (dzl is free now), one 32-row k-tile at a time, fixed order
    float* out = slab + (size_t)blockIdx.x * G::WG_SLAB;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) dzl[wave * 2048 + mfma_row(r, hi) * 64 + ct * 32 + li] = acc[kt][ct][r];
        __syncthreads();
        for (int i = tid; i < 2048; i += 256) out[kt * 2048 + i] = (dzl[i] + dzl[2048 + i]) + (dzl[4096 + i] + dzl[6144 + i]);
    }
}

// Fused first-layer backward: dz is never materialised.  The kernel reads the forward's pre-BN
// activations z plus the pooled tensors (p, dp) and forms, while committing a tile to LDS,
//   y = z*scale + shift;  dy = (y == p && p > 0) ? dp : 0      (maxpool + ReLU backward: y == p is exact,
//   xhat = (z - mean)*invstd; dz = scale*(dy - c1 - xhat*c2)     the forward computed p with the same fmaf)
// i.e. bn_pool_bwd_dz + conv_first_wgrad in one pass.  coef = [mean|invstd|scale|shift|c1|c2] x 64.
// Each thread owns a 4-row x 4-pixel x 4-channel block of the tile, so it needs ONE pooled column
// (PF = 4) and at most two pooled rows of p / dp.
template <int CIN, int PT>
__global__ __launch_bounds__(256, 2) void conv_first_wgrad_fused_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                                     const float* __restrict__ p, const float* __restrict__ dp,
                                                                     const unsigned char* __restrict__ amax,
                                                                     const float* __restrict__ coef, float* __restrict__ slab,
                                                                     int B, int H) {
    using G = FirstGeom<CIN>;
    constexpr int K = G::K, ROWF = G::ROWF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzl = smem;                 // [256 px][64 co]
    float* patch = smem + 256 * 64;    // [6][66][CIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    // wave w owns pixel row w of the tile (64 px) and all 4 output tiles (k 0-31 / 32-63) x (co 0-31 / 32-63):
    // 4 independent accumulator chains per wave (a single dependent chain ran the MFMA pipe at ~30%)
    constexpr int NKT = G::NKT;
    int koffs[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int k = kt * 32 + li;   // this lane's im2col column (A-operand row) in k-tile kt
        const int kh = k / (3 * CIN);
        // k == K is the bias row: its A operand is the constant 1.0 parked at patch[PATCH];
        // k > K (padding of the last k-tile) reads the 0.0 parked at patch[PATCH + 1]
        koffs[kt] = (k < K) ? wave * ROWF + kh * ROWF + (k - kh * 3 * CIN) : (k == K ? -1 : -2);
    }
    int pa[NKT], sa[NKT];    // patch address of this lane's A operand at step 0, and its stride per step (0 for the constant rows)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        pa[kt] = koffs[kt] >= 0 ? koffs[kt] + hi * CIN : G::PATCH - 1 - koffs[kt];
        sa[kt] = koffs[kt] >= 0 ? 2 * CIN : 0;
    }
    const int tiles_per_img = (H + 3) >> 2;
    const int ntiles = B * tiles_per_img;
    const int Hp = H / PT;
    const int g = tid & 15, q = tid >> 4;        // channel group, pooled column of this thread's block
    const float4 mu4 = reinterpret_cast<const float4*>(coef)[g], is4 = reinterpret_cast<const float4*>(coef + 64)[g];
    const float4 sc4 = reinterpret_cast<const float4*>(coef + 128)[g], sh4 = reinterpret_cast<const float4*>(coef + 192)[g];
    const float4 c14 = reinterpret_cast<const float4*>(coef + 256)[g], c24 = reinterpret_cast<const float4*>(coef + 320)[g];
    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
    (void)sh4;
    float ka[4], kb[4];   // dz = z*ka + kb (+ scale*dp at the argmax): ka = -scale*c2*invstd, kb = -scale*c1 - ka*mean
    {
        const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
        const float c1[4] = {c14.x, c14.y, c14.z, c14.w}, c2[4] = {c24.x, c24.y, c24.z, c24.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ka[c] = -sc[c] * c2[c] * is[c];
            kb[c] = -sc[c] * c1[c] - ka[c] * mu[c];
        }
    }
    PatchStage<CIN> stg;
    float4 zr[16], pr[2], dpr[2];
    unsigned ar[2];    // window positions of the extreme, 4 channels packed (amax bytes)
    int st_t0 = 0;     // t0 of the staged tile
    const float* zn = z;   // staged tile: this thread's first z element (row 0, pixel 4q, channels 4g..)
    // the 16 z loads of a tile are issued ONE PER TWO K-STEPS inside the previous tile's MFMA loop (WGF_ZLOAD): issued
    // in a burst at the top of the tile they kept the wave ~6 k cycles in back-pressured VMEM issue while HBM then
    // sat idle for the rest of the tile (cycle counters: 3.9 TB/s of the 6.3 the card streams)
    auto issue_small = [&](int b, int t0) {
        st_t0 = t0;
        zn = z + ((size_t)(b * H + t0) * 64 + 4 * q) * 64 + g * 4;
        const int pr0 = t0 / PT;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int prow = min(pr0 + v, Hp - 1);
            const size_t off = (((size_t)b * Hp + prow) * 16 + q) * 64 + g * 4;
            pr[v] = *reinterpret_cast<const float4*>(p + off);
            dpr[v] = *reinterpret_cast<const float4*>(dp + off);
            ar[v] = *reinterpret_cast<const unsigned*>(amax + off);
        }
    };
#define WGF_ZLOAD(u_)                                                                                   \
    {                                                                                                   \
        const bool ok_ = st_t0 + ((u_) >> 2) < H;          /* uniform */                                \
        const u32x4 v_ = *reinterpret_cast<const u32x4*>(ok_ ? zn + (((u_) >> 2) * 64 + ((u_) & 3)) * 64 : z) & (ok_ ? 0xffffffffu : 0u); \
        zr[u_] = make_float4(__uint_as_float(v_.x), __uint_as_float(v_.y), __uint_as_float(v_.z), __uint_as_float(v_.w)); \
    }
    auto commit_dz = [&]() {
        // dz = scale*(dy - c1 - xhat*c2) = fma(z, ka, kb) + (argmax ? scale*dp : 0).  The argmax is the ONE position the
        // forward recorded (MaxPoolGrad routes to a single element): testing y == p instead double-counts dp whenever
        // two pixels of a window round to the same fp32 y — a few windows per step at B = 32, each worth ~1e-3 of a
        // kernel gradient's scale.  Per pooled row and channel: key = that position (255 = no gradient: p <= 0) and
        // kbg = kb + scale*dp, so that an element costs a compare, a select and one fma: fma(z, ka, hit ? kbg : kb).
        const int pr0 = st_t0 / PT;
        unsigned key[2][4];
        float kbg[2][4];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const float pp[4] = {pr[v].x, pr[v].y, pr[v].z, pr[v].w}, dd[4] = {dpr[v].x, dpr[v].y, dpr[v].z, dpr[v].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                key[v][c] = pp[c] > 0.f ? ((ar[v] >> (8 * c)) & 255u) : 255u;
                kbg[v][c] = kb[c] + sc[c] * dd[c];
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int row = u >> 2, px = 4 * q + (u & 3);
            const int t = st_t0 + row;                                              // uniform
            const int v = (t / PT) != pr0 ? 1 : 0;
            const unsigned pos = (unsigned)((t - (t / PT) * PT) * 4 + (u & 3));    // this pixel's position in its window
            const float zz[4] = {zr[u].x, zr[u].y, zr[u].z, zr[u].w};
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned kc = v ? key[1][c] : key[0][c];
                const float kg = v ? kbg[1][c] : kbg[0][c];
                o[c] = fmaf(zz[c], ka[c], kc == pos ? kg : kb[c]);
            }
            if (t >= H) o[0] = o[1] = o[2] = o[3] = 0.f;                            // ragged last tile (uniform)
            *reinterpret_cast<float4*>(dzl + (size_t)(row * 64 + px) * 64 + g * 4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    zero_patch<CIN>(patch, tid);
    if (tid == 0) { patch[G::PATCH] = 1.f; patch[G::PATCH + 1] = 0.f; }   // A operands of the bias row / k padding
    int tile = blockIdx.x;
    __syncthreads();
    if (tile < ntiles) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * 4;
        stg.issue(x, b, t0, H, tid);
        issue_small(b, t0);
#pragma unroll
        for (int u = 0; u < 16; ++u) WGF_ZLOAD(u)
        stg.commit(patch, tid);
        commit_dz();
    }
    __syncthreads();
    f32x16 acc[NKT][2];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[kt][ct] = zero16();
    for (; tile < ntiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        const bool has_next = nxt < ntiles;
        {   // unconditional (the last tile re-stages itself): a conditional issue makes the staged registers a phi
            const int st = has_next ? nxt : tile;
            const int nb = st / tiles_per_img, nt0 = (st - nb * tiles_per_img) * 4;
            stg.issue(x, nb, nt0, H, tid);
            issue_small(nb, nt0);
        }
        // per step (2 pixels): 2 A reads (k-tiles) + 2 B reads (co-tiles) feed 4 MFMAs; one-step pipeline
        const float* bp0 = dzl + (wave * 64 + hi) * 64 + li;
        // A address = pa[kt] + s * sa[kt] (one multiply-add per read) and B address = bp0 + immediate: with per-step
        // selects the fully unrolled loop had 64 loop-invariant addresses hoisted into registers
        auto ld = [&](int s, float (&a)[NKT], float& b0, float& b1) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) a[kt] = patch[pa[kt] + s * sa[kt]];
            b0 = bp0[s * 128];
            b1 = bp0[s * 128 + 32];
        };
        float a[NKT], b0, b1;
        ld(0, a, b0, b1);
#pragma unroll
        for (int s = 0; s < 32; ++s) {      // fully unrolled: precise lgkmcnt, compile-time z slot
            float na[NKT], nb0 = 0.f, nb1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) na[kt] = 0.f;
            if (s + 1 < 32) ld(s + 1, na, nb0, nb1);
            if ((s & 1) == 0) WGF_ZLOAD(s >> 1)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                acc[kt][0] = MFMA_F32_32x32x2(a[kt], b0, acc[kt][0]);
                acc[kt][1] = MFMA_F32_32x32x2(a[kt], b1, acc[kt][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) a[kt] = na[kt];
            b0 = nb0; b1 = nb1;
        }
        lds_barrier();
        if (has_next) {
            stg.commit(patch, tid);
            commit_dz();
        }
        lds_barrier();
    }
    // combine the 4 waves' partial sums through LDS (dzl is free now), one 32-row k-tile at a time, fixed order
    float* out = slab + (size_t)blockIdx.x * G::WG_SLAB;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) dzl[wave * 2048 + mfma_row(r, hi) * 64 + ct * 32 + li] = acc[kt][ct][r];
        __syncthreads();
        for (int i = tid; i < 2048; i += 256) out[kt * 2048 + i] = (dzl[i] + dzl[2048 + i]) + (dzl[4096 + i] + dzl[6144 + i]);
    }
}

int launch_conv_first_wgrad_fused(hipStream_t st, const float* x, const float* z, const float* p, const float* dp,
                                  const unsigned char* amax, const float* coef, float* slab, int* n_slab, int B, int H,
                                  int Cin, int pt, int pf) {
    if ((Cin != 7 && Cin != 10) || pf != 4 || H % pt || !amax) return -2;
    const int ntiles = B * ((H + 3) / 4);
    const int grid = ntiles < WGRAD_MAX_BLOCKS ? ntiles : WGRAD_MAX_BLOCKS;
#define LAUNCH_FUSED(CI, PT)                                                                                      \
    {                                                                                                             \
        const size_t smem = (size_t)(256 * 64 + FirstGeom<CI>::PATCH + 4) * sizeof(float);                        \
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_wgrad_fused_kernel<CI, PT>),                 \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                               \
        hipLaunchKernelGGL((conv_first_wgrad_fused_kernel<CI, PT>), dim3(grid), dim3(256), smem, st, x, z, p, dp, \
                           amax, coef, slab, B, H);                                                                     \
    }
#define LAUNCH_FUSED_PT(CI)                                                                                       \
    if (pt == 5) LAUNCH_FUSED(CI, 5) else if (pt == 4) LAUNCH_FUSED(CI, 4) else if (pt == 2) LAUNCH_FUSED(CI, 2)  \
    else if (pt == 1) LAUNCH_FUSED(CI, 1) else return -2;
    if (Cin == 7) { LAUNCH_FUSED_PT(7) } else { LAUNCH_FUSED_PT(10) }
#undef LAUNCH_FUSED_PT
#undef LAUNCH_FUSED
    *n_slab = grid;
    return 0;
}

int launch_conv_first_wgrad(hipStream_t st, const float* x, const float* dz, float* slab, int* n_slab,
                            int B, int H, int Cin) {
    const int ntiles = B * ((H + 3) / 4);
    const int grid = ntiles < WGRAD_MAX_BLOCKS ? ntiles : WGRAD_MAX_BLOCKS;
#define LAUNCH_WG1(CI)                                                                                   \
    {                                                                                                    \
        const size_t smem = (size_t)(256 * 64 + FirstGeom<CI>::PATCH + 4) * sizeof(float);               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_wgrad_kernel<CI>),                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                      \
        hipLaunchKernelGGL(conv_first_wgrad_kernel<CI>, dim3(grid), dim3(256), smem, st, x, dz, slab, B, H); \
    }
    if (Cin == 7) LAUNCH_WG1(7) else if (Cin == 10) LAUNCH_WG1(10) else return -2;
#undef LAUNCH_WG1
    *n_slab = grid;
    return 0;
}

// ================================================================================================
// Cin = Cout = 64 kernel/bias gradient.  Chunk = R rows x W cols = 128 pixels of one image.
// slab layout per block: [9][64 ci][64 co] then [64] bias partial.
// ================================================================================================
#define WG64_SLAB (9 * 4096 + 64)

constexpr int conv64_wgrad_chunk_px(int W) { return W == 16 ? 128 : 96; }

template <int WLOG2>
__global__ __launch_bounds__(256) void conv64_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                           float* __restrict__ slab, int B, int H) {
    // pixels per chunk: 128 at W = 16 (78 KB of LDS), 96 elsewhere so that two blocks fit a CU (at W = 4 a 128-pixel
    // chunk needs 84 KB: one block per CU, nobody to hide the chunk load behind — measured 32 % MFMA utilisation)
    constexpr int W = 1 << WLOG2, PXC = conv64_wgrad_chunk_px(W), R = PXC / W, RW = W + 2, RR = R + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xr = smem;                    // [RR][RW][64]
    float* dzl = smem + RR * RW * 64;    // [PXC][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    const int cih = wave >> 1, coh = wave & 1;
    const int chunks_per_img = (H + R - 1) / R;
    const int nchunks = B * chunks_per_img;
    f32x16 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = zero16();
    float brun = 0.f;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int b = chunk / chunks_per_img, t0 = (chunk - b * chunks_per_img) * R;
        __syncthreads();
        // halo region of x
        for (int idx = tid; idx < RR * RW * 16; idx += 256) {
            const int g = idx & 15, pix = idx >> 4;
            const int rr = pix / RW, cc = pix - rr * RW;
            const int t = t0 - 1 + rr, f = cc - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < H && f >= 0 && f < W)
                v = *reinterpret_cast<const float4*>(x + ((size_t)(b * H + t) * W + f) * 64 + g * 4);
            reinterpret_cast<float4*>(xr)[idx] = v;
        }
#pragma unroll
        for (int u = 0; u < PXC / 16; ++u) {
            const int idx = tid + 256 * u;  // float4 index, PXC * 16 per chunk
            const int px = idx >> 4;
            const int t = t0 + (px >> WLOG2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < H) v = reinterpret_cast<const float4*>(dz + (size_t)(b * H + t0) * W * 64)[idx];
            reinterpret_cast<float4*>(dzl)[idx] = v;
        }
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            for (int px = 0; px < PXC; ++px) s += dzl[px * 64 + tid];
            brun += s;
        }
        {
            // one-step operand pipeline pinned with sched_barrier: the 10 LDS reads of step s+1 are in
            // flight under the 9 MFMAs (576 pipe cycles) of step s
            float ca[9], cb;
            auto ldw = [&](int s, float (&a9)[9], float& bb) {
                const int px = 2 * s + hi;
                const int r = px >> WLOG2, c = px & (W - 1);
                bb = dzl[px * 64 + coh * 32 + li];
                const float* ap = xr + (r * RW + c) * 64 + cih * 32 + li;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) a9[tap] = ap[((tap / 3) * RW + (tap % 3)) * 64];
            };
            ldw(0, ca, cb);
            // fully unrolled: across a loop back-edge the waitcnt bookkeeping fell back to lgkmcnt(0) in front of the
            // MFMAs, i.e. it also waited for the prefetch just issued (one LDS round trip per two steps)
#pragma unroll
            for (int s = 0; s < PXC / 2; ++s) {
                float na[9], nb = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) na[tap] = 0.f;
                if (s + 1 < PXC / 2) ldw(s + 1, na, nb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) acc[tap] = MFMA_F32_32x32x2(ca[tap], cb, acc[tap]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) ca[tap] = na[tap];
                cb = nb;
            }
        }
    }
    float* out = slab + (size_t)blockIdx.x * WG64_SLAB;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[tap * 4096 + (cih * 32 + mfma_row(r, hi)) * 64 + coh * 32 + li] = acc[tap][r];
    if (tid < 64) out[9 * 4096 + tid] = brun;
}

int launch_conv64_wgrad(hipStream_t st, const float* x, const float* dz, float* slab, int* n_slab,
                        int B, int H, int W) {
    int wl = -1;
    if (W == 16) wl = 4; else if (W == 4) wl = 2; else if (W == 8) wl = 3; else if (W == 32) wl = 5; else if (W == 2) wl = 1;
    if (wl < 0) return -2;
    const int PXC = conv64_wgrad_chunk_px(W), R = PXC / W;
    const int nchunks = B * ((H + R - 1) / R);
    const int grid = nchunks < WGRAD_MAX_BLOCKS ? nchunks : WGRAD_MAX_BLOCKS;
    const size_t smem = (size_t)((R + 2) * (W + 2) * 64 + PXC * 64) * sizeof(float);
#define LAUNCH_WG(L)                                                                                          \
    {                                                                                                         \
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_wgrad_kernel<L>),                            \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                           \
        hipLaunchKernelGGL(conv64_wgrad_kernel<L>, dim3(grid), dim3(256), smem, st, x, dz, slab, B, H);       \
    }
    switch (wl) {
        case 1: LAUNCH_WG(1) break;
        case 2: LAUNCH_WG(2) break;
        case 3: LAUNCH_WG(3) break;
        case 4: LAUNCH_WG(4) break;
        case 5: LAUNCH_WG(5) break;
    }
#undef LAUNCH_WG
    *n_slab = grid;
    return 0;
}
