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

#define CONV_MAX_PERSISTENT 1024
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
};

// patch[r][c][ci] for image rows t0-1..t0+4, cols -1..64 (zeros outside the image)
template <int CIN>
__device__ __forceinline__ void stage_patch(float* patch, const float* __restrict__ x, int b, int t0, int H, int tid) {
    constexpr int ROWF = FirstGeom<CIN>::ROWF;
    for (int idx = tid; idx < 6 * ROWF; idx += 256) {
        const int r = idx / ROWF, rem = idx - r * ROWF;
        const int c = rem / CIN, ci = rem - c * CIN;
        const int t = t0 - 1 + r;
        float v = 0.f;
        if (t >= 0 && t < H && c >= 1 && c <= 64) v = x[((size_t)(b * H + t) * 64 + (c - 1)) * CIN + ci];
        patch[idx] = v;
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ z,
                                                             float* __restrict__ stat_partial, int B, int H) {
    using G = FirstGeom<CIN>;
    constexpr int K = G::K, KPAD = G::KPAD, ROWF = G::ROWF;
    __shared__ __attribute__((aligned(16))) float Wl[KPAD * 64];
    __shared__ __attribute__((aligned(16))) float patch[G::PATCH];
    __shared__ float red[4 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    for (int idx = tid; idx < KPAD * 64; idx += 256) {
        const int k = idx >> 6, co = idx & 63;
        Wl[idx] = (k < K) ? w[idx] : (k == K ? (bias ? bias[co] : 0.f) : 0.f);
    }
    const int tiles_per_img = (H + 3) >> 2;
    const int ntiles = B * tiles_per_img;
    float run = 0.f;  // tid<64: sum z of channel tid; 64<=tid<128: sum z^2 of channel tid-64
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * 4;
        __syncthreads();  // previous tile's patch reads / red reads are done
        stage_patch<CIN>(patch, x, b, t0, H, tid);
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[p][c] = zero16();
        const int base0 = (wave * 66 + li) * CIN;  // pixel (row = wave, f = li) at tap (0,0)
        const int base1 = base0 + 32 * CIN;
#pragma unroll
        for (int s = 0; s < KPAD / 2; ++s) {
            const int k = 2 * s + hi;
            const int kh = k / (3 * CIN);
            const int off = kh * ROWF + (k - kh * 3 * CIN);
            float a0, a1;
            if (2 * s + 1 < K) {  // both lanes' k are real im2col columns (compile-time per s)
                a0 = patch[base0 + off];
                a1 = patch[base1 + off];
            } else {
                const float one = (k == K) ? 1.f : 0.f;
                a0 = (k < K) ? patch[base0 + (k < K ? off : 0)] : one;
                a1 = (k < K) ? patch[base1 + (k < K ? off : 0)] : one;
            }
            const float b0 = Wl[k * 64 + li], b1 = Wl[k * 64 + 32 + li];
            acc[0][0] = MFMA_F32_32x32x2(a0, b0, acc[0][0]);
            acc[0][1] = MFMA_F32_32x32x2(a0, b1, acc[0][1]);
            acc[1][0] = MFMA_F32_32x32x2(a1, b0, acc[1][0]);
            acc[1][1] = MFMA_F32_32x32x2(a1, b1, acc[1][1]);
        }
        const int t = t0 + wave;
        const bool row_ok = t < H;
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
        if (row_ok) {
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
                        s2[c] += v * v;
                    }
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
            if (tid < 128) run += red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
        }
    }
    if (stat_partial && tid < 128) stat_partial[(size_t)blockIdx.x * 128 + tid] = run;
}

int launch_conv_first_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* z,
                          float* stat_partial, int* n_partial, int B, int H, int Cin) {
    const int ntiles = B * ((H + 3) / 4);
    const int grid = ntiles < CONV_MAX_PERSISTENT ? ntiles : CONV_MAX_PERSISTENT;
    if (Cin == 7)
        hipLaunchKernelGGL(conv_first_fwd_kernel<7>, dim3(grid), dim3(256), 0, st, x, w, bias, z, stat_partial, B, H);
    else
        return -2;
    if (n_partial) *n_partial = grid;
    return 0;
}

// ================================================================================================
// Cin = Cout = 64 forward (and input gradient with flipped weights)
// ================================================================================================
#define C64_LDA 129  // A tile [ci][px], +1 pad

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
    float run = 0.f;
    const int g4 = (tid & 15) * 4;   // channel group of this thread's staging loads
    const int pxs = tid >> 4;        // staging pixel slot (0..15), pixels pxs + 16u
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
        f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dt = tap / 3 - 1, df = tap % 3 - 1;
            const int shift = dt * W + df;
            __syncthreads();  // previous tap's MFMA reads done
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int px = pxs + 16 * u;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((vmask[u] >> tap) & 1u)
                    v = *reinterpret_cast<const float4*>(x + (size_t)(p0 + px + shift) * 64 + g4);
                As[(g4 + 0) * C64_LDA + px] = v.x;
                As[(g4 + 1) * C64_LDA + px] = v.y;
                As[(g4 + 2) * C64_LDA + px] = v.z;
                As[(g4 + 3) * C64_LDA + px] = v.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + 256 * u;
                reinterpret_cast<float4*>(Wt)[idx] = reinterpret_cast<const float4*>(w9 + (size_t)tap * 4096)[idx];
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const int k = 2 * s + hi;
                const float a = As[k * C64_LDA + wave * 32 + li];
                const float b0 = Wt[k * 64 + li], b1 = Wt[k * 64 + 32 + li];
                acc[0] = MFMA_F32_32x32x2(a, b0, acc[0]);
                acc[1] = MFMA_F32_32x32x2(a, b1, acc[1]);
            }
        }
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
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
                    s2[c] += v * v;
                }
            }
        }
        if (STATS) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                s1[c] += __shfl_xor(s1[c], 32);
                s2[c] += __shfl_xor(s2[c], 32);
            }
            __syncthreads();  // red from the previous tile consumed
            if (hi == 0) {
                red[wave * 128 + li] = s1[0];
                red[wave * 128 + 32 + li] = s1[1];
                red[wave * 128 + 64 + li] = s2[0];
                red[wave * 128 + 96 + li] = s2[1];
            }
            __syncthreads();
            if (tid < 128) run += red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
        }
    }
    if (STATS && tid < 128) stat_partial[(size_t)blockIdx.x * 128 + tid] = run;
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
template <int CIN>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                               float* __restrict__ slab, int B, int H) {
    using G = FirstGeom<CIN>;
    constexpr int K = G::K, KPAD = G::KPAD, ROWF = G::ROWF;
    static_assert(KPAD == 64, "one 64x64 output (4 wave tiles) per block");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dzl = smem;                 // [256 px][64 co]
    float* patch = smem + 256 * 64;    // [6][66][CIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    const int kt = wave >> 1, ct = wave & 1;
    const int k = kt * 32 + li;  // this lane's im2col column (A-operand row)
    const int kh = k / (3 * CIN);
    const int koff = (k < K) ? kh * ROWF + (k - kh * 3 * CIN) : 0;
    const float kone = (k == K) ? 1.f : 0.f;
    const int tiles_per_img = (H + 3) >> 2;
    const int ntiles = B * tiles_per_img;
    f32x16 acc = zero16();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * 4;
        __syncthreads();
        stage_patch<CIN>(patch, x, b, t0, H, tid);
        // dz tile: 4 rows x 64 px x 64 co, contiguous in memory when all rows are inside the image
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = tid + 256 * u;       // float4 index, 4096 per tile
            const int row = idx >> 10;           // 1024 float4 per image row
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + row < H) v = reinterpret_cast<const float4*>(dz + (size_t)(b * H + t0) * 4096)[idx];
            reinterpret_cast<float4*>(dzl)[idx] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int s = 0; s < 128; ++s) {
            const int px = 2 * s + hi;
            const int row = px >> 6, f = px & 63;
            const float a = (k < K) ? patch[(row * 66 + f) * CIN + koff] : kone;
            const float bb = dzl[px * 64 + ct * 32 + li];
            acc = MFMA_F32_32x32x2(a, bb, acc);
        }
    }
    float* out = slab + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(kt * 32 + mfma_row(r, hi)) * 64 + ct * 32 + li] = acc[r];
}

int launch_conv_first_wgrad(hipStream_t st, const float* x, const float* dz, float* slab, int* n_slab,
                            int B, int H, int Cin) {
    const int ntiles = B * ((H + 3) / 4);
    const int grid = ntiles < WGRAD_MAX_BLOCKS ? ntiles : WGRAD_MAX_BLOCKS;
    if (Cin == 7) {
        const size_t smem = (size_t)(256 * 64 + FirstGeom<7>::PATCH) * sizeof(float);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_wgrad_kernel<7>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(conv_first_wgrad_kernel<7>, dim3(grid), dim3(256), smem, st, x, dz, slab, B, H);
    } else
        return -2;
    *n_slab = grid;
    return 0;
}

// ================================================================================================
// Cin = Cout = 64 kernel/bias gradient.  Chunk = R rows x W cols = 128 pixels of one image.
// slab layout per block: [9][64 ci][64 co] then [64] bias partial.
// ================================================================================================
#define WG64_SLAB (9 * 4096 + 64)

template <int WLOG2>
__global__ __launch_bounds__(256) void conv64_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                           float* __restrict__ slab, int B, int H) {
    constexpr int W = 1 << WLOG2, R = 128 / W, RW = W + 2, RR = R + 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xr = smem;                    // [RR][RW][64]
    float* dzl = smem + RR * RW * 64;    // [128][64]
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
        for (int u = 0; u < 8; ++u) {
            const int idx = tid + 256 * u;  // float4 index, 2048 per chunk
            const int px = idx >> 4;
            const int t = t0 + (px >> WLOG2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < H) v = reinterpret_cast<const float4*>(dz + (size_t)(b * H + t0) * W * 64)[idx];
            reinterpret_cast<float4*>(dzl)[idx] = v;
        }
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            for (int px = 0; px < 128; ++px) s += dzl[px * 64 + tid];
            brun += s;
        }
#pragma unroll 2
        for (int s = 0; s < 64; ++s) {
            const int px = 2 * s + hi;
            const int r = px >> WLOG2, c = px & (W - 1);
            const float bb = dzl[px * 64 + coh * 32 + li];
            const float* ap = xr + (r * RW + c) * 64 + cih * 32 + li;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float a = ap[((tap / 3) * RW + (tap % 3)) * 64];
                acc[tap] = MFMA_F32_32x32x2(a, bb, acc[tap]);
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
    const int R = 128 / W;
    const int nchunks = B * ((H + R - 1) / R);
    const int grid = nchunks < WGRAD_MAX_BLOCKS ? nchunks : WGRAD_MAX_BLOCKS;
    const size_t smem = (size_t)((R + 2) * (W + 2) * 64 + 128 * 64) * sizeof(float);
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
