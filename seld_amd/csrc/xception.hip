// xception.hip — the kernels of `xception_block`'s middle flow (spec/XCEPTION_BLOCK.md; model_config/xception_gru.json:2-11 names the
// block, the reference snapshot does not define it): residual modules of three  ReLU -> SeparableConv2D(64, 3, use_bias=False) ->
// BatchNormalization  on [B, S, 16, 64] NHWC fp32.  A separable convolution is a depthwise 3x3 (9 MACs per element: HBM-bound,
// written here as streaming kernels with float4 over the channels) followed by a 1x1 convolution = a [pixels, 64] x [64, 64]
// product (the fp32 MFMA GEMM of gemm.hip; its kernel gradient is the TN product of gemm.hip).  BatchNorm here has neither ReLU nor
// pooling behind it (the NEXT unit's ReLU is applied by its depthwise kernel on load), so it gets plain statistics / apply /
// backward kernels; the per-channel finalisation is bn_pool.hip's.
//
//   dw3x3_fwd        y[p][c]  = sum_taps k[tap][c] relu(x[p + tap][c])                       ('same' padding)
//   dw3x3_bwd_data   dx[p][c] = [x[p][c] > 0] sum_taps k[tap][c] dy[p - tap][c]  (+ add[p][c]: the residual branch's gradient)
//   dw3x3_bwd_w      dk[tap][c] = sum_p relu(x[p + tap][c]) dy[p][c]                         (per-workgroup slabs, fixed-order reduce)
//                    (fusing the two backward kernels into one pass over x / dy was measured: 5.1 ms per step against 3.0 — the
//                    row-per-workgroup shape the kernel-gradient sums need has 1/40 of the data kernel's parallelism)
//   bn_stats_plain   per-workgroup [sum z | sum z^2]                                        -> bn_finalize (bn_pool.hip)
//   bn_apply         out = z scale + shift (+ res)
//   bn_bwd_reduce_plain   per-workgroup [sum dy | sum dy xhat]                              -> bn_bwd_finalize
//   bn_bwd_dz_plain  dz = scale (dy - c1 - xhat c2)
#include "common.h"

#define XC_MAX_PARTIAL 512      // depthwise kernel-gradient slabs (one per workgroup of image rows): 2048 cost 0.4 ms per step more (slab traffic + combine)
#define XC_BN_PARTIAL 512       // BatchNorm partial sums: the single-workgroup finalisation reads all of them
int xc_partial_capacity() { return XC_MAX_PARTIAL; }

__device__ __forceinline__ float4 relu4(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }
__device__ __forceinline__ float4 fma4v(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// one thread per (pixel, group of 4 channels); C = 64 -> 16 groups; k [9][64] (Keras depthwise_kernel [3,3,64,1]).  (A sliding-window
// form — a thread walks image rows, keeps its 3 x 3 neighbourhood in registers and loads one new row per output — was measured: 1.8
// against 1.4 ms per step forward: the nine L1-served loads of 4.9 M independent threads beat a third of the loads from 0.3 M.)
template <bool BWD>
__global__ __launch_bounds__(256) void dw3x3_kernel(const float* __restrict__ src, const float* __restrict__ k, const float* __restrict__ xin,
                                                    const float* __restrict__ add, float* __restrict__ dst, int64_t npix, int H, int W) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npix * 16) return;
    const int g = (int)(gid & 15);
    const int64_t p = gid >> 4;
    const int f = (int)(p % W), t = (int)((p / W) % H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dt = tap / 3 - 1, df = tap % 3 - 1;
        // forward: y[p] += k[tap] relu(x[p + tap]);  backward: dx[p] += k[tap] dy[p - tap]
        const int tt = BWD ? t - dt : t + dt, ff = BWD ? f - df : f + df;
        if (tt >= 0 && tt < H && ff >= 0 && ff < W) {
            float4 v = *reinterpret_cast<const float4*>(src + (p + (int64_t)(BWD ? -1 : 1) * (dt * W + df)) * 64 + 4 * g);
            if (!BWD) v = relu4(v);
            acc = fma4v(reinterpret_cast<const float4*>(k + tap * 64)[g], v, acc);
        }
    }
    if (BWD) {
        const float4 xv = *reinterpret_cast<const float4*>(xin + p * 64 + 4 * g);
        acc = make_float4(xv.x > 0.f ? acc.x : 0.f, xv.y > 0.f ? acc.y : 0.f, xv.z > 0.f ? acc.z : 0.f, xv.w > 0.f ? acc.w : 0.f);
        if (add) {
            const float4 a = *reinterpret_cast<const float4*>(add + p * 64 + 4 * g);
            acc = make_float4(acc.x + a.x, acc.y + a.y, acc.z + a.z, acc.w + a.w);
        }
    }
    *reinterpret_cast<float4*>(dst + p * 64 + 4 * g) = acc;
}

int launch_dw3x3_fwd(hipStream_t st, const float* x, const float* k, float* y, int B, int H, int W) {
    const int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(dw3x3_kernel<false>, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, x, k, nullptr, nullptr, y, npix, H, W);
    return 0;
}
int launch_dw3x3_bwd_data(hipStream_t st, const float* dy, const float* k, const float* xin, const float* add, float* dx, int B, int H,
                          int W) {
    const int64_t npix = (int64_t)B * H * W;
    hipLaunchKernelGGL(dw3x3_kernel<true>, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, dy, k, xin, add, dx, npix, H, W);
    return 0;
}

// dk[tap][c] = sum_p relu(x[p + tap][c]) dy[p][c].  A workgroup takes whole image rows (W pixels x 64 channels = one contiguous
// 4 W floats x 16 line): thread (slot = tid >> 4, g = tid & 15) owns pixel column `slot` (W = 16) of the rows blockIdx, + gridDim, ...
// and keeps the 3 x 3 x 4 sums in registers; the three input rows it needs are read once each per output row through L1 (the
// neighbouring columns belong to the neighbouring threads of the same workgroup).  Slots combined through LDS in a fixed order
// -> slab[blockIdx][9][64]
__global__ __launch_bounds__(256) void dw3x3_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab,
                                                          int64_t nrows, int H, int W) {
    __shared__ float red[16][9 * 64 + 4];
    const int tid = threadIdx.x, g = tid & 15, slot = tid >> 4;
    float4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int t = (int)(r % H);
        for (int f = slot; f < W; f += 16) {
            const int64_t p = r * W + f;
            const float4 d = *reinterpret_cast<const float4*>(dy + p * 64 + 4 * g);
            float4 v[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {          // all nine loads in flight before the first FMA
                const int dt = tap / 3 - 1, df = tap % 3 - 1;
                const bool ok = t + dt >= 0 && t + dt < H && f + df >= 0 && f + df < W;
                v[tap] = ok ? *reinterpret_cast<const float4*>(x + (p + dt * W + df) * 64 + 4 * g) : zero4;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[tap] = fma4v(relu4(v[tap]), d, acc[tap]);
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<float4*>(&red[slot][tap * 64 + 4 * g]) = acc[tap];
    __syncthreads();
    for (int i = tid; i < 9 * 64; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) s += red[sl][i];
        slab[(size_t)blockIdx.x * 576 + i] = s;
    }
}

int launch_dw3x3_bwd_w(hipStream_t st, const float* x, const float* dy, float* slab, int* nslab, int B, int H, int W) {
    const int64_t nrows = (int64_t)B * H;
    int64_t blocks = nrows < XC_MAX_PARTIAL ? nrows : XC_MAX_PARTIAL;
    hipLaunchKernelGGL(dw3x3_bwd_w_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, dy, slab, nrows, H, W);
    *nslab = (int)blocks;
    return 0;
}

// per-workgroup [sum a | sum a b] over the pixels, 64 channels: STATS: a = b = z (sum z, sum z^2);
// BWD: a = dy, b = xhat = (z - mean) invstd (sum dy, sum dy xhat).  partial[blockIdx][128]
template <bool BWD>
__global__ __launch_bounds__(256) void xc_reduce_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        float* __restrict__ partial, int64_t npix) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, g = tid & 15, slot = tid >> 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, mu = s1, is = s1;
    if (BWD) { mu = reinterpret_cast<const float4*>(mean)[g]; is = reinterpret_cast<const float4*>(invstd)[g]; }
    for (int64_t p = (int64_t)blockIdx.x * 16 + slot; p < npix; p += (int64_t)gridDim.x * 16) {
        const float4 zv = *reinterpret_cast<const float4*>(z + p * 64 + 4 * g);
        if (BWD) {
            const float4 d = *reinterpret_cast<const float4*>(dy + p * 64 + 4 * g);
            s1 = make_float4(s1.x + d.x, s1.y + d.y, s1.z + d.z, s1.w + d.w);
            s2 = make_float4(s2.x + d.x * (zv.x - mu.x) * is.x, s2.y + d.y * (zv.y - mu.y) * is.y, s2.z + d.z * (zv.z - mu.z) * is.z,
                             s2.w + d.w * (zv.w - mu.w) * is.w);
        } else {
            s1 = make_float4(s1.x + zv.x, s1.y + zv.y, s1.z + zv.z, s1.w + zv.w);
            s2 = fma4v(zv, zv, s2);
        }
    }
    *reinterpret_cast<float4*>(&red[tid * 8]) = s1;
    *reinterpret_cast<float4*>(&red[tid * 8 + 4]) = s2;
    __syncthreads();
    if (tid < 128) {
        const int kind = tid >> 6, ch = tid & 63, gg = ch >> 2, cc = ch & 3;
        float s = 0.f;
        for (int sl = 0; sl < 16; ++sl) s += red[(sl * 16 + gg) * 8 + kind * 4 + cc];
        partial[(size_t)blockIdx.x * 128 + tid] = s;
    }
}

int launch_xc_bn_stats(hipStream_t st, const float* z, float* partial, int* npartial, int64_t npix) {
    int64_t blocks = (npix + 15) / 16;
    if (blocks > XC_BN_PARTIAL) blocks = XC_BN_PARTIAL;
    hipLaunchKernelGGL(xc_reduce_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, z, nullptr, nullptr, nullptr, partial, npix);
    *npartial = (int)blocks;
    return 0;
}
int launch_xc_bn_bwd_reduce(hipStream_t st, const float* z, const float* dy, const float* mean, const float* invstd, float* partial,
                            int* npartial, int64_t npix) {
    int64_t blocks = (npix + 15) / 16;
    if (blocks > XC_BN_PARTIAL) blocks = XC_BN_PARTIAL;
    hipLaunchKernelGGL(xc_reduce_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, z, dy, mean, invstd, partial, npix);
    *npartial = (int)blocks;
    return 0;
}

// out = z scale + shift (+ res)
__global__ __launch_bounds__(256) void xc_bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ res,
                                                          float* __restrict__ out, int64_t n4) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int g = (int)(gid & 15);
    float4 o = fma4v(reinterpret_cast<const float4*>(z)[gid], reinterpret_cast<const float4*>(scale)[g], reinterpret_cast<const float4*>(shift)[g]);
    if (res) {
        const float4 r = reinterpret_cast<const float4*>(res)[gid];
        o = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w);
    }
    reinterpret_cast<float4*>(out)[gid] = o;
}
int launch_xc_bn_apply(hipStream_t st, const float* z, const float* scale, const float* shift, const float* res, float* out, int64_t npix) {
    hipLaunchKernelGGL(xc_bn_apply_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, z, scale, shift, res, out, npix * 16);
    return 0;
}

// dz = scale (dy - c1 - xhat c2), xhat = (z - mean) invstd;  c1c2 = [c1 | c2]
__global__ __launch_bounds__(256) void xc_bn_bwd_dz_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ c1c2,
                                                           float* __restrict__ dz, int64_t n4) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int g = (int)(gid & 15);
    const float4 zv = reinterpret_cast<const float4*>(z)[gid], d = reinterpret_cast<const float4*>(dy)[gid];
    const float4 mu = reinterpret_cast<const float4*>(mean)[g], is = reinterpret_cast<const float4*>(invstd)[g];
    const float4 sc = reinterpret_cast<const float4*>(scale)[g], c1 = reinterpret_cast<const float4*>(c1c2)[g];
    const float4 c2 = reinterpret_cast<const float4*>(c1c2 + 64)[g];
    float4 o;
    o.x = sc.x * (d.x - c1.x - (zv.x - mu.x) * is.x * c2.x);
    o.y = sc.y * (d.y - c1.y - (zv.y - mu.y) * is.y * c2.y);
    o.z = sc.z * (d.z - c1.z - (zv.z - mu.z) * is.z * c2.z);
    o.w = sc.w * (d.w - c1.w - (zv.w - mu.w) * is.w * c2.w);
    reinterpret_cast<float4*>(dz)[gid] = o;
}
int launch_xc_bn_bwd_dz(hipStream_t st, const float* z, const float* dy, const float* mean, const float* invstd, const float* scale,
                        const float* c1c2, float* dz, int64_t npix) {
    hipLaunchKernelGGL(xc_bn_bwd_dz_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, z, dy, mean, invstd, scale, c1c2, dz,
                       npix * 16);
    return 0;
}

// identity BatchNorm coefficients for the exit pool (ReLU -> MaxPooling2D((1, 8)) reuses bn_relu_pool_fwd / bn_pool_bwd_dz):
// ident = [mean 0 | invstd 1 | scale 1 | shift 0 | c1 0 | c2 0] x 64
__global__ void xc_ident_kernel(float* ident) {
    const int i = threadIdx.x;      // 384
    ident[i] = (i >= 64 && i < 192) ? 1.f : 0.f;
}
int launch_xc_ident(hipStream_t st, float* ident) {
    hipLaunchKernelGGL(xc_ident_kernel, dim3(1), dim3(384), 0, st, ident);
    return 0;
}
