// resnet.hip — channel-count-generic layer kernels for `resnet50_block` (spec/RESNET50_BLOCK.md; model_config/resnet50_gru.json:2-11
// names the block, the reference snapshot does not define it).  Every convolution is a product — a 1x1 convolution directly (a
// frequency stride of 2 is a doubled row stride of the operand), a 3x3 convolution on im2col rows that the split-bf16 kernels form on
// load (stages 2-3), the conv blocks' implicit-GEMM kernels (stage 1) or an explicit im2col (stage 0) —, dispatched by shape at the end
// of this file (launch_rn_product_* / launch_rn_conv3_*: split-bf16 kernels where the shape allows, else the fp32 MFMA GEMM / TN
// product of gemm.hip); BatchNormalization / ReLU / residual adds are streaming kernels over [pixels][C] with float4 over the
// channels, C any multiple of 32.
//
//   im2col3x3 / col2im3x3     col[p][tap C + c] = y[p + tap][c] ('same' padding)  /  dy[p][c] = sum_taps dcol[p - tap][tap C + c]
//   rn_reduce<BWD>            per-workgroup, per 64-channel chunk [sum z | sum z^2]  resp.  [sum dy' | sum dy' xhat], dy' = dy [mask > 0]
//   rn_bn_finalize / rn_bn_bwd_finalize   the per-channel finalisation of bn_pool.hip for any C (one workgroup per 64 channels)
//   rn_bn_apply / rn_bn_apply2   out = [relu](z scale + shift [+ res])  /  relu(BN(z) + BN_r(z_r)) for a projection block
//   rn_bn_bwd_dz              dz = scale (dy' - c1 - xhat c2)   (GATEZ forms of both backward kernels: the gate recomputed from z)
//   rn_add_masked             dst += dy [mask > 0]
#include "common.h"
#include <algorithm>

#define RN_MAX_PARTIAL 256
int rn_partial_capacity() { return RN_MAX_PARTIAL; }

__device__ __forceinline__ float4 rn_fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// thread per (output pixel, tap, group of 4 channels): G = C / 4 groups
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ y, float* __restrict__ col, int64_t npix, int H, int W, int G) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npix * 9 * G) return;
    const int g = (int)(gid % G);
    const int64_t pt = gid / G;
    const int tap = (int)(pt % 9);
    const int64_t p = pt / 9;
    const int f = (int)(p % W), t = (int)((p / W) % H);
    const int dt = tap / 3 - 1, df = tap % 3 - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t + dt >= 0 && t + dt < H && f + df >= 0 && f + df < W) v = reinterpret_cast<const float4*>(y)[(p + dt * W + df) * G + g];
    reinterpret_cast<float4*>(col)[(p * 9 + tap) * G + g] = v;
}
int launch_im2col3x3(hipStream_t st, const float* y, float* col, int B, int H, int W, int C) {
    const int64_t npix = (int64_t)B * H * W, n = npix * 9 * (C / 4);
    hipLaunchKernelGGL(im2col3x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, col, npix, H, W, C / 4);
    return 0;
}

// dy[p][c] = sum_taps dcol[p - tap][tap C + c]: thread per (pixel, group)
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcol, float* __restrict__ dy, int64_t npix, int H, int W, int G) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npix * G) return;
    const int g = (int)(gid % G);
    const int64_t p = gid / G;
    const int f = (int)(p % W), t = (int)((p / W) % H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dt = tap / 3 - 1, df = tap % 3 - 1;
        if (t - dt >= 0 && t - dt < H && f - df >= 0 && f - df < W) {
            const float4 v = reinterpret_cast<const float4*>(dcol)[((p - dt * W - df) * 9 + tap) * G + g];
            acc = make_float4(acc.x + v.x, acc.y + v.y, acc.z + v.z, acc.w + v.w);
        }
    }
    reinterpret_cast<float4*>(dy)[p * G + g] = acc;
}
int launch_col2im3x3(hipStream_t st, const float* dcol, float* dy, int B, int H, int W, int C) {
    const int64_t npix = (int64_t)B * H * W, n = npix * (C / 4);
    hipLaunchKernelGGL(col2im3x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dcol, dy, npix, H, W, C / 4);
    return 0;
}

// grid (blocks_x, C / 64): workgroup (x, y) reduces the pixels x, x + gridDim.x, ... of channels [64 y, 64 y + 64).
// partial[(y * gridDim.x + x)][128]
// The backward's gate dy' = dy [y > 0], by MASKM: 0 = read from the float tensor `mask` (or none); 1 (no residual behind the
// BatchNorm) = recomputed from z — fma(z, scale, shift) > 0 is bit for bit what rn_bn_apply wrote through its max(., 0); 2 = read
// from the gate bytes rn_bn_apply left beside the block output (one byte per 4 channels, bit j = channel j's output > 0: a
// sixteenth of the float tensor's traffic, and the block output's gate is read three times in the backward)
template <bool BWD, int MASKM = 0>
__global__ __launch_bounds__(256) void rn_reduce_kernel(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ mask,
                                                        const float* __restrict__ coef, float* __restrict__ partial, int64_t npix, int C) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, g = tid & 15, slot = tid >> 4;
    const int c0 = blockIdx.y * 64 + 4 * g;
    const bool gok = c0 < C;                 // C = 32: half of the chunk's groups have no channels
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, mu = s1, is = s1, gsc = s1, gsh = s1;
    if (BWD && gok) { mu = *reinterpret_cast<const float4*>(coef + c0); is = *reinterpret_cast<const float4*>(coef + C + c0); }
    if (MASKM == 1 && gok) { gsc = *reinterpret_cast<const float4*>(coef + 2 * C + c0); gsh = *reinterpret_cast<const float4*>(coef + 3 * C + c0); }
    // one pixel's terms into the running sums — always in ascending pixel order, whatever the number of loads in flight: the same bits
    auto add = [&](const float4& zv, float4 d, unsigned gm, const float4& mf) {
        if (BWD) {
            if (MASKM == 1) {
                const float4 m = rn_fma4(zv, gsc, gsh);
                d = make_float4(m.x > 0.f ? d.x : 0.f, m.y > 0.f ? d.y : 0.f, m.z > 0.f ? d.z : 0.f, m.w > 0.f ? d.w : 0.f);
            } else if (MASKM == 2) {
                d = make_float4((gm & 1) ? d.x : 0.f, (gm & 2) ? d.y : 0.f, (gm & 4) ? d.z : 0.f, (gm & 8) ? d.w : 0.f);
            } else if (mask) {
                d = make_float4(mf.x > 0.f ? d.x : 0.f, mf.y > 0.f ? d.y : 0.f, mf.z > 0.f ? d.z : 0.f, mf.w > 0.f ? d.w : 0.f);
            }
            s1 = make_float4(s1.x + d.x, s1.y + d.y, s1.z + d.z, s1.w + d.w);
            s2 = make_float4(s2.x + d.x * (zv.x - mu.x) * is.x, s2.y + d.y * (zv.y - mu.y) * is.y, s2.z + d.z * (zv.z - mu.z) * is.z,
                             s2.w + d.w * (zv.w - mu.w) * is.w);
        } else {
            s1 = make_float4(s1.x + zv.x, s1.y + zv.y, s1.z + zv.z, s1.w + zv.w);
            s2 = rn_fma4(zv, zv, s2);
        }
    };
    // Four pixels' loads in flight per thread (round 5): with <= 256 workgroups a wave is alone on its SIMD and the one-pixel loop paid a full
    // memory round trip per pixel (the 19.7 MB tensors: 24 us = 1.6 TB/s).
    const int64_t stride = (int64_t)gridDim.x * 16;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t p = (int64_t)blockIdx.x * 16 + slot;
    // (the gate-byte form walks the 78.6 MB block outputs on 512+ workgroups: enough waves per SIMD already, and four pixels in flight cost it 35 -> 43 us)
    for (; MASKM != 2 && gok && p + 3 * stride < npix; p += 4 * stride) {
        float4 zv[4], d[4], mf[4];
        unsigned gm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t e = (p + u * stride) * C + c0;
            zv[u] = *reinterpret_cast<const float4*>(z + e);
            d[u] = BWD ? *reinterpret_cast<const float4*>(dy + e) : zero4;
            gm[u] = (BWD && MASKM == 2) ? reinterpret_cast<const unsigned char*>(mask)[e >> 2] : 0u;
            mf[u] = (BWD && MASKM == 0 && mask) ? *reinterpret_cast<const float4*>(mask + e) : zero4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) add(zv[u], d[u], gm[u], mf[u]);
    }
    for (; gok && p < npix; p += stride) {
        const int64_t e = p * C + c0;
        const float4 zv = *reinterpret_cast<const float4*>(z + e);
        const float4 d = BWD ? *reinterpret_cast<const float4*>(dy + e) : zero4;
        const unsigned gm = (BWD && MASKM == 2) ? reinterpret_cast<const unsigned char*>(mask)[e >> 2] : 0u;
        const float4 mf = (BWD && MASKM == 0 && mask) ? *reinterpret_cast<const float4*>(mask + e) : zero4;
        add(zv, d, gm, mf);
    }
    *reinterpret_cast<float4*>(&red[tid * 8]) = s1;
    *reinterpret_cast<float4*>(&red[tid * 8 + 4]) = s2;
    __syncthreads();
    if (tid < 128) {
        const int kind = tid >> 6, ch = tid & 63, gg = ch >> 2, cc = ch & 3;
        float s = 0.f;
        for (int sl = 0; sl < 16; ++sl) s += red[(sl * 16 + gg) * 8 + kind * 4 + cc];
        partial[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 128 + tid] = s;
    }
}
static int rn_blocks_x(int64_t npix) {
    int64_t b = (npix + 15) / 16;
    return (int)(b > RN_MAX_PARTIAL ? RN_MAX_PARTIAL : b);
}
int launch_rn_bn_stats(hipStream_t st, const float* z, float* partial, int* nbx, int64_t npix, int C) {
    if (C % 32) return -2;
    *nbx = rn_blocks_x(npix);
    hipLaunchKernelGGL((rn_reduce_kernel<false, 0>), dim3(*nbx, (C + 63) / 64), dim3(256), 0, st, z, nullptr, nullptr, nullptr, partial, npix, C);
    return 0;
}
int launch_rn_bn_bwd_reduce(hipStream_t st, const float* z, const float* dy, const float* mask, const float* coef, float* partial, int* nbx,
                            int64_t npix, int C, int gate_z) {
    if (C % 32) return -2;
    *nbx = rn_blocks_x(npix);
    if (gate_z == 1) hipLaunchKernelGGL((rn_reduce_kernel<true, 1>), dim3(*nbx, (C + 63) / 64), dim3(256), 0, st, z, dy, nullptr, coef, partial, npix, C);
    else if (gate_z == 2) hipLaunchKernelGGL((rn_reduce_kernel<true, 2>), dim3(*nbx, (C + 63) / 64), dim3(256), 0, st, z, dy, mask, coef, partial, npix, C);
    else hipLaunchKernelGGL((rn_reduce_kernel<true, 0>), dim3(*nbx, (C + 63) / 64), dim3(256), 0, st, z, dy, mask, coef, partial, npix, C);
    return 0;
}

// The chunk's partials [nbx][128] summed in double by a 1024-thread workgroup: thread (v, part) takes partials part, part + 8, ... on four
// independent accumulators (a lone dependent chain of 128 loads per thread made these two finalisations 33 us launches, 15 % of the
// resnet50_gru step); the assignment is fixed, so the sum is the same bits every run.  Returns the total for v in threads 0..127.
__device__ __forceinline__ double rn_sum_partials(const float* __restrict__ partial, int chunk, int nbx, double* red /* [1024] */) {
    const int tid = threadIdx.x, v = tid & 127, part = tid >> 7;
    const float* p = partial + (size_t)chunk * nbx * 128 + v;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = part;
    // (round 5: the products' epilogues leave up to 1 200 partials per chunk instead of 256 — eight loads in flight per thread, still a fixed order)
    for (; i + 56 < nbx; i += 64) {
        const float a = p[(size_t)i * 128], b = p[(size_t)(i + 8) * 128], c = p[(size_t)(i + 16) * 128], d = p[(size_t)(i + 24) * 128];
        const float e = p[(size_t)(i + 32) * 128], f = p[(size_t)(i + 40) * 128], g = p[(size_t)(i + 48) * 128], h = p[(size_t)(i + 56) * 128];
        s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
        s0 += (double)e; s1 += (double)f; s2 += (double)g; s3 += (double)h;
    }
    for (; i + 24 < nbx; i += 32) {
        const float a = p[(size_t)i * 128], b = p[(size_t)(i + 8) * 128], c = p[(size_t)(i + 16) * 128], d = p[(size_t)(i + 24) * 128];
        s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
    }
    for (; i < nbx; i += 8) s0 += (double)p[(size_t)i * 128];
    red[tid] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    double t = 0.0;
    if (tid < 128) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 128 + tid];
    }
    return t;
}

// one workgroup per 64-channel chunk: the chunk's partials [nbx][128] -> coef = [mean | invstd | scale | shift | c1 | c2] x C.
// Synchronised BatchNorm (seld_set_sync_bn) splits it around the host's all-reduce: phase 1 stops after the chunk's sums
// (sums[chunk][128] doubles), phase 2 starts from the all-reduced sums with the global count; phase 0 does both in one go.
__global__ __launch_bounds__(1024) void rn_bn_finalize_kernel(const float* __restrict__ partial, int nbx, double count, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ mov_mean,
                                                             float* __restrict__ mov_var, float* __restrict__ coef, int C, int training,
                                                             double* __restrict__ sums, int phase) {
    __shared__ double red[1024];
    __shared__ double tot[128];
    const int c0 = blockIdx.x * 64, tid = threadIdx.x;
    if (training && phase != 2) {
        const double t = rn_sum_partials(partial, blockIdx.x, nbx, red);
        if (tid < 128) tot[tid] = t;
        __syncthreads();
        if (phase == 1) {
            if (tid < 128) sums[(size_t)blockIdx.x * 128 + tid] = tot[tid];
            if (tid == 128 && blockIdx.x == 0) sums[(size_t)gridDim.x * 128] = count;     // this rank's element count, all-reduced with the sums
            return;
        }
    }
    if (training && phase == 2) {
        if (tid < 128) tot[tid] = sums[(size_t)blockIdx.x * 128 + tid];
        __syncthreads();
        count = sums[(size_t)gridDim.x * 128];                                             // the global count
    }
    if (tid >= 64) return;
    const int c = c0 + tid;
    if (c >= C) return;
    if (training) {
        const double mean = tot[tid] / count;
        double var = tot[64 + tid] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)SELD_BN_EPS));
        const float sc = gamma[c] * invstd;
        coef[c] = (float)mean; coef[C + c] = invstd; coef[2 * C + c] = sc; coef[3 * C + c] = beta[c] - (float)mean * sc;
        const float f = 1.f - SELD_BN_MOMENTUM;
        const double bessel = count > 1.0 ? count / (count - 1.0) : 1.0;
        mov_mean[c] = mov_mean[c] * (1.f - f) + (float)mean * f;
        mov_var[c] = mov_var[c] * (1.f - f) + (float)(var * bessel) * f;
    } else {
        const float sc = gamma[c] * rsqrtf(mov_var[c] + SELD_BN_EPS);
        coef[2 * C + c] = sc; coef[3 * C + c] = beta[c] - mov_mean[c] * sc;
    }
}
int launch_rn_bn_finalize(hipStream_t st, const float* partial, int nbx, double count, const float* gamma, const float* beta, float* mov_mean,
                          float* mov_var, float* coef, int C, int training, double* sums, int phase) {
    hipLaunchKernelGGL(rn_bn_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, partial, nbx, count, gamma, beta, mov_mean, mov_var, coef, C,
                       training, sums, phase);
    return 0;
}
// backward: [sum dy' | sum dy' xhat] per chunk.  phase 0: dgamma / dbeta and c1 / c2 = the sums / count.  Synchronised BatchNorm:
// phase 1 writes THIS rank's dgamma / dbeta (they are all-reduced with the gradient buffer like every other gradient) and the chunk's
// sums for the host's all-reduce; phase 2 forms c1 / c2 from the global sums and the global count.
__global__ __launch_bounds__(1024) void rn_bn_bwd_finalize_kernel(const float* __restrict__ partial, int nbx, double count, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ coef, int C,
                                                                 double* __restrict__ sums, int phase) {
    __shared__ double red[1024];
    const int c0 = blockIdx.x * 64, tid = threadIdx.x;
    double t;
    if (phase == 2) {
        if (tid >= 128) return;
        t = sums[(size_t)blockIdx.x * 128 + tid];
        count = sums[(size_t)gridDim.x * 128];                                             // the global count
    } else {
        t = rn_sum_partials(partial, blockIdx.x, nbx, red);
        if (tid == 128 && blockIdx.x == 0 && phase == 1) sums[(size_t)gridDim.x * 128] = count;    // this rank's element count
        if (tid >= 128) return;
        if (phase == 1) sums[(size_t)blockIdx.x * 128 + tid] = t;
    }
    if (c0 + (tid & 63) >= C) return;
    if (tid < 64) {
        if (phase != 2) dbeta[c0 + tid] = (float)t;
        if (phase != 1) coef[4 * C + c0 + tid] = (float)(t / count);
    } else {
        if (phase != 2) dgamma[c0 + tid - 64] = (float)t;
        if (phase != 1) coef[5 * C + c0 + tid - 64] = (float)(t / count);
    }
}
int launch_rn_bn_bwd_finalize(hipStream_t st, const float* partial, int nbx, double count, float* dgamma, float* dbeta, float* coef, int C,
                              double* sums, int phase) {
    hipLaunchKernelGGL(rn_bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, partial, nbx, count, dgamma, dbeta, coef, C, sums, phase);
    return 0;
}

// out = [relu](z scale + shift [+ res]); G = C / 4
__global__ __launch_bounds__(256) void rn_bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ coef, const float* __restrict__ res,
                                                          float* __restrict__ out, int64_t n4, int G, int relu, unsigned char* __restrict__ gate4) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int g = (int)(gid % G), C = 4 * G;
    float4 o = rn_fma4(reinterpret_cast<const float4*>(z)[gid], reinterpret_cast<const float4*>(coef + 2 * C)[g],
                       reinterpret_cast<const float4*>(coef + 3 * C)[g]);
    if (res) {
        const float4 r = reinterpret_cast<const float4*>(res)[gid];
        o = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w);
    }
    if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    reinterpret_cast<float4*>(out)[gid] = o;
    if (gate4) gate4[gid] = (unsigned char)((o.x > 0.f) | ((o.y > 0.f) << 1) | ((o.z > 0.f) << 2) | ((o.w > 0.f) << 3));
}
// out = relu(BN(z) + BN_r(zr)): the block output of a projection bottleneck in one pass (both BatchNorms applied on load)
__global__ __launch_bounds__(256) void rn_bn_apply2_kernel(const float* __restrict__ z, const float* __restrict__ coef, const float* __restrict__ zr,
                                                           const float* __restrict__ coef_r, float* __restrict__ out, int64_t n4, int G,
                                                           unsigned char* __restrict__ gate4) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int g = (int)(gid % G), C = 4 * G;
    // the shortcut's value is rounded to fp32 before the add, exactly as when it was written to memory first
    const float4 r = rn_fma4(reinterpret_cast<const float4*>(zr)[gid], reinterpret_cast<const float4*>(coef_r + 2 * C)[g],
                             reinterpret_cast<const float4*>(coef_r + 3 * C)[g]);
    float4 o = rn_fma4(reinterpret_cast<const float4*>(z)[gid], reinterpret_cast<const float4*>(coef + 2 * C)[g],
                       reinterpret_cast<const float4*>(coef + 3 * C)[g]);
    o = make_float4(fmaxf(o.x + r.x, 0.f), fmaxf(o.y + r.y, 0.f), fmaxf(o.z + r.z, 0.f), fmaxf(o.w + r.w, 0.f));
    reinterpret_cast<float4*>(out)[gid] = o;
    if (gate4) gate4[gid] = (unsigned char)((o.x > 0.f) | ((o.y > 0.f) << 1) | ((o.z > 0.f) << 2) | ((o.w > 0.f) << 3));
}
int launch_rn_bn_apply2(hipStream_t st, const float* z, const float* coef, const float* zr, const float* coef_r, float* out, int64_t npix, int C,
                        unsigned char* gate4) {
    const int64_t n4 = npix * (C / 4);
    hipLaunchKernelGGL(rn_bn_apply2_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, z, coef, zr, coef_r, out, n4, C / 4, gate4);
    return 0;
}
int launch_rn_bn_apply(hipStream_t st, const float* z, const float* coef, const float* res, float* out, int64_t npix, int C, int relu,
                       unsigned char* gate4) {
    const int64_t n4 = npix * (C / 4);
    hipLaunchKernelGGL(rn_bn_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, z, coef, res, out, n4, C / 4, relu, gate4);
    return 0;
}

// dz = scale (dy' - c1 - xhat c2), dy' = dy [mask > 0]
template <int MASKM>
__global__ __launch_bounds__(256) void rn_bn_bwd_dz_kernel(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ mask,
                                                           const float* __restrict__ coef, float* __restrict__ dz, int64_t n4, int G) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int g = (int)(gid % G), C = 4 * G;
    const float4 zv = reinterpret_cast<const float4*>(z)[gid];
    float4 d = reinterpret_cast<const float4*>(dy)[gid];
    const float4 mu = reinterpret_cast<const float4*>(coef)[g], is = reinterpret_cast<const float4*>(coef + C)[g];
    const float4 sc = reinterpret_cast<const float4*>(coef + 2 * C)[g], c1 = reinterpret_cast<const float4*>(coef + 4 * C)[g];
    if (MASKM == 1) {
        const float4 m = rn_fma4(zv, sc, reinterpret_cast<const float4*>(coef + 3 * C)[g]);
        d = make_float4(m.x > 0.f ? d.x : 0.f, m.y > 0.f ? d.y : 0.f, m.z > 0.f ? d.z : 0.f, m.w > 0.f ? d.w : 0.f);
    } else if (MASKM == 2) {
        const unsigned m = reinterpret_cast<const unsigned char*>(mask)[gid];
        d = make_float4((m & 1) ? d.x : 0.f, (m & 2) ? d.y : 0.f, (m & 4) ? d.z : 0.f, (m & 8) ? d.w : 0.f);
    } else if (mask) {
        const float4 m = reinterpret_cast<const float4*>(mask)[gid];
        d = make_float4(m.x > 0.f ? d.x : 0.f, m.y > 0.f ? d.y : 0.f, m.z > 0.f ? d.z : 0.f, m.w > 0.f ? d.w : 0.f);
    }
    const float4 c2 = reinterpret_cast<const float4*>(coef + 5 * C)[g];
    float4 o;
    o.x = sc.x * (d.x - c1.x - (zv.x - mu.x) * is.x * c2.x);
    o.y = sc.y * (d.y - c1.y - (zv.y - mu.y) * is.y * c2.y);
    o.z = sc.z * (d.z - c1.z - (zv.z - mu.z) * is.z * c2.z);
    o.w = sc.w * (d.w - c1.w - (zv.w - mu.w) * is.w * c2.w);
    reinterpret_cast<float4*>(dz)[gid] = o;
}
int launch_rn_bn_bwd_dz(hipStream_t st, const float* z, const float* dy, const float* mask, const float* coef, float* dz, int64_t npix, int C,
                        int gate_z) {
    const int64_t n4 = npix * (C / 4);
    if (gate_z == 1) hipLaunchKernelGGL(rn_bn_bwd_dz_kernel<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, z, dy, nullptr, coef, dz, n4, C / 4);
    else if (gate_z == 2) hipLaunchKernelGGL(rn_bn_bwd_dz_kernel<2>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, z, dy, mask, coef, dz, n4, C / 4);
    else hipLaunchKernelGGL(rn_bn_bwd_dz_kernel<0>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, z, dy, mask, coef, dz, n4, C / 4);
    return 0;
}

// dst += dy [mask > 0]   (identity shortcut: the block input receives the gradient behind the block's final ReLU)
__global__ __launch_bounds__(256) void rn_add_masked_kernel(float* __restrict__ dst, const float* __restrict__ dy, const float* __restrict__ mask, int64_t n4) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const float4 d = reinterpret_cast<const float4*>(dy)[gid], m = reinterpret_cast<const float4*>(mask)[gid];
    float4 o = reinterpret_cast<float4*>(dst)[gid];
    o = make_float4(o.x + (m.x > 0.f ? d.x : 0.f), o.y + (m.y > 0.f ? d.y : 0.f), o.z + (m.z > 0.f ? d.z : 0.f), o.w + (m.w > 0.f ? d.w : 0.f));
    reinterpret_cast<float4*>(dst)[gid] = o;
}
__global__ __launch_bounds__(256) void rn_add_gated_kernel(float* __restrict__ dst, const float* __restrict__ dy, const unsigned char* __restrict__ gate4,
                                                           int64_t n4) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const float4 d = reinterpret_cast<const float4*>(dy)[gid];
    const unsigned m = gate4[gid];
    float4 o = reinterpret_cast<float4*>(dst)[gid];
    o = make_float4(o.x + ((m & 1) ? d.x : 0.f), o.y + ((m & 2) ? d.y : 0.f), o.z + ((m & 4) ? d.z : 0.f), o.w + ((m & 8) ? d.w : 0.f));
    reinterpret_cast<float4*>(dst)[gid] = o;
}
int launch_rn_add_gated(hipStream_t st, float* dst, const float* dy, const unsigned char* gate4, int64_t n) {
    hipLaunchKernelGGL(rn_add_gated_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, dst, dy, gate4, n / 4);
    return 0;
}
int launch_rn_add_masked(hipStream_t st, float* dst, const float* dy, const float* mask, int64_t n) {
    hipLaunchKernelGGL(rn_add_masked_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, dst, dy, mask, n / 4);
    return 0;
}

// ---- the products of a resnet50_block convolution ---------------------------------------------------------------------
// A convolution here is a product on [M = B*S*Wout] rows (1x1: the input rows, a frequency stride is a doubled row stride; 3x3: the
// im2col rows).  Where the shape allows, the product runs on the split-bf16 kernels of gemm_sb.hip / gemm_tn_sb.hip (fp32 operands split
// exactly into three bf16 terms, 6 MFMA products, fp32-level error) from weight planes the caller pre-split for this step; every other
// shape (the 32- and 64-channel products of the first two stages) takes the fp32 MFMA GEMM.
int rn_sb_fwd_ok(int K, int N) { return (K % 32) == 0 && (N % 128) == 0; }               // z = A w           [M,K] x [K,N]
int rn_sb_dgrad_ok(int K, int N) { return (N % 32) == 0 && (K % 128) == 0; }             // dA = dz w^T       [M,N] x [N,K]
int rn_sb_wgrad_ok(int K, int N) { return (K % 128) == 0 && (N % 128) == 0; }            // dw = A^T dz       [K,M] x [M,N]

// stat_part (may be null) + nbx: the convolution's BatchNorm statistics leave with the product's epilogue (common.h GemmEpi) — *nbx receives the
// number of [sum | sum of squares] partials per 64-channel chunk that rn_bn_finalize has to fold
int launch_rn_product_fwd(hipStream_t st, const float* A, int lda, const float* w, const unsigned short* wsp, float* z, int M, int K, int N,
                          float* stat_part, int* nbx) {
    GemmEpiScope epi_(stat_part);
    if (wsp && rn_sb_fwd_ok(K, N) && gemm_sb_usable(A, lda, N, K)) {
        if (nbx) *nbx = stat_part ? gemm_epi_row_blocks(M, 1) : 0;
        return launch_gemm_sb(st, A, nullptr, lda, wsp, nullptr, nullptr, nullptr, z, nullptr, N, M, N, K, 0, 0);
    }
    if (nbx) *nbx = stat_part ? gemm_epi_row_blocks(M, 0) : 0;
    return launch_gemm(st, A, lda, w, N, nullptr, z, N, M, N, K, 0, 0, 0);
}
// wsp_t: the planes of w^T (launch_gemm_split_b with transb = 1, K' = N, N' = K)
// addg + gate4 (may be null): dA += addg [gate bit] in the product's epilogue (the identity shortcut's gated gradient: common.h GemmEpi) — the
// split-bf16 kernels only; returns 1 (and adds nothing) where the shape takes the fp32 GEMM: the caller then runs launch_rn_add_gated
int launch_rn_product_dgrad(hipStream_t st, const float* dz, const float* w, const unsigned short* wsp_t, float* dA, int ldd, int M, int K, int N,
                            int accumulate, const float* addg, const unsigned char* gate4) {
    BwdFourScope four_;
    if (wsp_t && rn_sb_dgrad_ok(K, N) && gemm_sb_usable(dz, N, K, N)) {
        // the fused add pays in the 16-wave kernel only (one column group: its epilogue holds four consecutive columns of a row per lane = one
        // float4 of addg and ONE gate byte); in the 4-wave kernel a lane holds 16 rows of one column — 64 dword + 64 byte loads per lane — and the
        // resnet50_gru step was 0.78 ms SLOWER with it than with the separate three-pass add (round 5, same box)
        const bool fuse = addg && K == 128;
        GemmEpiScope epi_(nullptr, fuse ? addg : nullptr, fuse ? gate4 : nullptr);
        const int rc = launch_gemm_sb(st, dz, nullptr, N, wsp_t, nullptr, nullptr, nullptr, dA, nullptr, ldd, M, K, N, 0, 0, accumulate);
        return rc ? rc : (addg && !fuse ? 1 : 0);
    }
    const int rc = launch_gemm(st, dz, N, w, N, nullptr, dA, ldd, M, K, N, 1, 0, accumulate);
    return rc ? rc : (addg ? 1 : 0);
}
// slab: scratch of slab_cap floats for the row splits, combined in a fixed order into dw
int launch_rn_product_wgrad(hipStream_t st, const float* A, int lda, const float* dz, float* slab, int64_t slab_cap, float* dw, int M, int K, int N,
                            int split_bf16) {
    int ns = 0;
    if (split_bf16 && rn_sb_wgrad_ok(K, N) && launch_gemm_tn_sb_tiles(st, A, lda, dz, N, slab, slab_cap, &ns, M, K, N) == 0)
        return launch_reduce_slabs2(st, slab, ns, (int64_t)K * N, dw, (int64_t)K * N, nullptr, 0);
    const int64_t per = (int64_t)K * N + N;
    const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(512, slab_cap / per));
    if (launch_gemm_tn(st, A, lda, dz, N, slab, &ns, M, K, N, 0, 0, 0, splits, 64)) return -1;
    return launch_reduce_slabs2(st, slab, ns, per, dw, (int64_t)K * N, nullptr, 0);
}

// ---- 3x3 'same' convolution of an NHWC image [B*H*W][C] -> [B*H*W][N] as split-bf16 products whose im2col rows are formed on load
// (gemm_sb.hip / gemm_tn_sb.hip: conv_C): no col / dcol tensors, no col2im.  C and N powers of two, multiples of 128 (stages 2-3).
int rn_conv3_sb_ok(int C, int N) { return C >= 128 && N >= 128 && (C & (C - 1)) == 0 && (N & (N - 1)) == 0; }
// wsp: launch_gemm_split_b(w [9 C][N], transb 0)
int launch_rn_conv3_fwd(hipStream_t st, const float* img, const unsigned short* wsp, float* z, int B, int H, int W, int C, int N, float* stat_part,
                        int* nbx) {
    GemmEpiScope epi_(stat_part);
    if (nbx) *nbx = stat_part ? gemm_epi_row_blocks(B * H * W, 1) : 0;
    return launch_gemm_sb(st, img, nullptr, C, wsp, nullptr, nullptr, nullptr, z, nullptr, N, B * H * W, N, 9 * C, 0, 0, 0, C, H, W);
}
// input gradient = the convolution of dz [.][N] with the flipped, channel-swapped kernel; wsp_flip: launch_gemm_split_b(w, ldb N, transb 2, K 9 N, N C)
int launch_rn_conv3_dgrad(hipStream_t st, const float* dz, const unsigned short* wsp_flip, float* dimg, int B, int H, int W, int C, int N) {
    BwdFourScope four_;
    return launch_gemm_sb(st, dz, nullptr, N, wsp_flip, nullptr, nullptr, nullptr, dimg, nullptr, C, B * H * W, C, 9 * N, 0, 0, 0, N, H, W);
}
int launch_rn_conv3_wgrad(hipStream_t st, const float* img, const float* dz, float* slab, int64_t slab_cap, float* dw, int B, int H, int W, int C, int N) {
    int ns = 0;
    if (launch_gemm_tn_sb_tiles(st, img, C, dz, N, slab, slab_cap, &ns, B * H * W, 9 * C, N, C, H, W)) return -1;
    return launch_reduce_slabs2(st, slab, ns, (int64_t)9 * C * N, dw, (int64_t)9 * C * N, nullptr, 0);
}

// ---- stage 0's 3x3 convolution (32 -> 32 channels on 16 frequency bins) on the 64-channel kernels of the conv blocks -------------
// Two neighbouring bins of 32 channels ARE one "super-pixel" of 64 channels in memory ([.., 16, 32] = [.., 8, 64]), and the 3x3
// convolution over pixels is a 3x3 convolution over super-pixels with the kernel
//     W2[dy][DX][32 h' + ci][32 h + co] = w[dy][dx = 2 DX + h' - h][ci][co]   (0 where dx is outside -1..1)
// (input pixel 2 (J + DX) + h' feeds output pixel 2 J + h): half of W2 is zero, i.e. twice the FLOP — on kernels that run five times
// the rate of the fp32 MFMA product on a materialised im2col, and with no im2col / col2im.  The kernel gradient of w is the sum of the
// (two) entries of dW2 that hold each w[dy][dx].
__global__ __launch_bounds__(256) void rn_w32_embed_kernel(const float* __restrict__ w, float* __restrict__ w2) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 9 * 4096) return;
    const int tap = idx >> 12, cp = (idx >> 6) & 63, op = idx & 63;
    const int dyi = tap / 3, DX = tap - 3 * dyi - 1, hp = cp >> 5, ci = cp & 31, h = op >> 5, co = op & 31;
    const int dx = 2 * DX + hp - h;
    w2[idx] = (dx >= -1 && dx <= 1) ? w[((dyi * 3 + dx + 1) * 32 + ci) * 32 + co] : 0.f;
}
__global__ __launch_bounds__(256) void rn_w32_extract_kernel(const float* __restrict__ dw2, float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 9 * 1024) return;
    const int tap = idx >> 10, ci = (idx >> 5) & 31, co = idx & 31;
    const int dyi = tap / 3, dx = tap - 3 * dyi - 1;
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            const int t = dx + h - hp;
            if ((t & 1) == 0 && t >= -2 && t <= 2) s += dw2[((dyi * 3 + t / 2 + 1) * 64 + hp * 32 + ci) * 64 + h * 32 + co];
        }
    dw[idx] = s;
}
int launch_rn_w32_embed(hipStream_t st, const float* w, float* w2) {
    hipLaunchKernelGGL(rn_w32_embed_kernel, dim3(9 * 4096 / 256), dim3(256), 0, st, w, w2);
    return 0;
}
int launch_rn_w32_extract(hipStream_t st, const float* dw2, float* dw) {
    hipLaunchKernelGGL(rn_w32_extract_kernel, dim3(9 * 1024 / 256), dim3(256), 0, st, dw2, dw);
    return 0;
}
