// augment.hip — the batch augmentations of the reference's training pipeline (train.py:157-165), on the device so
// that a batch goes host -> HBM once and is augmented where the step reads it:
//   aug_mask          transforms.mask (transforms.py:6-44): per sample and per `period`-frame segment, zero a random
//                     run of frames (axis -3) and/or of frequency bins (axis -2); the random draws come from the host
//   aug_gather_sign   the channel shuffles with sign flips of foa_intensity_vec_aug / acs_aug (transforms.py:73-114,
//                     159-207): out[b, o, r, i] = sgn[b, r] * in[b, o, src[b, r], i], in place
// Pure data movement: results are bit-identical to the numpy restatement (oracle/transforms_oracle.py).
#include "common.h"
#include "../../include/seld_hip.h"

namespace {

// x [B, T, F, C]; segment s = t / period of sample b: frames [t_off, t_off + t_size) of the segment and bins
// [f_off, f_off + f_size) (for every frame of the segment) are zeroed.  One thread per (b, t, f): C contiguous floats.
__global__ __launch_bounds__(256) void aug_mask_kernel(float* __restrict__ x, int T, int F, int C, int period, int nseg,
                                                       const int* __restrict__ t_off, const int* __restrict__ t_size,
                                                       const int* __restrict__ f_off, const int* __restrict__ f_size,
                                                       int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int f = (int)(i % F);
    const int64_t bt = i / F;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const int seg = b * nseg + t / period, tt = t % period;
    bool kill = false;
    if (t_off) kill = tt >= t_off[seg] && tt < t_off[seg] + t_size[seg];
    if (f_off) kill = kill || (f >= f_off[seg] && f < f_off[seg] + f_size[seg]);
    if (kill) {
        float* p = x + i * C;
        for (int c = 0; c < C; ++c) p[c] = 0.f;
    }
}

#define AUG_MAX_R 32
// one thread per (b, o, i): reads its R values, writes them back permuted and signed
__global__ __launch_bounds__(256) void aug_gather_sign_kernel(float* __restrict__ x, int64_t outer, int R, int64_t inner,
                                                              const int* __restrict__ src, const float* __restrict__ sgn,
                                                              int64_t n) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n) return;
    const int64_t i = g % inner, bo = g / inner;
    const int b = (int)(bo / outer);
    float* p = x + bo * R * inner + i;
    float v[AUG_MAX_R];
#pragma unroll
    for (int r = 0; r < AUG_MAX_R; ++r)
        if (r < R) v[r] = p[(int64_t)r * inner];
#pragma unroll
    for (int r = 0; r < AUG_MAX_R; ++r)
        if (r < R) {
            const int s = src[b * R + r];
            float val = 0.f;
#pragma unroll
            for (int q = 0; q < AUG_MAX_R; ++q) val = (q == s) ? v[q] : val;     // register select: no dynamic indexing
            p[(int64_t)r * inner] = sgn[b * R + r] * val;
        }
}

}  // namespace

extern "C" {

int seld_aug_mask(float* x, int B, int T, int F, int C, int period, const int* t_off, const int* t_size, const int* f_off,
                  const int* f_size, void* stream) {
    if (!x || B <= 0 || T <= 0 || F <= 0 || C <= 0 || period <= 0) return SELD_ERR_INVALID;
    if (T % period) return SELD_ERR_INVALID;                 /* transforms.py:39-40: ValueError */
    if ((t_off == nullptr) != (t_size == nullptr) || (f_off == nullptr) != (f_size == nullptr)) return SELD_ERR_INVALID;
    if (!t_off && !f_off) return SELD_OK;
    const int64_t n = (int64_t)B * T * F;
    hipLaunchKernelGGL(aug_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, T, F, C, period,
                       T / period, t_off, t_size, f_off, f_size, n);
    return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP;
}

int seld_aug_gather_sign(float* x, int B, int64_t outer, int R, int64_t inner, const int* src, const float* sgn, void* stream) {
    if (!x || !src || !sgn || B <= 0 || outer <= 0 || inner <= 0 || R <= 0) return SELD_ERR_INVALID;
    if (R > AUG_MAX_R) return SELD_ERR_UNSUPPORTED;
    const int64_t n = (int64_t)B * outer * inner;
    hipLaunchKernelGGL(aug_gather_sign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, outer, R,
                       inner, src, sgn, n);
    return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP;
}

}  // extern "C"
