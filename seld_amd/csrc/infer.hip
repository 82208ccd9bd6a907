// infer.hip — sliding-window inference helpers of evaluator.ensemble_outputs / trainv2.ensemble_outputs
// (evaluator.py:16-50, trainv2.py:158-192): tf.signal.frame(x, win, step) and the overlap-and-add average
// of the per-window outputs (frame_step 1 on the label axis, divided by the per-frame window count).
#include "common.h"
#include "../../include/seld_hip.h"

namespace {

// windows[w][i][:] = x[(w0 + w)*step + i][:], row = FC floats (FC % 4 == 0)
__global__ __launch_bounds__(256) void frame_windows_kernel(const float* __restrict__ x, float* __restrict__ win, int FC4,
                                                            int win_size, int step, int w0, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)(i % FC4);
    const int64_t r = i / FC4;
    const int row = (int)(r % win_size);
    const int w = (int)(r / win_size);
    reinterpret_cast<float4*>(win)[i] = reinterpret_cast<const float4*>(x)[((int64_t)(w0 + w) * step + row) * FC4 + c];
}

// out[t][d] = mean over windows w with 0 <= t - w < L of y[w][t - w][d]
__global__ __launch_bounds__(256) void overlap_average_kernel(const float* __restrict__ y, float* __restrict__ out, int n_win,
                                                              int L, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int T = n_win - 1 + L;
    if (i >= (int64_t)T * D) return;
    const int d = (int)(i % D), t = (int)(i / D);
    const int wlo = max(0, t - L + 1), whi = min(n_win - 1, t);
    float s = 0.f;
    int w = wlo;
    for (; w + 7 <= whi; w += 8) {           // eight loads in flight, added in window order (one dependent load per window took 28 us for 21 600 outputs)
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = y[((size_t)(w + k) * L + (t - w - k)) * D + d];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; w <= whi; ++w) s += y[((size_t)w * L + (t - w)) * D + d];
    out[i] = s / (float)(whi - wlo + 1);
}

}  // namespace

extern "C" {

int seld_frame_windows(const float* x, float* windows, int T, int FC, int win_size, int step, int first_window, int n_windows,
                       void* stream) {
    if (!x || !windows || T <= 0 || FC <= 0 || (FC & 3) || win_size <= 0 || step <= 0 || first_window < 0 || n_windows <= 0)
        return SELD_ERR_INVALID;
    if ((int64_t)(first_window + n_windows - 1) * step + win_size > T) return SELD_ERR_INVALID;
    const int64_t total4 = (int64_t)n_windows * win_size * (FC / 4);
    hipLaunchKernelGGL(frame_windows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, windows,
                       FC / 4, win_size, step, first_window, total4);
    return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP;
}

int seld_overlap_average(const float* y, float* out, int n_windows, int L, int D, void* stream) {
    if (!y || !out || n_windows <= 0 || L <= 0 || D <= 0) return SELD_ERR_INVALID;
    const int64_t n = (int64_t)(n_windows - 1 + L) * D;
    hipLaunchKernelGGL(overlap_average_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, out,
                       n_windows, L, D);
    return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP;
}

}  // extern "C"
