// metrics.hip — on-device SELD metrics of metrics.SELDMetrics.update_states (metrics.py:60-154): block-wise
// (10 label frames) location-sensitive detection and class-sensitive localisation counters.  The
// reference updates them in TF eager mode on the host after EVERY train step (train.py:82-83, a
// per-step device sync); here one small kernel accumulates into a device-resident state vector.
//
// state layout (doubles): [0]TP [1]FP [2]TN [3]FN [4]S [5]D [6]I [7]Nref [8]Nsys [9]total_DE [10]DE_TP
//                         [11 + 0*nc ..] class_tp, [11 + nc ..] class_fp, [11 + 2nc ..] class_tn, [11 + 3nc ..] class_fn
#include "common.h"
#include "../../include/seld_hip.h"

namespace {

#define MET_SCALARS 11
#define MET_MAX_BLOCK 32

// one thread per (batch row, block of frames): all classes, all frames of the block
__global__ __launch_bounds__(128) void seld_metrics_items_kernel(const float* __restrict__ sed_true, const float* __restrict__ doa_true,
                                                                 const float* __restrict__ sed_pred, const float* __restrict__ doa_pred,
                                                                 float* __restrict__ items, int B, int S, int nc, int block_size,
                                                                 float doa_threshold) {
    const int nblk = (S + block_size - 1) / block_size;
    const int it = blockIdx.x * 128 + threadIdx.x;
    if (it >= B * nblk) return;
    const int b = it / nblk, blk = it - b * nblk;
    const int f0 = blk * block_size, f1 = min(S, f0 + block_size);
    const int ncol = MET_SCALARS + 4 * nc;
    float* out = items + (size_t)it * ncol;
    float acc[MET_SCALARS];
#pragma unroll
    for (int i = 0; i < MET_SCALARS; ++i) acc[i] = 0.f;
    float loc_fn = 0.f, loc_fp = 0.f;
    for (int c = 0; c < nc; ++c) {
        float t = 0.f, p = 0.f;                        // class present in the block (reduce_max over frames)
        for (int f = f0; f < f1; ++f) {
            const size_t i = ((size_t)b * S + f) * nc + c;
            t = fmaxf(t, sed_true[i]);
            p = fmaxf(p, sed_pred[i] > 0.5f ? 1.f : 0.f);
        }
        acc[7] += t;
        acc[8] += p;
        const float fn = t * (1.f - p), fp = (1.f - t) * p, tn = (1.f - t) * (1.f - p), tp = t * p;
        acc[2] += tn;
        out[MET_SCALARS + c] = tp;
        out[MET_SCALARS + nc + c] = fp;
        out[MET_SCALARS + 2 * nc + c] = tn;
        out[MET_SCALARS + 3 * nc + c] = fn;
        acc[3] += fn;
        acc[1] += fp;
        loc_fn += fn;
        loc_fp += fp;
        // frames where the class is active in both reference and prediction
        float matched = 0.f, ang_sum = 0.f;
        for (int f = f0; f < f1; ++f) {
            const size_t i = ((size_t)b * S + f) * nc + c;
            const float fm = (sed_true[i] * tp) * ((sed_pred[i] > 0.5f ? 1.f : 0.f) * tp);
            matched += fm;
            const float* dt = doa_true + ((size_t)b * S + f) * 3 * nc + c;
            const float* dp = doa_pred + ((size_t)b * S + f) * 3 * nc + c;
            float a[3], q[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { a[k] = dt[k * nc] * fm; q[k] = dp[k * nc] * fm; }
            // tf.math.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12))
            const float na = rsqrtf(fmaxf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2], 1e-12f));
            const float nq = rsqrtf(fmaxf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2], 1e-12f));
            float sa = 0.f, sq = 0.f, dot = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { a[k] *= na; q[k] *= nq; sa += a[k]; sq += q[k]; dot += a[k] * q[k]; }
            const float zeros = (sa == 0.f ? 1.f : 0.f) * (sq == 0.f ? 1.f : 0.f);
            dot = fminf(fmaxf(dot, -1.f), 1.f);
            ang_sum += acosf(dot) / 3.14159265358979323846f * 180.f * (1.f - zeros);
        }
        const float exist = matched > 0.f ? 1.f : 0.f;
        acc[10] += exist;
        const float fn2 = tp * (1.f - exist);
        acc[3] += fn2;
        loc_fn += fn2;
        const float avg = ang_sum / fmaxf(matched, 1e-8f);
        acc[9] += avg;
        const float close = avg <= doa_threshold ? 1.f : 0.f;
        acc[0] += close * exist;
        const float fn3 = (1.f - close) * exist;
        acc[3] += fn3;
        loc_fn += fn3;
    }
    acc[4] = fminf(loc_fp, loc_fn);
    acc[5] = fmaxf(0.f, loc_fn - loc_fp);
    acc[6] = fmaxf(0.f, loc_fp - loc_fn);
#pragma unroll
    for (int i = 0; i < MET_SCALARS; ++i) out[i] = acc[i];
}

// state[col] += sum_rows items[row][col]   (one block per column, fixed order)
__global__ __launch_bounds__(256) void seld_metrics_reduce_kernel(const float* __restrict__ items, int rows, int ncol,
                                                                  double* __restrict__ state) {
    __shared__ double red[256];
    const int col = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)items[(size_t)r * ncol + col];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) state[col] += red[0];
}

}  // namespace

extern "C" {

int seld_metrics_state_size(int n_classes) { return MET_SCALARS + 4 * n_classes; }

int64_t seld_metrics_scratch_floats(int B, int S, int n_classes, int block_size) {
    if (block_size <= 0) return -1;
    return (int64_t)B * ((S + block_size - 1) / block_size) * (MET_SCALARS + 4 * n_classes);
}

int seld_metrics_update(const float* sed_true, const float* doa_true, const float* sed_pred, const float* doa_pred, int B, int S,
                        int n_classes, int block_size, float doa_threshold, double* state, float* scratch, void* stream) {
    if (!sed_true || !doa_true || !sed_pred || !doa_pred || !state || !scratch) return SELD_ERR_INVALID;
    if (B <= 0 || S <= 0 || n_classes <= 0 || block_size <= 0 || block_size > MET_MAX_BLOCK) return SELD_ERR_INVALID;
    const int items = B * ((S + block_size - 1) / block_size);
    const int ncol = MET_SCALARS + 4 * n_classes;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(seld_metrics_items_kernel, dim3((items + 127) / 128), dim3(128), 0, st, sed_true, doa_true, sed_pred, doa_pred,
                       scratch, B, S, n_classes, block_size, doa_threshold);
    hipLaunchKernelGGL(seld_metrics_reduce_kernel, dim3(ncol), dim3(256), 0, st, scratch, items, ncol, state);
    return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP;
}

}  // extern "C"
