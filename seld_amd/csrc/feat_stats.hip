// feat_stats.hip — feature_extractor.calculate_statistics on the device (feature_extractor.py:218-224): per-(freq, chan) mean and
// population standard deviation over ALL frames of a list of feature files, the statistics apply_normalizer (:226-234) uses.
// The reference concatenates every file on the host and calls numpy's mean / std; here a file (or a batch of files) already in HBM
// is folded into a device-resident accumulator [sum | sum of squares | row count] of doubles, one call per tensor, and a finalise
// kernel turns the accumulator into float mean / std.  HBM-bound: one read of the features (5.4 MB per 60-s FOA clip).
// Deterministic: rows are split into at most FST_MAX_BLOCKS contiguous chunks, each chunk's sums are taken in row order in double,
// and the chunk partials are added in chunk order — no atomics; the same calls in the same order give the same bits.
#include "common.h"
#include "../../include/seld_hip.h"

namespace {

#define FST_MAX_BLOCKS 512
#define FST_THREADS 256

// block b sums rows [r0, r1) of feat [rows][FC]; thread t owns columns t, t + 256, ... (consecutive threads = consecutive floats)
__global__ __launch_bounds__(FST_THREADS) void feat_stats_partial_kernel(const float* __restrict__ feat, int64_t rows, int FC,
                                                                         double* __restrict__ partial) {
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(rows, r0 + per);
    for (int c = threadIdx.x; c < FC; c += FST_THREADS) {
        double s1 = 0.0, s2 = 0.0;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {      // four loads in flight; the sums stay in row order
            const float v0 = feat[r * FC + c], v1 = feat[(r + 1) * FC + c], v2 = feat[(r + 2) * FC + c], v3 = feat[(r + 3) * FC + c];
            s1 += v0; s2 += (double)v0 * v0;
            s1 += v1; s2 += (double)v1 * v1;
            s1 += v2; s2 += (double)v2 * v2;
            s1 += v3; s2 += (double)v3 * v3;
        }
        for (; r < r1; ++r) {
            const float v = feat[r * FC + c];
            s1 += v; s2 += (double)v * v;
        }
        partial[(size_t)blockIdx.x * 2 * FC + c] = s1;
        partial[(size_t)blockIdx.x * 2 * FC + FC + c] = s2;
    }
}

__global__ __launch_bounds__(FST_THREADS) void feat_stats_fold_kernel(const double* __restrict__ partial, int nblk, int FC, int64_t rows,
                                                                      double* __restrict__ acc) {
    const int c = blockIdx.x * FST_THREADS + threadIdx.x;
    if (c < 2 * FC) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * 2 * FC + c];
        acc[c] += s;
    }
    if (c == 0) acc[2 * FC] += (double)rows;
}

__global__ __launch_bounds__(FST_THREADS) void feat_stats_finalize_kernel(const double* __restrict__ acc, int FC, float* __restrict__ mean,
                                                                          float* __restrict__ stdv) {
    const int c = blockIdx.x * FST_THREADS + threadIdx.x;
    if (c >= FC) return;
    const double n = acc[2 * FC];
    const double m = n > 0 ? acc[c] / n : 0.0;
    const double var = n > 0 ? acc[FC + c] / n - m * m : 0.0;     // numpy's std: population variance (ddof = 0)
    mean[c] = (float)m;
    stdv[c] = (float)sqrt(var > 0 ? var : 0.0);
}

}  // namespace

extern "C" {

int64_t seld_feat_stats_scratch_doubles(int FC) { return FC > 0 ? (int64_t)FST_MAX_BLOCKS * 2 * FC : -1; }

int seld_feat_stats_accumulate(const float* feat, int64_t rows, int FC, double* acc, double* scratch, void* stream) {
    if (!feat || !acc || !scratch || rows < 0 || FC < 1) return SELD_ERR_INVALID;
    if (rows == 0) return SELD_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t want = (rows + 15) / 16;
    const int nblk = (int)(want < FST_MAX_BLOCKS ? want : FST_MAX_BLOCKS);
    hipLaunchKernelGGL(feat_stats_partial_kernel, dim3(nblk), dim3(FST_THREADS), 0, st, feat, rows, FC, scratch);
    hipLaunchKernelGGL(feat_stats_fold_kernel, dim3((2 * FC + FST_THREADS - 1) / FST_THREADS), dim3(FST_THREADS), 0, st, scratch, nblk, FC,
                       rows, acc);
    return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP;
}

int seld_feat_stats_finalize(const double* acc, int FC, float* mean, float* stdv, void* stream) {
    if (!acc || !mean || !stdv || FC < 1) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(feat_stats_finalize_kernel, dim3((FC + FST_THREADS - 1) / FST_THREADS), dim3(FST_THREADS), 0,
                       static_cast<hipStream_t>(stream), acc, FC, mean, stdv);
    return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP;
}

}  // extern "C"
