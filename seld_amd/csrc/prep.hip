// prep.hip — all per-step weight pre-passes in ONE launch: each of the three stand-alone kernels is a few microseconds of
// mostly ramp-up, and they are independent (they only read the weights).  Block row blockIdx.y selects the job:
// [0, na) GEMM weight splits, [na, na + nb) conv weight splits, na + nb: the folded head weights (4 rows of Weff per block).
#include "prep.h"

__global__ __launch_bounds__(256) void weight_prep_kernel(GemmSplitJobs a, int na, SplitWeightJobs b, int nb, HeadsLin h,
                                                          float* __restrict__ weff) {
    const int y = blockIdx.y;
    if (y < na) gemm_split_b_body(a, y, blockIdx.x, gridDim.x);
    else if (y < na + nb) split_weights_body(b, y - na, blockIdx.x);
    else if (weff) heads_weff_body(h, weff, 4 * blockIdx.x + (threadIdx.x >> 6), threadIdx.x & 63);
}

int launch_weight_prep(hipStream_t st, const GemmSplitJobs& a, int na, const SplitWeightJobs& b, int nb, const HeadsLin& h, float* weff) {
    if (na < 0 || na > GSB_MAX_JOBS || nb < 0 || nb > 8) return -1;
    if (weff && (h.n[0] + h.n[1] > 64 || h.K + 1 > 4 * 144)) return -1;
    const int ny = na + nb + (weff ? 1 : 0);
    if (ny == 0) return 0;
    // 144 block columns: what a conv weight tensor needs (9 * 4096 / 256); the GEMM splits stride over them
    hipLaunchKernelGGL(weight_prep_kernel, dim3(144, ny), dim3(256), 0, st, a, na, b, nb, h, weff);
    return 0;
}
