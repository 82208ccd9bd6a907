// Sustained MFMA rate of this card (clock under load included): back-to-back v_mfma_f32_32x32x2_f32 and
// v_mfma_f32_32x32x16_bf16 from registers, 10 independent chains per wave, 2 waves per SIMD, ~0.4 ms per launch.
// hipcc --offload-arch=gfx950 -O3 mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    f32x16 acc[10];
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    bf16x8 ah, bh;
    for (int j = 0; j < 8; ++j) { ah[j] = (__bf16)(a + j); bh[j] = (__bf16)(b - j); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, double flop_per_mfma, float* d) {
    const int grid = 512, iters = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 60; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);   // warm clocks
    (void)hipEventRecord(e0, 0);
    const int n = 50;
    for (int rep = 0; rep < n; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 10 * flop_per_mfma;
    printf("%s: %.4f ms per launch, %.1f TFLOP/s\n", name, ms / n, flops / (ms / n * 1e-3) / 1e12);
}

int main() {
    float* d;
    (void)hipMalloc(&d, 512 * 256 * 4);
    run<0>("f32 32x32x2 ", 2.0 * 32 * 32 * 2, d);
    run<1>("bf16 32x32x16", 2.0 * 32 * 32 * 16, d);
    run<0>("f32 32x32x2 ", 2.0 * 32 * 32 * 2, d);
    return 0;
}
