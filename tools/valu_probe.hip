// fp32 VALU issue rates on gfx950: v_fma_f32 vs v_pk_fma_f32 vs v_exp_f32/v_rcp_f32, cycles per wave64
// instruction per SIMD (1 and 2 waves per SIMD).  Decides whether the GRU recurrence's mat-vec gains from
// packed math.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, int iters) {
    float a = threadIdx.x * 1e-3f;
    float x[16];
    f32x2 y[16];
    for (int i = 0; i < 16; ++i) { x[i] = a + i; y[i] = f32x2{a + i, a - i}; }
    const f32x2 m2 = {1.0001f, 0.9999f}, c2 = {1e-3f, -1e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(1.0001f), "v"(1e-3f));
            else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(m2), "v"(c2));
            else if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
            else asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i]));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, float* d, int threads, int grid = 256) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 100; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e0, 0);
    const int n = 20;
    for (int rep = 0; rep < n; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms / n * 1e-3 * 2.4e9;
    const double instr_per_simd = (double)(threads / 64) / 4 * iters * 16 * (grid / 256);
    printf("%s %d waves/SIMD: %.2f cycles per wave64 instruction per SIMD (at 2.4 GHz)\n", name, threads / 256 * (grid / 256), cyc / instr_per_simd);
}
int main() {
    float* d;
    (void)hipMalloc(&d, 512 * 1024 * 4);
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32   ", d, threads);
        run<1>("v_pk_fma_f32", d, threads);
        run<2>("v_exp_f32   ", d, threads);
        run<3>("v_rcp_f32   ", d, threads);
    }
    run<0>("v_fma_f32   ", d, 1024, 512);
    run<1>("v_pk_fma_f32", d, 1024, 512);
    run<2>("v_exp_f32   ", d, 1024, 512);
    return 0;
}
