// How long does one GRU-like step take on ONE CU when its 49 152 MACs are spread over 8 or 16 waves?  Each iteration: every wave issues
// its share of the mat-vec as v_fma_f32 (or v_pk_fma_f32) on 12 independent accumulators, a short dependent tail (DPP adds, exp, rcp)
// and a workgroup barrier.  Prints cycles per iteration (s_memtime) for 512 / 1024 threads.
//   hipcc --offload-arch=gfx950 -O3 tools/gru_issue_probe.hip -o /tmp/gip && /tmp/gip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NFMA, bool PK>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, int iters) {
    float a = threadIdx.x * 1e-3f;
    float x[12];
    f32x2 y[12];
    for (int i = 0; i < 12; ++i) { x[i] = a + i; y[i] = f32x2{a + i, a - i}; }
    const f32x2 m2 = {1.0001f, 0.9999f}, c2 = {1e-3f, -1e-3f};
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NFMA; ++i) {
            if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i % 12]) : "v"(m2), "v"(c2));
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i % 12]) : "v"(1.0001f), "v"(1e-3f));
        }
        // a GRU-like dependent tail: 3 cross-lane adds, one exp, one rcp, a few fmas
        float s = PK ? y[0].x + y[1].y : x[0] + x[1];
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));
        s = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-s));
        x[2] = fmaf(s, x[3], x[2]); y[2].x = x[2];
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float r = 0.f;
    for (int i = 0; i < 12; ++i) r += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NFMA, bool PK>
static void run(const char* name, int threads) {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 64 * 1024 * 4); (void)hipMalloc(&c, 8);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<NFMA, PK>), dim3(64), dim3(threads), 0, 0, d, c, iters);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-52s %4d threads: %7.1f memtime ticks per step (100 MHz ticks x 24 = cycles at 2.4 GHz: %6.0f)\n", name, threads, (double)h / iters, (double)h / iters * 24.0);
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    run<96, false>("96 v_fma_f32 per wave (8 waves)", 512);
    run<48, true>("48 v_pk_fma_f32 per wave (8 waves)", 512);
    run<48, false>("48 v_fma_f32 per wave (16 waves)", 1024);
    run<24, true>("24 v_pk_fma_f32 per wave (16 waves)", 1024);
    run<192, false>("192 v_fma_f32 per wave (4 waves)", 256);
    return 0;
}
