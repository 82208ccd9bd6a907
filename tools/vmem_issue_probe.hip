// Per-instruction cost of vector memory instructions on one CU: every wave rewrites / rereads its own 16 KB
// (L2/L1-resident), with dword, dwordx2 or dwordx4 per lane.  If time tracks the instruction count and not
// the bytes, the address path (not bandwidth) bounds kernels that move accumulator tiles one dword at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
template <int VEC, bool LOAD>
__global__ __launch_bounds__(256) void probe(float* buf, int iters, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = buf + ((size_t)blockIdx.x * 4 + wave) * 4096;      // 16 KB per wave
    float acc = 0.f, lv1 = 0.f, sv1 = 1.f;
    v2 lv2 = {0.f, 0.f}, sv2 = {1.f, 2.f};
    v4 lv4 = {0.f, 0.f, 0.f, 0.f}, sv4 = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64 / VEC; ++i) {                          // 64 floats per lane per iteration
            float* p = base + (i * 64 + lane) * VEC;
            if (LOAD) {
                if (VEC == 1) { float v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); lv1 = v; }
                else if (VEC == 2) { v2 v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); lv2 = v; }
                else { v4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); lv4 = v; }
            } else {
                if (VEC == 1) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(sv1) : "memory");
                else if (VEC == 2) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(sv2) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(sv4) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    acc = lv1 + lv2.x + lv4.x;
    if (LOAD) sink[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int VEC, bool LOAD>
static void run(const char* name, float* d, float* sink) {
    const int grid = 512, iters = 200;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL((probe<VEC, LOAD>), dim3(grid), dim3(256), 0, 0, d, iters, sink);
    (void)hipEventRecord(e0, 0);
    const int n = 10;
    for (int rep = 0; rep < n; ++rep) hipLaunchKernelGGL((probe<VEC, LOAD>), dim3(grid), dim3(256), 0, 0, d, iters, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_cu = (double)grid / 256 * 4 * iters * (64 / VEC);
    const double cyc = ms / n * 1e-3 * 2.4e9;
    printf("%s: %.4f ms, %.1f cycles (at 2.4 GHz) per wave instruction per CU, %.2f TB/s\n", name, ms / n, cyc / instr_per_cu,
           (double)grid * 4 * iters * 64 * 64 * 4 / (ms / n * 1e-3) / 1e12);
}
int main() {
    float *d, *sink;
    (void)hipMalloc(&d, (size_t)512 * 4 * 4096 * 4);
    (void)hipMalloc(&sink, 512 * 256 * 4);
    run<1, false>("store dword  ", d, sink);
    run<2, false>("store dwordx2", d, sink);
    run<4, false>("store dwordx4", d, sink);
    run<1, true>("load  dword  ", d, sink);
    run<2, true>("load  dwordx2", d, sink);
    run<4, true>("load  dwordx4", d, sink);
    return 0;
}
