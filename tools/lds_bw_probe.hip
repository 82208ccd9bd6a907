// LDS read bandwidth per CU for ds_read_b32 / b64 / b128 (conflict-free, lane-contiguous), 4 and 8 waves per CU.
// Decides how many operand bytes per MFMA an LDS-fed kernel can afford.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));
template <int VEC>
__global__ void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const unsigned addr = (unsigned)(threadIdx.x & 63) * VEC * 4 + (threadIdx.x >> 6) * 1024 * 0;   // byte address, lane-contiguous
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (VEC == 1) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(0)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += 0.f * 0.f; (void)v; }
            else if (VEC == 2) { v2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); (void)v; }
            else { v4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); (void)v; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int VEC>
static void run(const char* name, float* d, int threads) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(probe<VEC>, dim3(grid), dim3(threads), 65536, 0, d, iters);
    (void)hipEventRecord(e0, 0);
    const int n = 10;
    for (int rep = 0; rep < n; ++rep) hipLaunchKernelGGL(probe<VEC>, dim3(grid), dim3(threads), 65536, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)(threads / 64) * iters * 16 * 64 * VEC * 4;
    const double cyc = ms / n * 1e-3 * 2.4e9;
    printf("%s %d waves/CU: %.1f B/clk/CU (at 2.4 GHz), %.1f cycles per wave instruction per CU\n", name, threads / 64, bytes_per_cu / cyc,
           cyc / ((double)(threads / 64) * iters * 16));
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 1024 * 4);
    for (int t : {256, 512}) {
        run<1>("ds_read_b32 ", d, t);
        run<2>("ds_read_b64 ", d, t);
        run<4>("ds_read_b128", d, t);
    }
    return 0;
}
