// Facts the row-broadcast GRU recurrence rests on, measured on the box:
//  (1) semantics of `row_newbcast:n`, v_permlane16_swap, v_permlane32_swap (printed for one wave);
//  (2) issue rate of v_fmac_f32_dpp row_newbcast against plain v_fmac_f32 / v_pk_fma_f32, 8 waves per CU (two per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_probe.hip -o /tmp/dpp_probe && /tmp/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void sem(float* o) {
    const int l = threadIdx.x;
    float acc = 0.f, h = (float)l, one = 1.f;
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(h), "v"(one));
    o[l] = acc;
    unsigned a = 100 + l, b = 200 + l;
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[64 + l] = (float)s[0]; o[128 + l] = (float)s[1];
    auto t = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[192 + l] = (float)t[0]; o[256 + l] = (float)t[1];
}
template <int MODE>
__global__ __launch_bounds__(512) void rate(float* out, unsigned long long* cyc, int iters) {
    float a = threadIdx.x * 1e-3f;
    float x[6];
    f32x2 y[6];
    for (int i = 0; i < 6; ++i) { x[i] = a + i; y[i] = f32x2{a + i, a - i}; }
    const f32x2 m2 = {1.0001f, 0.9999f}, c2 = {1e-3f, -1e-3f};
    float hh = a, uu = 1.0001f;
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 96; ++i) {
            if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i % 6]) : "v"(hh), "v"(uu));
            if (MODE == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[i % 6]) : "v"(hh), "v"(uu));
            if (MODE == 2 && i < 48) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i % 6]) : "v"(m2), "v"(c2));
            if (MODE == 3) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i % 6]) : "v"(hh), "v"(uu));
        }
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float r = 0.f;
    for (int i = 0; i < 6; ++i) r += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> static void run(const char* name, int threads) {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 64 * 1024 * 4); (void)hipMalloc(&c, 8);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((rate<MODE>), dim3(64), dim3(threads), 0, 0, d, c, iters);
    unsigned long long h = 0;
    (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-44s %4d threads: %7.1f cycles per iteration (+ barrier)\n", name, threads, (double)h / iters);
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    float* d; (void)hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d);
    float h[320]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[5] = {"fmac row_newbcast:5 of lane id", "permlane16_swap vdst(100+l)", "permlane16_swap src(200+l)", "permlane32_swap vdst(100+l)", "permlane32_swap src(200+l)"};
    for (int k = 0; k < 5; ++k) { printf("%s:\n ", nm[k]); for (int l = 0; l < 64; ++l) printf("%g%s", h[64 * k + l], l % 16 == 15 ? "\n " : " "); printf("\n"); }
    run<0>("96 v_fmac_f32 per wave", 512);
    run<1>("96 v_fmac_f32_dpp row_newbcast per wave", 512);
    run<3>("96 v_fmac_f32_dpp quad_perm per wave", 512);
    run<2>("48 v_pk_fma_f32 per wave", 512);
    run<0>("96 v_fmac_f32 per wave", 256);
    run<1>("96 v_fmac_f32_dpp row_newbcast per wave", 256);
    run<0>("96 v_fmac_f32 per wave", 1024);
    run<1>("96 v_fmac_f32_dpp row_newbcast per wave", 1024);
    return 0;
}
