// Dependent-chain latency of the VALU instructions in the GRU recurrence's gate tail (gfx950): one wave issues N dependent copies of
// an instruction; cycles per instruction = issue-to-issue distance of DEPENDENT instructions.  1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_latency_probe.hip -o /tmp/vlat && /tmp/vlat
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)
template <int MODE>
__global__ __launch_bounds__(512) void lat(float* out, unsigned long long* cyc) {
    float x = threadIdx.x * 1e-3f + 1.f, y = 1.0001f, z = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p = {x, y}, q = {1.0001f, 0.9999f}, r2 = {1e-3f, 1e-3f};
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < 16; ++it) {
        if (MODE == 0) { REP64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));) }
        if (MODE == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));) }
        if (MODE == 2) { REP64(asm volatile("v_exp_f32 %0, %0" : "+v"(x));) }
        if (MODE == 3) { REP64(asm volatile("v_rcp_f32 %0, %0" : "+v"(x));) }
        if (MODE == 4) { REP64(asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));) }
        if (MODE == 5) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(q), "v"(r2));) }
        if (MODE == 6) { REP64(asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));) }
        if (MODE == 7) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));) }
        if (MODE == 8) { REP64(asm volatile("v_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0" : "+v"(x));) }   // 3 instructions per copy
        if (MODE == 9) { REP64(asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(x), "+v"(z) : "v"(y));) }   // two independent chains, 2 per copy
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + z + p.x + p.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> static void run(const char* name, int per_copy) {
    float* d; unsigned long long* c;
    (void)hipMalloc(&d, 64 * 1024 * 4); (void)hipMalloc(&c, 8);
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((lat<MODE>), dim3(64), dim3(threads), 0, 0, d, c);
        unsigned long long h = 0;
        (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%-52s %d wave(s)/SIMD: %6.2f cycles per instruction\n", name, threads / 256, (double)h / (16.0 * 64 * per_copy));
    }
    (void)hipFree(d); (void)hipFree(c);
}
int main() {
    run<0>("v_add_f32, dependent", 1);
    run<9>("v_add_f32, two independent chains", 2);
    run<1>("v_fma_f32, dependent", 1);
    run<5>("v_pk_fma_f32, dependent", 1);
    run<2>("v_exp_f32, dependent", 1);
    run<3>("v_rcp_f32, dependent", 1);
    run<8>("exp, add, rcp dependent", 3);
    run<4>("s_nop 1 + v_add_f32_dpp quad_perm, dependent", 1);
    run<6>("s_nop 1 + v_permlane32_swap, dependent", 1);
    run<7>("v_cndmask_b32, dependent", 1);
    return 0;
}
