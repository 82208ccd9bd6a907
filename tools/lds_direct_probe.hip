// Does __builtin_amdgcn_global_load_lds (global_load_lds_dwordx4, gfx950) land data where the kernels need it?
// Each wave copies 64 float4 from global straight into LDS at a wave-uniform base (+ lane * 16 B by hardware),
// then the block reads the tile back with ordinary LDS loads.  Prints mismatches (expect 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tile = lds + 1024;                       // not at LDS offset 0
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 1024 + 4096; i += 256) lds[i] = -1.f;
    __syncthreads();
    for (int u = 0; u < 4; ++u) {                   // float4 slot = tid + 256 u -> wave-uniform base + lane * 16
        const float* src = g + (size_t)(blockIdx.x * 1024 + u * 256 + tid) * 4;
        float* dst = tile + (u * 256 + wave * 64) * 4;          // same for every lane of the wave
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4096; i += 256) out[(size_t)blockIdx.x * 4096 + i] = tile[i];
}
int main() {
    const int nb = 64, n = nb * 4096;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *dout;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&dout, n * 4);
    (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), (1024 + 4096) * 4, 0, d, dout);
    (void)hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += o[i] != h[i];
    printf("global_load_lds_dwordx4: %d mismatches of %d\n", bad, n);
    return bad != 0;
}
