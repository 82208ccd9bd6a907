// Where do the blocks of a 512 x 256-thread, 63 KB-LDS persistent grid land?  Prints, per block, the XCC,
// SE, CU and the wave slot of its first wave, to see which blocks share a CU.  hipcc --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out) {
    extern __shared__ float smem[];
    smem[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID, 32 bits
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127);                 // stay resident until all are placed
}
int main() {
    const int grid = 512, smem = 62848;
    unsigned* d;
    hipMalloc(&d, grid * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), smem, 0, d);
    std::vector<unsigned> h(grid * 2);
    hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < grid; ++b) {
        const unsigned v = h[2 * b], x = h[2 * b + 1];
        printf("block %3d  xcc %u  se %u  sh %u  cu %2u  simd %u  slot %u\n", b, x & 15, (v >> 13) & 7, (v >> 12) & 1, (v >> 8) & 15,
               (v >> 4) & 3, v & 15);
    }
    return 0;
}
