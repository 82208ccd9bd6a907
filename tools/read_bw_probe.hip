// HBM read bandwidth with a plain streaming kernel (float4 per lane, grid-stride), for 1.57 GB (the first
// block's pre-BN tensor) — what a read-bound kernel can hope for on this card.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int UNROLL>
__global__ __launch_bounds__(256) void rd(const v4* __restrict__ p, float* out, size_t n4) {
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
    for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < n4; i += stride) {
        v4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + u * 256 < n4) ? p[i + u * 256] : v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
template <int UNROLL>
static void run(const v4* d, float* o, size_t n4, int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rd<UNROLL>, dim3(grid), dim3(256), 0, 0, d, o, n4);
    (void)hipEventRecord(e0, 0);
    const int n = 20;
    for (int r = 0; r < n; ++r) hipLaunchKernelGGL(rd<UNROLL>, dim3(grid), dim3(256), 0, 0, d, o, n4);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("unroll %d grid %5d: %.3f ms  %.2f TB/s\n", UNROLL, grid, ms / n, n4 * 16.0 / (ms / n * 1e-3) / 1e12);
}
int main() {
    const size_t n4 = (size_t)32 * 3000 * 64 * 64 / 4;
    v4* d; float* o;
    (void)hipMalloc(&d, n4 * 16); (void)hipMalloc(&o, 4);
    (void)hipMemset(d, 0, n4 * 16);
    for (int grid : {512, 1024, 2048, 8192}) { run<4>(d, o, n4, grid); run<8>(d, o, n4, grid); }
    return 0;
}
