// How fast can 1.57 GB be written with the access shapes a 32x32 MFMA accumulator tile can produce?
//   A: global_store_dword,   lane = channel: 2 x 128 B segments per instruction (what the conv epilogue does)
//   B: global_store_dwordx4, 4x4-transposed: 8 x 128 B segments per instruction
//   C: global_store_dwordx4, 1 KB contiguous per instruction (after an LDS transpose)
// Same block->tile ownership as the conv kernel (512 persistent blocks, tile = 10 rows x 64 px x 64 ch = 160 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* z, int ntiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        float* base = z + (size_t)tile * 10 * 4096 + (size_t)(wave >> 1) * 5 * 4096 + (wave & 1) * 32 * 64;   // wave: 5 rows x 32 px x 64 ch
        if (MODE == 0) {
            for (int c = 0; c < 2; ++c)
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) base[j * 4096 + ((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + c * 32 + li] = (float)r;
        } else if (MODE == 1) {
            const int k = li & 3, jq = li >> 2;
            for (int c = 0; c < 2; ++c)
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(base + j * 4096 + (k + 8 * q + 4 * hi) * 64 + c * 32 + 4 * jq) = make_float4(1.f, 2.f, 3.f, (float)q);
        } else {
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    *reinterpret_cast<float4*>(base + j * 4096 + q * 256 + lane * 4) = make_float4(1.f, 2.f, 3.f, (float)q);
        }
    }
}
template <int MODE>
static void run(const char* name, float* d) {
    const int ntiles = 9600;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(512), dim3(256), 0, 0, d, ntiles);
    (void)hipEventRecord(e0, 0);
    const int n = 30;
    for (int rep = 0; rep < n; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(512), dim3(256), 0, 0, d, ntiles);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.4f ms, %.2f TB/s\n", name, ms / n, 9600.0 * 10 * 4096 * 4 / (ms / n * 1e-3) / 1e12);
}
int main() {
    float* d;
    (void)hipMalloc(&d, (size_t)9600 * 10 * 4096 * 4);
    run<0>("A dword, 2x128B     ", d);
    run<1>("B dwordx4, 8x128B   ", d);
    run<2>("C dwordx4, 1KB      ", d);
    run<0>("A dword, 2x128B     ", d);
    return 0;
}
