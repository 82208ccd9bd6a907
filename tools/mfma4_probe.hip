// v_mfma_f32_4x4x4_16B_f16 operand layout, checked on the card (hipcc --offload-arch=gfx950 tools/mfma4_probe.hip -o /tmp/mfma4_probe):
// 16 independent 4x4x4 blocks; hypothesis: lane l = 4 b + i supplies A_b[i][0..3] and B_b[0..3][j = l & 3]; D_b[r][j] in VGPR r of lane 4 b + j.
// Also times a chain of 3 accumulators (the mel projection's issue pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(const _Float16* A, const _Float16* B, float* D) {
    const int l = threadIdx.x, b = l >> 2, i = l & 3;
    h4 a, bb;
    for (int k = 0; k < 4; ++k) { a[k] = A[(b * 4 + i) * 4 + k]; bb[k] = B[(b * 4 + k) * 4 + i]; }
    f4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x4f16(a, bb, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = d[r];
}
// v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4 x 4 x 1; hypothesis: lane 4 b + i supplies A_b[i][0] and B_b[0][j = l & 3]; D as above
__global__ void probe32(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f4 d = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 4; ++k) d = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l * 4 + k], B[((l >> 2) * 4 + k) * 4 + (l & 3)], d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = d[r];
}
__global__ void rate32(float* out, int n) {
    float a = 1.5f, b = 0.25f;
    f4 d0 = {0, 0, 0, 0}, d1 = d0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d1, 0, 0, 0);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = (float)(t1 - t0) / (2.f * n); }
    out[1 + threadIdx.x] = d0[0] + d1[1];
}
__global__ void rate(float* out, int n) {
    h4 a = {(_Float16)1.f, (_Float16)2.f, (_Float16)0.5f, (_Float16)1.f}, b = a;
    f4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        d0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, d1, 0, 0, 0);
        d2 = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, d2, 0, 0, 0);
    }
    long long t1 = clock64();
    if (threadIdx.x == 0) { out[0] = (float)(t1 - t0) / (3.f * n); }
    out[1 + threadIdx.x] = d0[0] + d1[1] + d2[2];
}
int main() {
    _Float16 hA[256], hB[256];
    float fA[256], fB[256], hD[256];
    for (int i = 0; i < 256; ++i) { fA[i] = (float)(rand() % 17 - 8); fB[i] = (float)(rand() % 13 - 6); hA[i] = (_Float16)fA[i]; hB[i] = (_Float16)fB[i]; }
    _Float16 *dA, *dB; float *dD, *dR;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024); hipMalloc(&dR, 1024);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        float ref = 0.f; const int b = l >> 2, j = l & 3;
        for (int k = 0; k < 4; ++k) ref += fA[(b * 4 + r) * 4 + k] * fB[(b * 4 + k) * 4 + j];
        if (ref != hD[l * 4 + r]) ++bad;
    }
    printf("layout hypothesis: %s (%d mismatches of 256)\n", bad ? "WRONG" : "confirmed", bad);
    {
        float *dfa, *dfb;
        hipMalloc(&dfa, 1024); hipMalloc(&dfb, 1024);
        hipMemcpy(dfa, fA, 1024, hipMemcpyHostToDevice); hipMemcpy(dfb, fB, 1024, hipMemcpyHostToDevice);
        probe32<<<1, 64>>>(dfa, dfb, dD);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        int bad32 = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            float ref = 0.f; const int b = l >> 2, j = l & 3;
            for (int k = 0; k < 4; ++k) ref += fA[(b * 4 + r) * 4 + k] * fB[(b * 4 + k) * 4 + j];
            if (ref != hD[l * 4 + r]) ++bad32;
        }
        printf("4x4x1 f32 layout hypothesis: %s (%d mismatches of 256)\n", bad32 ? "WRONG" : "confirmed", bad32);
        rate32<<<1, 64>>>(dR, 10000);
        float c32; hipMemcpy(&c32, dR, 4, hipMemcpyDeviceToHost);
        printf("4x4x1 f32, two independent accumulators in turn: %.2f clock64 ticks per MFMA (one wave)\n", c32);
        bad += bad32;
    }
    rate<<<1, 64>>>(dR, 10000);
    float c; hipMemcpy(&c, dR, 4, hipMemcpyDeviceToHost);
    printf("4x4x4 f16, three independent accumulators in turn: %.2f clock64 ticks per MFMA (one wave)\n", c);
    return bad != 0;
}
