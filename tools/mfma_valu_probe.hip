// Do a SIMD's matrix pipe and its VALU run concurrently when the instructions come from two DIFFERENT waves?  Workgroup = 8 waves
// (2 per SIMD); mode 0: all waves issue bf16 MFMAs; 1: all issue v_fma_f32; 2: the first wave of each SIMD issues MFMAs, the second
// v_fma_f32; 3/4: only one wave per SIMD is active (MFMA / VALU alone).  Prints time per mode; if the pipes overlap, mode 2 takes
// about max(mode 3, mode 4), not their sum.        hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_probe.hip -o /tmp/mvp && /tmp/mvp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    // waves 0..3 land on SIMDs 0..3, waves 4..7 on SIMDs 0..3 again (round-robin placement)
    const bool first = wave < 4;
    const bool even = (wave & 1) == 0;
    const bool do_mfma = mode == 0 || (mode == 2 && first) || (mode == 3 && first) || (mode == 5 && even);
    const bool do_valu = mode == 1 || (mode == 2 && !first) || (mode == 4 && !first) || (mode == 5 && !even);
    if (mode == 6) {        // which SIMD does each wave of the workgroup run on?  HW_ID bits [5:4]
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
        if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = (float)((id >> 4) & 3);
        return;
    }
    float a = threadIdx.x * 1e-3f, s = 0.f;
    __shared__ float4 lbuf[4096];
    if ((mode == 7 && !first) || (mode == 8 && !first)) {      // LDS-read wave: 64 ds_read_b128 (1 KB each) per iteration
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) lbuf[i] = make_float4(a, a, a, a);
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* p = lbuf + (threadIdx.x & 63);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                float4 r;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)(size_t)p), "n"(0));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                t.x += r.x;
            }
        }
        s = t.x;
    } else if (mode == 7 && first) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        bf16x8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(a + j); y[j] = (__bf16)(a - j); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    } else if ((mode == 9 || mode == 10) && first) {      // MFMA with the accumulators in AGPRs (inline asm), alone (10) or beside a VALU wave (9)
        f32x16 c0, c1, c2, c3;
        for (int j = 0; j < 16; ++j) { c0[j] = 0.f; c1[j] = 0.f; c2[j] = 0.f; c3[j] = 0.f; }
        bf16x8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(a + j); y[j] = (__bf16)(a - j); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(x), "v"(y));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(x), "v"(y));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c2) : "v"(x), "v"(y));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c3) : "v"(x), "v"(y));
            }
        }
        s += c0[0] + c1[0] + c2[0] + c3[7];
    } else if (mode == 9 && !first) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = a + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(1e-3f));
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    } else if (do_mfma) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        bf16x8 x, y;
        for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(a + j); y[j] = (__bf16)(a - j); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[i & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    } else if (do_valu) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = a + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k)          // 8 x 16 = 128 v_fma per iteration: about the issue time of 16 MFMAs (512 cycles)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0001f), "v"(1e-3f));
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    const int iters = 4000;
    const char* names[] = {"all waves MFMA (2 per SIMD)", "all waves VALU (2 per SIMD)", "one MFMA wave + one VALU wave per SIMD", "one MFMA wave per SIMD alone",
                           "one VALU wave per SIMD alone", "even waves MFMA, odd waves VALU", "", "one MFMA wave + one LDS wave (64 ds_read_b128 per iteration) per SIMD",
                           "one LDS wave per SIMD alone", "one MFMA wave (AGPR accumulators) + one VALU wave per SIMD", "one MFMA wave (AGPR accumulators) alone"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 11; ++mode) {
            if (mode == 6) continue;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, iters, mode);
            (void)hipEventRecord(e0, 0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, d, iters, mode);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%-45s %.3f ms per launch (%d x [16 MFMA 32x32x16 | 128 v_fma] per active wave)\n", names[mode], ms / 5, iters);
        }
    hipLaunchKernelGGL(probe, dim3(1), dim3(512), 0, 0, d, 1, 6);
    float h[8];
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("SIMD of waves 0..7 of a workgroup:");
    for (int i = 0; i < 8; ++i) printf(" %d", (int)h[i]);
    printf("\n");
    return 0;
}
