// gru.hip — recurrence of Bidirectional(GRU(128, reset_after=True, return_sequences=True), merge_mode='mul')
// (modules.py:311-316), forward and BPTT, exact fp32.
//
// The recurrence is a serial chain of S steps (600 at T=3000), each a [1,128]x[128,384] product per batch
// row: latency-bound, not bandwidth- or MFMA-bound.  Design: ONE workgroup (512 threads) per
// (batch row, direction); the recurrent kernel U lives in registers for the whole sequence (96 fp32 per
// thread), h is exchanged through a double-buffered 512-byte LDS vector with one barrier per step, the
// next step's global operands are issued before the current step's FMAs.  Batch rows are independent, so
// B x 2 workgroups run concurrently (64 of the 256 CUs at B = 32) with no inter-workgroup traffic.
//
// thread (j = tid>>2, q = tid&3): unit j, quarter q of the reduction axis; the 4 partial sums of a unit sit
// in 4 adjacent lanes and are combined with two cross-lane adds.
//
// Keras equations (reset_after=True, gate order z|r|h):
//   gh = h U + b_rec;  z = sigmoid(gx_z + gh_z);  r = sigmoid(gx_r + gh_r)
//   hh = tanh(gx_h + r * gh_h);  h' = z*h + (1-z)*hh
#include "common.h"

#define GRU_U 128
#define GRU_G 384

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}

__global__ __launch_bounds__(512) void gru_fwd_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_b,
                                                      const float* __restrict__ U_f, const float* __restrict__ U_b,
                                                      const float* __restrict__ brec_f, const float* __restrict__ brec_b,
                                                      float* __restrict__ h_f, float* __restrict__ h_b,
                                                      float* __restrict__ sv_f, float* __restrict__ sv_b, int S) {
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* gx = (dir ? gx_b : gx_f) + (size_t)b * S * GRU_G;
    const float* U = dir ? U_b : U_f;
    const float* brec = dir ? brec_b : brec_f;
    float* H = (dir ? h_b : h_f) + (size_t)b * S * GRU_U;
    float* sv = dir ? sv_b : sv_f;
    if (sv) sv += (size_t)b * S * 4 * GRU_U;
    const int tid = threadIdx.x, j = tid >> 2, q = tid & 3;
    // padded h vector: index k lives at k + 4*(k>>5) so the 4 quarters start in different bank groups
    __shared__ __attribute__((aligned(16))) float hl[2][144];
    float u[3][32];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) u[g][kk] = U[(size_t)(32 * q + kk) * GRU_G + g * GRU_U + j];
    const float bz = brec[j], br = brec[GRU_U + j], bh = brec[2 * GRU_U + j];
    if (tid < 144) { hl[0][tid] = 0.f; hl[1][tid] = 0.f; }
    float h_own = 0.f;
    __syncthreads();
    int t = dir ? S - 1 : 0;
    const int dt = dir ? -1 : 1;
    float gxz = gx[(size_t)t * GRU_G + j], gxr = gx[(size_t)t * GRU_G + GRU_U + j], gxh = gx[(size_t)t * GRU_G + 2 * GRU_U + j];
    for (int step = 0; step < S; ++step, t += dt) {
        // prefetch next step's input projections
        float nz = 0.f, nr = 0.f, nh = 0.f;
        if (step + 1 < S) {
            const float* gn = gx + (size_t)(t + dt) * GRU_G;
            nz = gn[j]; nr = gn[GRU_U + j]; nh = gn[2 * GRU_U + j];
        }
        const float* hp = &hl[step & 1][36 * q];
        float az = 0.f, ar = 0.f, ah = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
            const float4 hv = *reinterpret_cast<const float4*>(hp + 4 * k4);
            az = fmaf(hv.x, u[0][4 * k4 + 0], az); ar = fmaf(hv.x, u[1][4 * k4 + 0], ar); ah = fmaf(hv.x, u[2][4 * k4 + 0], ah);
            az = fmaf(hv.y, u[0][4 * k4 + 1], az); ar = fmaf(hv.y, u[1][4 * k4 + 1], ar); ah = fmaf(hv.y, u[2][4 * k4 + 1], ah);
            az = fmaf(hv.z, u[0][4 * k4 + 2], az); ar = fmaf(hv.z, u[1][4 * k4 + 2], ar); ah = fmaf(hv.z, u[2][4 * k4 + 2], ah);
            az = fmaf(hv.w, u[0][4 * k4 + 3], az); ar = fmaf(hv.w, u[1][4 * k4 + 3], ar); ah = fmaf(hv.w, u[2][4 * k4 + 3], ah);
        }
        az = quad_sum(az); ar = quad_sum(ar); ah = quad_sum(ah);
        const float z = sigmoidf_(gxz + az + bz);
        const float r = sigmoidf_(gxr + ar + br);
        const float ghh = ah + bh;
        const float hh = tanhf(gxh + r * ghh);
        const float hn = z * h_own + (1.f - z) * hh;
        h_own = hn;
        if (q == 0) {
            hl[(step + 1) & 1][j + 4 * (j >> 5)] = hn;
            H[(size_t)t * GRU_U + j] = hn;
        }
        if (sv) {
            const float val = q == 0 ? z : (q == 1 ? r : (q == 2 ? hh : ghh));
            sv[((size_t)t * 4 + q) * GRU_U + j] = val;
        }
        gxz = nz; gxr = nr; gxh = nh;
        __syncthreads();
    }
}

int launch_gru_fwd(hipStream_t st, const float* gx_f, const float* gx_b, const float* U_f, const float* U_b,
                   const float* brec_f, const float* brec_b, float* h_f, float* h_b, float* sv_f, float* sv_b,
                   int B, int S) {
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(2 * B), dim3(512), 0, st, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b,
                       sv_f, sv_b, S);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// BPTT.  Per step (reverse of the forward processing order):
//   dh = dout[t]*h_other[t] + carry
//   dhh = dh*(1-z); dz = dh*(h_prev - hh); a_h = dhh*(1-hh^2); a_z = dz*z*(1-z); a_r = a_h*ghh*r*(1-r)
//   dgx[t] = [a_z, a_r, a_h]   (input side)      dgh[t] = [a_z, a_r, a_h*r]   (recurrent side)
//   carry  = dh*z + dgh[t] U^T
// thread (j, q) holds U[j][96q .. 96q+95] and reduces its quarter of the 384-long dot product.
__global__ __launch_bounds__(512) void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ h_f,
                                                      const float* __restrict__ h_b, const float* __restrict__ sv_f,
                                                      const float* __restrict__ sv_b, const float* __restrict__ U_f,
                                                      const float* __restrict__ U_b, float* __restrict__ dgx_f,
                                                      float* __restrict__ dgx_b, float* __restrict__ dgh_f,
                                                      float* __restrict__ dgh_b, int S) {
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* dO = dout + (size_t)b * S * GRU_U;
    const float* Hown = (dir ? h_b : h_f) + (size_t)b * S * GRU_U;
    const float* Hoth = (dir ? h_f : h_b) + (size_t)b * S * GRU_U;
    const float* sv = (dir ? sv_b : sv_f) + (size_t)b * S * 4 * GRU_U;
    const float* U = dir ? U_b : U_f;
    float* dgx = (dir ? dgx_b : dgx_f) + (size_t)b * S * GRU_G;
    float* dgh = (dir ? dgh_b : dgh_f) + (size_t)b * S * GRU_G;
    const int tid = threadIdx.x, j = tid >> 2, q = tid & 3;
    // padded gate-gradient vector: index c lives at c + 4*(c/96)
    __shared__ __attribute__((aligned(16))) float gl[2][400];
    float ut[96];
#pragma unroll
    for (int cc = 0; cc < 96; ++cc) ut[cc] = U[(size_t)j * GRU_G + 96 * q + cc];
    // forward processed t = 0..S-1 (dir 0) or S-1..0 (dir 1); BPTT walks it backwards
    int t = dir ? 0 : S - 1;
    const int dt = dir ? 1 : -1;        // BPTT direction
    float carry = 0.f;
    // operands of the current step
    float c_do = dO[(size_t)t * GRU_U + j] * Hoth[(size_t)t * GRU_U + j];
    float c_z = sv[((size_t)t * 4 + 0) * GRU_U + j], c_r = sv[((size_t)t * 4 + 1) * GRU_U + j];
    float c_hh = sv[((size_t)t * 4 + 2) * GRU_U + j], c_gh = sv[((size_t)t * 4 + 3) * GRU_U + j];
    float c_hp = (S > 1) ? Hown[(size_t)(t + dt) * GRU_U + j] : 0.f;  // h_prev = output of the step processed before t
    for (int step = 0; step < S; ++step, t += dt) {
        const bool last = (step + 1 == S);
        const float hp = last ? 0.f : c_hp;
        float n_do = 0.f, n_z = 0.f, n_r = 0.f, n_hh = 0.f, n_gh = 0.f, n_hp = 0.f;
        if (!last) {
            const int tn = t + dt;
            n_do = dO[(size_t)tn * GRU_U + j] * Hoth[(size_t)tn * GRU_U + j];
            n_z = sv[((size_t)tn * 4 + 0) * GRU_U + j]; n_r = sv[((size_t)tn * 4 + 1) * GRU_U + j];
            n_hh = sv[((size_t)tn * 4 + 2) * GRU_U + j]; n_gh = sv[((size_t)tn * 4 + 3) * GRU_U + j];
            if (step + 2 < S) n_hp = Hown[(size_t)(tn + dt) * GRU_U + j];
        }
        const float dh = c_do + carry;
        const float dhh = dh * (1.f - c_z);
        const float dzg = dh * (hp - c_hh);
        const float a_h = dhh * (1.f - c_hh * c_hh);
        const float a_z = dzg * c_z * (1.f - c_z);
        const float a_r = a_h * c_gh * c_r * (1.f - c_r);
        const float a_hr = a_h * c_r;
        float* gw = gl[step & 1];
        // padded positions: c + 4*(c/96)
        if (q == 0) { const int c = j; gw[c + 4 * (c / 96)] = a_z; dgx[(size_t)t * GRU_G + c] = a_z; dgh[(size_t)t * GRU_G + c] = a_z; }
        if (q == 1) { const int c = GRU_U + j; gw[c + 4 * (c / 96)] = a_r; dgx[(size_t)t * GRU_G + c] = a_r; dgh[(size_t)t * GRU_G + c] = a_r; }
        if (q == 2) { const int c = 2 * GRU_U + j; gw[c + 4 * (c / 96)] = a_hr; dgx[(size_t)t * GRU_G + c] = a_h; dgh[(size_t)t * GRU_G + c] = a_hr; }
        __syncthreads();
        const float* gp = gw + 100 * q;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < 24; ++c4) {
            const float4 gv = *reinterpret_cast<const float4*>(gp + 4 * c4);
            s0 = fmaf(gv.x, ut[4 * c4 + 0], s0);
            s1 = fmaf(gv.y, ut[4 * c4 + 1], s1);
            s0 = fmaf(gv.z, ut[4 * c4 + 2], s0);
            s1 = fmaf(gv.w, ut[4 * c4 + 3], s1);
        }
        const float sum = quad_sum(s0 + s1);
        carry = dh * c_z + sum;
        c_do = n_do; c_z = n_z; c_r = n_r; c_hh = n_hh; c_gh = n_gh; c_hp = n_hp;
    }
}

int launch_gru_bwd(hipStream_t st, const float* dout, const float* h_f, const float* h_b, const float* sv_f,
                   const float* sv_b, const float* U_f, const float* U_b, float* dgx_f, float* dgx_b,
                   float* dgh_f, float* dgh_b, int B, int S) {
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(2 * B), dim3(512), 0, st, dout, h_f, h_b, sv_f, sv_b, U_f, U_b, dgx_f, dgx_b,
                       dgh_f, dgh_b, S);
    return 0;
}
