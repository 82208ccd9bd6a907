// gru.hip — recurrence of Bidirectional(GRU(128, reset_after=True, return_sequences=True), merge_mode='mul')
// (modules.py:311-316), forward and BPTT, exact fp32.
//
// The recurrence is a serial chain of S steps (600 at T=3000), each a [1,128]x[128,384] product per batch
// row: latency-bound, not bandwidth- or MFMA-bound.  Design: ONE workgroup (512 threads) per
// (batch row, direction); the recurrent kernel U lives in registers for the whole sequence (96 fp32 per
// thread), h is exchanged through a double-buffered 512-byte LDS vector with ONE barrier per step.
// Global operands never sit on the step's critical path: they are staged per chunk of steps
// (issue the next chunk's float4 loads at the start of a chunk, commit them to LDS at its end), because
// gfx950's in-order vmcnt would otherwise make every step wait for the newest prefetch.
// Batch rows are independent, so B x 2 workgroups run concurrently (64 of the 256 CUs at B = 32) with
// no inter-workgroup traffic.
//
// thread (j = tid>>2, q = tid&3): unit j, quarter q of the reduction axis; the 4 partial sums of a unit sit
// in 4 adjacent lanes and are combined with two cross-lane adds.
//
// Keras equations (reset_after=True, gate order z|r|h):
//   gh = h U + b_rec;  z = sigmoid(gx_z + gh_z);  r = sigmoid(gx_r + gh_r)
//   hh = tanh(gx_h + r * gh_h);  h' = z*h + (1-z)*hh
#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
// v_pk_fma_f32: two fp32 FMAs per lane per instruction.  The scalar v_fma_f32 issues a wave64 in 4 cycles
// on gfx950, so the mat-vec of the recurrence (the step's longest phase) runs twice as fast packed.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// -DGRU_TIMING (diagnostic build only, tools/tune_gru.py): per-phase shader-cycle sums of wave 0 of every workgroup, read back
// with seld_k_gru_timing.  Stamps follow cdna_hip_programming.md section 7 (s_memtime + lgkmcnt(0) in ONE asm statement between
// sched_barriers); they perturb the schedule (their waits drain the LDS queue), so read the SHARES, not the total.
#ifdef GRU_TIMING
__device__ unsigned long long g_gru_timing[2][512][4];    // [fwd | bwd][workgroup][phase]
#define GRU_STAMP(t_)                                                                      \
    {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
#else
#define GRU_STAMP(t_)
#endif

// -DGRU_TRACE (diagnostic build, tools/trace_gru.py): absolute s_memtime stamps of EVERY wave of workgroup 0 over recurrence steps
// 100..107 at up to four points of a step -> a timeline of who waits for whom (read back with seld_k_gru_timing(2 | 3, ...)).
#ifdef GRU_TRACE
__device__ unsigned long long g_gru_trace[2][8][8][4];    // [fwd | bwd][wave][step - 100][stamp]
#define GRU_TR(w_, k_)                                                                                   \
    if (blockIdx.x == 0 && step >= 100 && step < 108) {                                                  \
        unsigned long long t_;                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if ((threadIdx.x & 63) == 0) g_gru_trace[w_][threadIdx.x >> 6][step - 100][k_] = t_;             \
    }
#else
#define GRU_TR(w_, k_)
#endif

// -DGRU_TRACE2 (diagnostic build, tools/trace_gru2.py): near-free stamps of the VAR 1 kernels — s_memtime into SGPR pairs with NO wait
// of their own (the barrier's lgkmcnt(0) collects them; an outstanding s_memtime only makes the compiler's counted LDS waits one more
// conservative), stored after the barrier for every wave of workgroup 0 over steps 100..107.
#ifdef GRU_TRACE2
__device__ unsigned long long g_gru_trace2[2][8][8][6];    // [fwd | bwd][wave][step - 100][stamp]
#define TR2(k_) asm volatile("s_memtime %0" : "=s"(tr2[k_]));
#define TR2_DECL unsigned long long tr2[6] = {0, 0, 0, 0, 0, 0};
// an s_memtime result arrives LATE: its SGPR pair must stay allocated until a wait has collected it, or the returning value lands in
// whatever the allocator put there next (an address: a memory fault).  Wherever stamps are not followed by a barrier + TR2_STORE:
#define TR2_DRAIN asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(tr2[0]), "s"(tr2[1]), "s"(tr2[2]), "s"(tr2[3]), "s"(tr2[4]), "s"(tr2[5]) : "memory");
// after a barrier (lgkmcnt(0) has collected every stamp): stamps [k0_, k1_) of step ts_
#define TR2_STORE(w_, ts_, k0_, k1_)                                                                                    \
    {                                                                                                                   \
        TR2_DRAIN                                                                                                       \
        if (blockIdx.x == 0 && (ts_) >= 100 && (ts_) < 108 && (threadIdx.x & 63) == 0)                                  \
            for (int k_ = k0_; k_ < k1_; ++k_) g_gru_trace2[w_][threadIdx.x >> 6][(ts_) - 100][k_] = tr2[k_];           \
    }
#else
#define TR2(k_)
#define TR2_DRAIN
#define TR2_DECL
#define TR2_STORE(w_, ts_, k0_, k1_)
#endif

#define GRU_U 128
#define GRU_G 384
#define GRUF_CH 16   // forward: steps per staged chunk
#define GRUB_CH 8    // backward
#define GRUB_ROW 896 // floats staged per backward step: dout | h_other | z r hh gh | h_prev

// branch-free activations: v_exp_f32 / v_rcp_f32 based, abs error ~1e-7 (parity bar 1e-4)
__device__ __forceinline__ float sigmoid_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
// sum over the 4 lanes of a quad with DPP quad_perm moves (VALU; no LDS round trip like ds_bpermute)
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
    return v;
}

// VAR 0: round 3's step body.  VAR 1 (round 4, profiles/r04_gru_experiments.txt): the same layout with the step body re-ordered —
//   * odd lanes of a quad hold U's z and r columns SWAPPED, so "mine" / "other" partial sums need no per-lane select before the fold;
//   * the whole K-quarter of h is read up front (8 ds_read_b128), then the z | r chains run FIRST and their fold + sigmoid are issued
//     among the candidate gate's 16 packed FMAs (two sub-chains), instead of every gate finishing together behind the last FMA;
//   * log2(e) factors folded into fma operands: sigmoid = rcp(1 + exp2(fma(s, -log2e, pre))), tanh through exp2(fma(r', gh, gx')),
//     and the blend is ONE fma behind the last rcp: h' = fma(-2(1-z), rc, z h + (1-z)).
// DROP (Keras GRU recurrent_dropout, modules.py:312-314; training only): the cell's previous state is multiplied by a per-(clip, unit) mask
// rm = 0 | 1 / (1 - rate), constant over the sequence, before it is used — GRUCell.call, implementation 2: `h_tm1 = h_tm1 * rec_dp_mask[0]`
// ahead of the recurrent product AND of the blend z * h_tm1 + (1 - z) * hh.  The layer's output stays the unmasked h (stored to H);
// the masked state is what the exchange buffer / h_own carry and, for the backward pass, what HM receives.
template <int VAR, bool SAVE, bool PRIO = false, bool DROP = false>
__global__ __launch_bounds__(512) void gru_fwd_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_b,
                                                      const float* __restrict__ U_f, const float* __restrict__ U_b,
                                                      const float* __restrict__ brec_f, const float* __restrict__ brec_b,
                                                      float* __restrict__ h_f, float* __restrict__ h_b,
                                                      float* __restrict__ sv_f, float* __restrict__ sv_b, int S,
                                                      const float* __restrict__ rm_f = nullptr, const float* __restrict__ rm_b = nullptr,
                                                      float* __restrict__ hm_f = nullptr, float* __restrict__ hm_b = nullptr) {
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* gx = (dir ? gx_b : gx_f) + (size_t)b * S * GRU_G;
    const float* U = dir ? U_b : U_f;
    const float* brec = dir ? brec_b : brec_f;
    float* H = (dir ? h_b : h_f) + (size_t)b * S * GRU_U;
    float* sv = dir ? sv_b : sv_f;
    if (sv) sv += (size_t)b * S * 4 * GRU_U;
    const int tid = threadIdx.x, j = tid >> 2, q = tid & 3;
    float mk = 1.f;
    float* HM = nullptr;
    if constexpr (DROP) { mk = (dir ? rm_b : rm_f)[b * GRU_U + j]; HM = (dir ? hm_b : hm_f) + (size_t)b * S * GRU_U; }
    __shared__ __attribute__((aligned(16))) float gxl[2][GRUF_CH * GRU_G];
    // padded h vector: index k lives at k + 4*(k>>5) so the 4 quarters start in different bank groups
    __shared__ __attribute__((aligned(16))) float hl[2][144];
    const bool odd = q & 1;
    f32x2 u[3][16];   // u[g][p] = (U[32q+2p][g*128+j], U[32q+2p+1][g*128+j]); VAR 1: gates 0 / 1 swapped in odd lanes (mine | other)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int gs = (VAR == 1 && g < 2 && odd) ? 1 - g : g;
            u[g][p].x = U[(size_t)(32 * q + 2 * p) * GRU_G + gs * GRU_U + j];
            u[g][p].y = U[(size_t)(32 * q + 2 * p + 1) * GRU_G + gs * GRU_U + j];
        }
    const int zr_off = (odd ? GRU_U : 0) + j;
    const float bzr = brec[zr_off], bh = brec[2 * GRU_U + j];
    if (tid < 144) { hl[0][tid] = 0.f; hl[1][tid] = 0.f; }
    float h_own = 0.f, pre_n = 0.f, gxh2_n = 0.f;
    TR2_DECL
    const unsigned h_off = 4u * j, sv_off = 4u * (4 * j + q);
    const int nchunks = (S + GRUF_CH - 1) / GRUF_CH;
    // staged chunk: three float4 per thread held in NAMED registers (an array captured by a lambda was
    // demoted to scratch memory by the compiler, which put a vmcnt wait right behind the loads)
    float4 stg0, stg1, stg2;
    // chunk c = processing steps [c*CH, c*CH+n); its rows are contiguous in memory from row tlo
#define GRUF_CHUNK_ROWS(c, n, tlo)                     \
    {                                                  \
        const int s0_ = (c) * GRUF_CH;                 \
        n = min(GRUF_CH, S - s0_);                     \
        tlo = dir ? S - s0_ - n : s0_;                 \
    }
#define GRUF_ISSUE(c)                                                                              \
    {                                                                                              \
        int n_, tlo_;                                                                              \
        GRUF_CHUNK_ROWS(c, n_, tlo_)                                                               \
        const float4* src_ = reinterpret_cast<const float4*>(gx + (size_t)tlo_ * GRU_G);           \
        const int lim_ = n_ * (GRU_G / 4);                                                         \
        stg0 = src_[tid < lim_ ? tid : 0];               /* rows past the chunk are never read */  \
        stg1 = src_[tid + 512 < lim_ ? tid + 512 : 0];                                             \
        stg2 = src_[tid + 1024 < lim_ ? tid + 1024 : 0];                                           \
    }
#define GRUF_COMMIT(buf)                                           \
    {                                                              \
        float4* d_ = reinterpret_cast<float4*>(gxl[buf]);          \
        d_[tid] = stg0; d_[tid + 512] = stg1; d_[tid + 1024] = stg2; \
    }
    GRUF_ISSUE(0)
    GRUF_COMMIT(0)
    __syncthreads();
#ifdef GRU_TIMING
    unsigned long long tm_mv = 0, tm_tail = 0, tm_bar = 0, tm_commit = 0, tm_last = 0;
#endif
    int step = 0;
    for (int c = 0; c < nchunks; ++c) {
        int n, tlo;
        GRUF_CHUNK_ROWS(c, n, tlo)
        // always issue (the last chunk re-reads itself, harmlessly): a conditional issue makes the staged
        // registers a phi of old/new values and the compiler waits for the loads right at the merge
        GRUF_ISSUE(min(c + 1, nchunks - 1))
        const float* gb = gxl[c & 1];
        auto do_step = [&](int i) {
#ifdef GRU_TIMING
            unsigned long long ts0, ts1, ts2;
            GRU_STAMP(ts0)
#endif
            GRU_TR(0, 0)
            const int row = dir ? n - 1 - i : i;
            const int t = tlo + row;
            // lanes q = 0, 2 of a quad finish the update gate, lanes 1, 3 the reset gate: each reads only its own input term
            const float gxzr = gb[row * GRU_G + zr_off], gxh = gb[row * GRU_G + 2 * GRU_U + j];
            const float* hp = &hl[step & 1][36 * q];
            f32x2 az2 = {0.f, 0.f}, ar2 = {0.f, 0.f}, ah2 = {0.f, 0.f};   // (even k, odd k) partial sums
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const float4 hv = *reinterpret_cast<const float4*>(hp + 4 * k4);
                const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
                az2 = pk_fma(h01, u[0][2 * k4], az2); ar2 = pk_fma(h01, u[1][2 * k4], ar2); ah2 = pk_fma(h01, u[2][2 * k4], ah2);
                az2 = pk_fma(h23, u[0][2 * k4 + 1], az2); ar2 = pk_fma(h23, u[1][2 * k4 + 1], ar2); ah2 = pk_fma(h23, u[2][2 * k4 + 1], ah2);
            }
#ifdef GRU_TIMING
            GRU_STAMP(ts1)
#endif
            GRU_TR(0, 1)
            const float az = az2.x + az2.y, ar = ar2.x + ar2.y;
            float ah = ah2.x + ah2.y;
            // fold the z and r sums instead of two quad sums: with its xor-1 neighbour a lane trades the sum it does not
            // finish (even lanes keep z, odd lanes r), then the xor-2 halves are added: 2 cross-lane adds for both gates,
            // ONE sigmoid per lane, and two quad_perm moves hand z and r back to all four lanes
            float zr = (odd ? ar : az) +
                       __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(odd ? az : ar), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
            zr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(zr), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
            ah = quad_sum(ah);
            const float sg = sigmoid_(gxzr + zr + bzr);
            const float z = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0xA0 /*quad_perm [0,0,2,2]*/, 0xF, 0xF, true));
            const float r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0xF5 /*quad_perm [1,1,3,3]*/, 0xF, 0xF, true));
            const float ghh = ah + bh;
            const float hh = tanh_(gxh + r * ghh);
            const float hn = fmaf(z, h_own - hh, hh);      // z h + (1 - z) hh
            h_own = DROP ? hn * mk : hn;
            GRU_TR(0, 2)
            if (q == 0) {
                hl[(step + 1) & 1][j + 4 * (j >> 5)] = h_own;
                H[(size_t)t * GRU_U + j] = hn;
                if constexpr (DROP) HM[(size_t)t * GRU_U + j] = h_own;
            }
            if (sv) {
                // lane q of a quad saves gate q (z | r | hh | gh).  Even lanes finished z and odd lanes r in `sg` itself, so lanes 0 / 1
                // store sg as it is; lanes 2 / 3 pick hh / gh: two selects (seven AND/OR ops with one-hot masks before)
                const float hi2 = q == 2 ? hh : ghh;
                sv[((size_t)t * GRU_U + j) * 4 + q] = q < 2 ? sg : hi2;      // saved gates: [t][unit][z r hh gh]
            }
#ifdef GRU_TIMING
            GRU_STAMP(ts2)
            tm_mv += ts1 - ts0; tm_tail += ts2 - ts1; tm_last = ts2;
#endif
            ++step;
        };
        // VAR 1.  The step's two input terms are read from the staged chunk a step AHEAD (before the barrier that ends the previous
        // step), so the reads queued behind the barrier are the eight of h alone and the first FMA waits for ONE of them.
        auto load_gx = [&](int i) {
            constexpr float L2E = 1.4426950408889634f;
            const int row = dir ? n - 1 - i : i;
            const float gxzr = gb[row * GRU_G + zr_off], gxh = gb[row * GRU_G + 2 * GRU_U + j];
            pre_n = (gxzr + bzr) * -L2E;
            gxh2_n = gxh * (2.f * L2E);
        };
        auto do_step1 = [&](int i, bool prefetch) {
            constexpr float L2E = 1.4426950408889634f;
            const int row = dir ? n - 1 - i : i;
            const int t = tlo + row;
            const float* hp = &hl[step & 1][36 * q];
            const float pre = pre_n, gxh2 = gxh2_n;
            TR2(0)
            float4 hv[8];
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) hv[k4] = *reinterpret_cast<const float4*>(hp + 4 * k4);
            f32x2 am2 = {0.f, 0.f}, ao2 = {0.f, 0.f};   // mine (z in even lanes, r in odd lanes) | other
            // PRIO: a wave's issue priority FALLS as it advances through its 48 FMAs (3, 2, 1, 0 per dozen), so of the two waves of a
            // SIMD the one that is behind wins the arbitration: round-robin at a dozen-FMA grain instead of oldest-first (under which
            // the older wave runs ahead and the younger one finishes its FMAs and its whole dependent tail alone on the SIMD)
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int k4 = 0; k4 < 8; ++k4) {
                const f32x2 h01 = {hv[k4].x, hv[k4].y}, h23 = {hv[k4].z, hv[k4].w};
                am2 = pk_fma(h01, u[0][2 * k4], am2); ao2 = pk_fma(h01, u[1][2 * k4], ao2);
                am2 = pk_fma(h23, u[0][2 * k4 + 1], am2); ao2 = pk_fma(h23, u[1][2 * k4 + 1], ao2);
                if constexpr (PRIO) { if (k4 == 2) __builtin_amdgcn_s_setprio(2); if (k4 == 5) __builtin_amdgcn_s_setprio(1); }
            }
            __builtin_amdgcn_sched_barrier(0);      // nothing of the candidate gate's chain moves up among the z | r chains
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            TR2(1)
            // fold: a lane adds its xor-1 neighbour's OTHER sum (that neighbour's other gate is this lane's own), then the xor-2 half
            float zr = (am2.x + am2.y) +
                       __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ao2.x + ao2.y), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
            zr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(zr), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
            const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(zr, -L2E, pre)));
            const float z = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0xA0 /*quad_perm [0,0,2,2]*/, 0xF, 0xF, true));
            const float r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0xF5 /*quad_perm [1,1,3,3]*/, 0xF, 0xF, true));
            const float r2 = r * (2.f * L2E), omz = 1.f - z;
            const float ba = fmaf(z, h_own, omz), bb = -2.f * omz;
            f32x2 aha = {0.f, 0.f}, ahb = {0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < 8; k4 += 2) {
                const f32x2 h01 = {hv[k4].x, hv[k4].y}, h23 = {hv[k4].z, hv[k4].w};
                const f32x2 g01 = {hv[k4 + 1].x, hv[k4 + 1].y}, g23 = {hv[k4 + 1].z, hv[k4 + 1].w};
                aha = pk_fma(h01, u[2][2 * k4], aha); ahb = pk_fma(g01, u[2][2 * k4 + 2], ahb);
                aha = pk_fma(h23, u[2][2 * k4 + 1], aha); ahb = pk_fma(g23, u[2][2 * k4 + 3], ahb);
            }
#ifdef GRU_TRACE2
            __builtin_amdgcn_sched_barrier(0);
            TR2(2)
#endif
            const f32x2 ah2 = aha + ahb;
            const float ghh = quad_sum(ah2.x + ah2.y) + bh;
            const float rc = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(r2, ghh, gxh2)));
            const float hn = fmaf(bb, rc, ba);      // z h + (1 - z) (1 - 2 rc)
            h_own = DROP ? hn * mk : hn;
#ifdef GRU_TRACE2
            __builtin_amdgcn_sched_barrier(0);
            TR2(3)
#endif
            // all four lanes of a quad hold the same hn and store it to the same word (LDS: a same-address 4-way write costs at most 4
            // array cycles; memory: one dword per quad either way): no exec-mask region, the step body stays ONE basic block
            hl[(step + 1) & 1][j + 4 * (j >> 5)] = h_own;
            __builtin_amdgcn_sched_barrier(0);      // the exchange write leaves first; output stores and the next step's input terms follow
            if constexpr (DROP) *reinterpret_cast<float*>(reinterpret_cast<char*>(HM) + ((unsigned)t * (GRU_U * 4u) + h_off)) = h_own;
            // uniform row base + 32-bit lane offset: the stores take the SGPR-base form, no 64-bit vector address arithmetic per step
            *reinterpret_cast<float*>(reinterpret_cast<char*>(H) + ((unsigned)t * (GRU_U * 4u) + h_off)) = hn;
            if (prefetch) load_gx(i + 1);
            if constexpr (SAVE) {
                const float hh = fmaf(rc, -2.f, 1.f);
                const float hi2 = q == 2 ? hh : ghh;
                *reinterpret_cast<float*>(reinterpret_cast<char*>(sv) + ((unsigned)t * (GRU_U * 16u) + sv_off)) = q < 2 ? sg : hi2;      // saved gates: [t][unit][z r hh gh]
            }
            TR2(4)
            ++step;
        };
#define GRUF_STEP(i_, pf_) { if constexpr (VAR == 1) do_step1(i_, pf_); else do_step(i_); }
        if constexpr (VAR == 1) load_gx(0);
        for (int i = 0; i < n - 1; ++i) {
            GRUF_STEP(i, true)
            lds_barrier();   // LDS-only: __syncthreads() would also drain vmcnt, i.e. wait for this step's global stores
            TR2_STORE(0, step - 1, 0, 5)
#ifdef GRU_TIMING
            { unsigned long long tb; GRU_STAMP(tb) tm_bar += tb - tm_last; }
#endif
        }
        GRUF_STEP(n - 1, false)
        TR2_DRAIN
        GRUF_COMMIT((c + 1) & 1)  // the only wait on the staged loads: one chunk after their issue
        lds_barrier();
#ifdef GRU_TIMING
        { unsigned long long tb; GRU_STAMP(tb) tm_commit += tb - tm_last; }
#endif
    }
#ifdef GRU_TIMING
    if (tid == 0 && blockIdx.x < 512) {
        g_gru_timing[0][blockIdx.x][0] = tm_mv; g_gru_timing[0][blockIdx.x][1] = tm_tail;
        g_gru_timing[0][blockIdx.x][2] = tm_bar; g_gru_timing[0][blockIdx.x][3] = tm_commit;
    }
#endif
}



int g_gru_var = 11;     // kernel choice (process-wide, option "gru_var"): bit 0 = forward step body VAR 1, bit 1 = backward VAR 1, bit 3 = falling
                        // issue priority through the FMAs (with bits 0 / 1); 0 = round 3's kernels (same-box A/B: tools/tune_gru.py)

int launch_gru_fwd(hipStream_t st, const float* gx_f, const float* gx_b, const float* U_f, const float* U_b,
                   const float* brec_f, const float* brec_b, float* h_f, float* h_b, float* sv_f, float* sv_b,
                   int B, int S, const float* rm_f, const float* rm_b, float* hm_f, float* hm_b) {
    if (rm_f) {      // recurrent dropout (training: the gates are saved): masks rm [B][128] per direction, masked state sequences hm [B][S][128]
        if (!rm_b || !hm_f || !hm_b || !sv_f) return -1;
        hipLaunchKernelGGL((gru_fwd_kernel<1, true, false, true>), dim3(2 * B), dim3(512), 0, st, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, sv_f, sv_b, S,
                           rm_f, rm_b, hm_f, hm_b);
        return 0;
    }
#define GRUF_GO(V_, SV_) hipLaunchKernelGGL((gru_fwd_kernel<V_, SV_>), dim3(2 * B), dim3(512), 0, st, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, sv_f, sv_b, S)
    const bool save = sv_f != nullptr;
    if (g_gru_var & 16) return launch_gru_fwd_df(st, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, sv_f, sv_b, B, S);      // gru_df.hip: no barrier between steps
    if ((g_gru_var & 9) == 9) {
        if (save) hipLaunchKernelGGL((gru_fwd_kernel<1, true, true>), dim3(2 * B), dim3(512), 0, st, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, sv_f, sv_b, S);
        else hipLaunchKernelGGL((gru_fwd_kernel<1, false, true>), dim3(2 * B), dim3(512), 0, st, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, sv_f, sv_b, S);
        return 0;
    }
    if (g_gru_var & 1) { if (save) GRUF_GO(1, true); else GRUF_GO(1, false); }
    else { if (save) GRUF_GO(0, true); else GRUF_GO(0, false); }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// BPTT.  Per step (reverse of the forward processing order):
//   dh = dout[t]*h_other[t] + carry
//   dhh = dh*(1-z); dz = dh*(h_prev - hh); a_h = dhh*(1-hh^2); a_z = dz*z*(1-z); a_r = a_h*ghh*r*(1-r)
//   dgx[t] = [a_z, a_r, a_h]   (input side)      dgh[t] = [a_z, a_r, a_h*r]   (recurrent side)
//   carry  = dh*z + dgh[t] U^T
// Register blocking of the 384 -> 128 mat-vec: lane (grp = lane>>4, cp = lane&15) of wave w owns the 4
// outputs j0..j0+3 (j0 = 4*(4w+grp)) over the 24 columns [24cp, 24cp+24): each staged gradient value is
// read from LDS once per 4 outputs (6 ds_read_b128 per step instead of 24 — the one-output-per-lane
// layout was LDS-bandwidth-bound), then an all-reduce over the 16 lanes of the row (quad_perm + row_ror
// DPP) leaves the 4 sums in every lane.  Gate gradients of unit j0 + (cp&3) are computed by the same lane.
__device__ __forceinline__ float row16_allsum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /*row_ror:8*/, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /*row_ror:4*/, 0xF, 0xF, true));
    return v;
}

#define GRUB_GL 448   // padded gate-gradient vector: column c lives at 28*(c/24) + c%24 (16 parts, conflict-free b128 reads)

// VAR 1 (round 4): (a) lane cp keeps U^T's four rows in the order a ^ (cp & 3), so that accumulator 0 is always the output the lane ends
// up owning and the fold over the quad is three DPP adds with NO per-lane selects (VAR 0: six v_cndmask on the carry's critical path);
// (b) the idle quarter of the lanes (role 3) repeats role 0's gate gradient — same value, same address — so the gate stage has no exec-mask
// region and the step is one basic block; (c) the exchange write is pinned ahead of the output stores.
// DROP (see gru_fwd_kernel): h_prev is the MASKED state sequence (hm_*, written by the forward), and the gradient carried to the previous step
// is the gradient w.r.t. that masked state times the mask: carry = rm * (dh z + dgh U^T); h_other (the merge partner) stays the unmasked output
template <int VAR, bool PRIO = false, bool DROP = false>
__global__ __launch_bounds__(512) void gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ h_f,
                                                      const float* __restrict__ h_b, const float* __restrict__ sv_f,
                                                      const float* __restrict__ sv_b, const float* __restrict__ U_f,
                                                      const float* __restrict__ U_b, float* __restrict__ dgx_f,
                                                      float* __restrict__ dgx_b, float* __restrict__ dgh_f,
                                                      float* __restrict__ dgh_b, int S,
                                                      const float* __restrict__ rm_f = nullptr, const float* __restrict__ rm_b = nullptr,
                                                      const float* __restrict__ hm_f = nullptr, const float* __restrict__ hm_b = nullptr) {
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* dO = dout + (size_t)b * S * GRU_U;
    const float* Hown = (DROP ? (dir ? hm_b : hm_f) : (dir ? h_b : h_f)) + (size_t)b * S * GRU_U;
    const float* Hoth = (dir ? h_f : h_b) + (size_t)b * S * GRU_U;
    const float* sv = (dir ? sv_b : sv_f) + (size_t)b * S * 4 * GRU_U;
    const float* U = dir ? U_b : U_f;
    float* dgx = (dir ? dgx_b : dgx_f) + (size_t)b * S * GRU_G;
    float* dgh = (dir ? dgh_b : dgh_f) + (size_t)b * S * GRU_G;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cp = lane & 15;                       // column part of the mat-vec
    const int j0 = 4 * (4 * wave + (lane >> 4));    // first of this lane's 4 outputs
    const int jm = j0 + (cp & 3), qr = cp >> 2;     // unit / role (z, r, h, -) of this lane in the gate stage
    float mk = 1.f;
    if constexpr (DROP) mk = (dir ? rm_b : rm_f)[b * GRU_U + jm];
    const bool b0 = cp & 1, b1 = cp & 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* stage = smem;                          // [2][GRUB_CH][GRUB_ROW]
    float* gl = smem + 2 * GRUB_CH * GRUB_ROW;    // [2][GRUB_GL]
    f32x2 ut[4][12];   // ut[a][p] = U[j0+a][24cp + 2p .. +1]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int p = 0; p < 12; ++p) {
            const int ar = VAR == 1 ? a ^ (cp & 3) : a;
            ut[a][p].x = U[(size_t)(j0 + ar) * GRU_G + 24 * cp + 2 * p];
            ut[a][p].y = U[(size_t)(j0 + ar) * GRU_G + 24 * cp + 2 * p + 1];
        }
    // the forward pass consumed t = 0..S-1 (dir 0) / S-1..0 (dir 1); BPTT walks that order backwards:
    // BPTT step s is time t = S-1-s (dir 0) or s (dir 1); h_prev(t) = H[t-1] (dir 0) / H[t+1] (dir 1)
    const int hshift = dir ? 1 : -1;
    const int nchunks = (S + GRUB_CH - 1) / GRUB_CH;
    float4 stg[4];
    auto chunk_rows = [&](int c, int& n, int& tlo) {
        const int s0 = c * GRUB_CH;
        n = min(GRUB_CH, S - s0);
        tlo = dir ? s0 : S - s0 - n;
    };
    // Each staging slot (tid, uu) always reads the same array: resolve (base pointer, row stride, time shift,
    // chunk row) ONCE, so that issuing a chunk is four independent, branch-free loads (a divergent
    // if/else chain per slot made the compiler serialise them with vmcnt(0) between the branches).
    const float* sbase[4];
    int sstride[4], sshift[4], srow[4];
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
        const int idx = tid + 512 * uu;           // float4 slot: row = idx / 224, col4 = idx % 224
        const int row = idx / (GRUB_ROW / 4), c4 = idx - row * (GRUB_ROW / 4);
        srow[uu] = row;                            // rows >= GRUB_CH never pass the `row < n` test
        sshift[uu] = 0;
        if (c4 < 32) { sbase[uu] = dO + c4 * 4; sstride[uu] = GRU_U; }
        else if (c4 < 64) { sbase[uu] = Hoth + (c4 - 32) * 4; sstride[uu] = GRU_U; }
        else if (c4 < 192) { sbase[uu] = sv + (c4 - 64) * 4; sstride[uu] = 4 * GRU_U; }
        else { sbase[uu] = Hown + (c4 - 192) * 4; sstride[uu] = GRU_U; sshift[uu] = hshift; }
    }
    unsigned okmask = 0;
    auto issue = [&](int c) {
        int n, tlo;
        chunk_rows(c, n, tlo);
        okmask = 0;
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int tt = tlo + srow[uu] + sshift[uu];
            const bool ok = (srow[uu] < n) && (tt >= 0) && (tt < S);
            okmask |= (ok ? 1u : 0u) << uu;
            stg[uu] = *reinterpret_cast<const float4*>(sbase[uu] + (size_t)(ok ? tt : 0) * sstride[uu]);   // always in bounds
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int uu = 0; uu < 4; ++uu) {
            const int idx = tid + 512 * uu;
            const float4 v = ((okmask >> uu) & 1u) ? stg[uu] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < GRUB_CH * (GRUB_ROW / 4)) reinterpret_cast<float4*>(stage + buf * GRUB_CH * GRUB_ROW)[idx] = v;
        }
    };
    issue(0);
    commit(0);
    __syncthreads();
    float carry = 0.f;    // carry of unit jm
    TR2_DECL
#ifdef GRU_TIMING
    unsigned long long tm_p1 = 0, tm_bar = 0, tm_p2 = 0;
#endif
    int step = 0;
    // Everything in a step's gate gradients except the factor dh = dout*h_other + carry is known before the step's carry
    // is: a_z = dh*kz, a_r = dh*kr, a_h = dh*kh, a_h*r = dh*khr with kh = (1-z)(1-hh^2), kz = (h_prev-hh) z (1-z),
    // kr = kh*gh*r*(1-r), khr = kh*r.  pre() forms this lane's two coefficients (its role qr picks them) for the NEXT step
    // while the current step's mat-vec runs; the chain behind the carry is then one add and two multiplies.
    const int cidx = (qr < 3 ? qr : 0) * GRU_U + jm;
    const int gl_slot = 28 * (cidx / 24) + cidx % 24;
    const unsigned c_off = 4u * cidx;
    float k_do = 0.f, k_x = 0.f, k_h = 0.f, k_z = 0.f, c_zs = 0.f;
    const float m_z = (qr == 0 || qr == 3) ? 1.f : 0.f, m_r = qr == 1 ? 1.f : 0.f, m_h = qr == 2 ? 1.f : 0.f;
    auto pre = [&](const float* sbuf, int row) {
        const float* rp = sbuf + row * GRUB_ROW + jm;
        k_do = rp[0] * rp[128];
        const float4 sg4 = *reinterpret_cast<const float4*>(sbuf + row * GRUB_ROW + 256 + 4 * jm);   // saved gates [unit][z r hh gh]
        const float c_z = sg4.x, c_r = sg4.y, c_hh = sg4.z, c_gh = sg4.w, hp = rp[768];
        const float kh = (1.f - c_z) * (1.f - c_hh * c_hh);
        const float kz = (hp - c_hh) * c_z * (1.f - c_z);
        const float kr = kh * c_gh * c_r * (1.f - c_r);
        if constexpr (VAR == 1) {
            // role by 0 / 1 lane masks (role 3 repeats role 0): plain multiply-adds — the compiler turned the nested selects into exec-mask
            // branches inside the step
            const float base = fmaf(m_z, kz, m_r * kr);
            k_x = fmaf(m_h, kh, base);
            k_h = fmaf(m_h, kh * c_r, base);
        } else {
            k_x = qr == 0 ? kz : (qr == 1 ? kr : kh);
            k_h = qr == 0 ? kz : (qr == 1 ? kr : kh * c_r);
        }
        k_z = c_z;
    };
    {
        int n0, tlo0;
        chunk_rows(0, n0, tlo0);
        pre(stage, dir ? 0 : n0 - 1);
    }
    for (int c = 0; c < nchunks; ++c) {
        int n, tlo;
        chunk_rows(c, n, tlo);
        issue(min(c + 1, nchunks - 1));   // unconditional: see gru_fwd_kernel
        const float* sb = stage + (c & 1) * GRUB_CH * GRUB_ROW;
        float dh = 0.f;
        float* gw = nullptr;
        auto part1 = [&](int i) {   // gate gradients of step i -> LDS vector + global: dh times the coefficients pre() prepared
            const int row = dir ? i : n - 1 - i;
            const int t = tlo + row;
            dh = k_do + carry;
            c_zs = k_z;
            gw = gl + (step & 1) * GRUB_GL;
            if constexpr (VAR == 1) {
                const float vx = dh * k_x, vh = dh * k_h;
                gw[gl_slot] = vh;
                __builtin_amdgcn_sched_barrier(0);
                // uniform base + 32-bit byte offset (S x 1536 B < 4 GB): SGPR-base stores, one v_add per step for both
                const unsigned off = (unsigned)t * (GRU_G * 4u) + c_off;
                *reinterpret_cast<float*>(reinterpret_cast<char*>(dgx) + off) = vx;
                *reinterpret_cast<float*>(reinterpret_cast<char*>(dgh) + off) = vh;
            } else if (qr < 3) {
                const float vx = dh * k_x, vh = dh * k_h;
                gw[gl_slot] = vh;
                dgx[(size_t)t * GRU_G + cidx] = vx;
                dgh[(size_t)t * GRU_G + cidx] = vh;
            }
        };
        auto part2 = [&]() {        // carry = dh*z + dgh U^T
            const float* gp = gw + 28 * cp;
            f32x2 s2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int c4 = 0; c4 < 6; ++c4) {
                // PRIO: falling issue priority through the 48 FMAs (see gru_fwd_kernel)
                // (tied to the four accumulator chains: a bare s_setprio is not held in place among the FMAs)
                if constexpr (PRIO) {
                    if (c4 == 0) __builtin_amdgcn_s_setprio(3);
                    if (c4 == 2) asm volatile("s_setprio 2" : "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]));
                    if (c4 == 4) asm volatile("s_setprio 1" : "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]));
                }
                const float4 gv = *reinterpret_cast<const float4*>(gp + 4 * c4);
                const f32x2 g01 = {gv.x, gv.y}, g23 = {gv.z, gv.w};
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    s2[a] = pk_fma(g01, ut[a][2 * c4], s2[a]);
                    s2[a] = pk_fma(g23, ut[a][2 * c4 + 1], s2[a]);
                }
            }
            // 4 partial sums x 16 lanes -> lane cp keeps the total of output cp & 3.  Fold instead of four all-reduces: with
            // its xor-1 neighbour a lane trades the two outputs of the other parity (2 adds), with its xor-2 neighbour the
            // remaining foreign one (1 add), then the quads of the row are summed (2 adds): 5 cross-lane adds + 6 selects
            // where four 16-lane all-reduces took 16 + the final select chain.
            if constexpr (PRIO) asm volatile("s_setprio 0" : "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]));
#ifdef GRU_TRACE2
            __builtin_amdgcn_sched_barrier(0);
            TR2(3)
#endif
            const float s0 = s2[0].x + s2[0].y, s1 = s2[1].x + s2[1].y, s2_ = s2[2].x + s2[2].y, s3 = s2[3].x + s2[3].y;
            if constexpr (VAR == 1) {
                // accumulator a holds output (cp & 3) ^ a: the xor-1 neighbour owns what this lane holds in slots 1 and 3, the xor-2 one slot 2
                const float a0 = s0 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
                const float a2 = s2_ + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s3), 0xB1, 0xF, 0xF, true));
                float mine = a0 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a2), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
                mine += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0x128 /*row_ror:8*/, 0xF, 0xF, true));
                mine += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0x124 /*row_ror:4*/, 0xF, 0xF, true));
                carry = dh * c_zs + mine;
                if constexpr (DROP) carry *= mk;
#ifdef GRU_TRACE2
                __builtin_amdgcn_sched_barrier(0);
                TR2(4)
#endif
                ++step;
                return;
            }
            const float k1 = b0 ? s1 : s0, g1 = b0 ? s0 : s1, k2 = b0 ? s3 : s2_, g2 = b0 ? s2_ : s3;
            const float a01 = k1 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g1), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
            const float a23 = k2 + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g2), 0xB1, 0xF, 0xF, true));
            const float kk = b1 ? a23 : a01, gg = b1 ? a01 : a23;
            float mine = kk + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(gg), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
            mine += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0x128 /*row_ror:8*/, 0xF, 0xF, true));
            mine += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0x124 /*row_ror:4*/, 0xF, 0xF, true));
            carry = dh * c_zs + mine;
            if constexpr (DROP) carry *= mk;
            ++step;
        };
        for (int i = 0; i < n - 1; ++i) {
#ifdef GRU_TIMING
            unsigned long long tb0, tb1, tb2, tb3;
            GRU_STAMP(tb0)
#endif
            GRU_TR(1, 0)
            TR2(0)
            part1(i);
            TR2(1)
            GRU_TR(1, 1)
#ifdef GRU_TIMING
            GRU_STAMP(tb1)
#endif
            lds_barrier();   // LDS-only barrier: never wait for the dgx/dgh stores
            TR2_STORE(1, step, 0, 2)
            TR2_STORE(1, step - 1, 2, 5)
            TR2(2)
#ifdef GRU_TIMING
            GRU_STAMP(tb2)
#endif
            GRU_TR(1, 2)
            pre(sb, dir ? i + 1 : n - 2 - i);
            part2();
#ifdef GRU_TIMING
            GRU_STAMP(tb3)
            tm_p1 += tb1 - tb0; tm_bar += tb2 - tb1; tm_p2 += tb3 - tb2;
#endif
        }
        part1(n - 1);
        TR2_DRAIN
        commit((c + 1) & 1);  // the only wait on the staged loads
        lds_barrier();
        {
            // first step of the next chunk (its rows were committed just above); after the last chunk: the same rows again, unused
            int n2, tlo2;
            chunk_rows(min(c + 1, nchunks - 1), n2, tlo2);
            pre(stage + ((c + 1) & 1) * GRUB_CH * GRUB_ROW, dir ? 0 : n2 - 1);
        }
        part2();
        TR2_DRAIN
    }
#ifdef GRU_TIMING
    if (tid == 0 && blockIdx.x < 512) {
        g_gru_timing[1][blockIdx.x][0] = tm_p1; g_gru_timing[1][blockIdx.x][1] = tm_bar;
        g_gru_timing[1][blockIdx.x][2] = tm_p2; g_gru_timing[1][blockIdx.x][3] = 0;
    }
#endif
}

int launch_gru_bwd(hipStream_t st, const float* dout, const float* h_f, const float* h_b, const float* sv_f,
                   const float* sv_b, const float* U_f, const float* U_b, float* dgx_f, float* dgx_b,
                   float* dgh_f, float* dgh_b, int B, int S, const float* rm_f, const float* rm_b, const float* hm_f, const float* hm_b) {
    const size_t smem = (size_t)(2 * GRUB_CH * GRUB_ROW + 2 * GRUB_GL) * sizeof(float);
    if (rm_f) {      // recurrent dropout: see gru_fwd_kernel
        if (!rm_b || !hm_f || !hm_b) return -1;
        auto kd = gru_bwd_kernel<1, false, true>;
        hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kd, dim3(2 * B), dim3(512), smem, st, dout, h_f, h_b, sv_f, sv_b, U_f, U_b, dgx_f, dgx_b, dgh_f, dgh_b, S, rm_f, rm_b, hm_f, hm_b);
        return 0;
    }
    auto kern = (g_gru_var & 2) ? ((g_gru_var & 8) ? gru_bwd_kernel<1, true> : gru_bwd_kernel<1, false>) : gru_bwd_kernel<0, false>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(kern, dim3(2 * B), dim3(512), smem, st, dout, h_f, h_b, sv_f, sv_b, U_f, U_b, dgx_f, dgx_b,
                       dgh_f, dgh_b, S, nullptr, nullptr, nullptr, nullptr);
    return 0;
}

// per-phase cycle sums of the last gru_fwd (which = 0) / gru_bwd (1) launch: out[blocks][4]; -2 unless built with -DGRU_TIMING
int gru_timing_read(int which, unsigned long long* out, int blocks) {
    if ((which == 6 || which == 7) && blocks == 128) { const int r_ = gru_df_trace_read(which - 6, out); return r_ == 1 ? 0 : r_; }      // gru_df.hip, -DDF_TRACE: [wave][16 steps][8]
#ifdef GRU_TRACE2
    if ((which == 4 || which == 5) && blocks == 64) {    // trace2 of workgroup 0: [wave][step][stamp 0..5]
        hipDeviceSynchronize();
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gru_trace2), 8 * 8 * 6 * sizeof(unsigned long long),
                                   (size_t)(which - 4) * 8 * 8 * 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
    }
#endif
#ifdef GRU_TRACE
    if ((which == 2 || which == 3) && blocks == 64) {    // trace of workgroup 0: [wave][step][stamp]
        hipDeviceSynchronize();
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gru_trace), 8 * 8 * 4 * sizeof(unsigned long long),
                                   (size_t)(which - 2) * 8 * 8 * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
    }
#endif
#ifdef GRU_TIMING
    if (which < 0 || which > 1 || blocks < 1 || blocks > 512) return -1;
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gru_timing), (size_t)blocks * 4 * sizeof(unsigned long long),
                               (size_t)which * 512 * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
#else
    return -2;
#endif
}
