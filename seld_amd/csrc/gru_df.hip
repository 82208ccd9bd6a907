// gru_df.hip — the GRU recurrence (modules.py:311-316; equations and layouts: gru.hip) WITHOUT a workgroup barrier between steps.
//
// Why.  gru.hip's step is [barrier] [8 LDS reads of h] [48 packed FMAs per wave] [gate tail] [write h'] [barrier]: the two waves of a SIMD
// run the same phases at the same time, so a step costs the SUM of its exposed latencies (h' write -> barrier -> h reads: ~375 cycles;
// the last wave's gate tail: ~200) plus the SIMD's vector issue for both waves (~720) = ~1 270 cycles, and each SIMD is idle a third of the
// step (profiles/r04_gru_experiments.txt section 2).  The chain of dependencies does not ask for that: h'(s) of a unit needs ALL of h(s-1),
// but a wave can start multiplying with the half of h(s-1) that is already there while the other half is still in its producer's gate tail.
//
// How.  The eight waves form two groups, X = waves 0-3 and Y = waves 4-7 (one wave of each group on every SIMD: waves are dealt to the SIMDs
// cyclically).  The reduction axis of every lane is split into the units X produces (0..63) and the units Y produces (64..127):
//     phase A: 24 packed FMAs on h_X(s-1)        phase B: 24 packed FMAs on h_Y(s-1), gate tail, publish h'(s)
// Publishing = the h' write followed by a per-wave step counter in LDS (a wave's LDS operations complete in order, so whoever sees the counter
// sees the data); a consumer polls the four counters of the group it needs — in the same LDS batch as the data reads, so a ready group costs
// one LDS round trip.  In the steady state one group runs ahead of the other by about half a step: while a SIMD's Y wave sits in its
// latency-bound gate tail its X wave issues phase A of the next step, and vice versa — the exposed latencies of one wave are filled with the
// other wave's FMAs, and the step approaches what the SIMD has to ISSUE.  A workgroup barrier remains once per staged chunk of 16 steps (the
// commit of the next chunk's input rows).  Double buffering of h is enough: a wave can be at most one step ahead of the slowest one (it needs
// every wave's h(s) before it can publish h(s+1), and every wave has finished reading h(s-1) before it publishes h(s)).
//
// Every spin is bounded (DF_SPIN_LIMIT polls, ~10 ms): a wave that never sees its flag gives up and the kernel ends with wrong results
// (which the parity tests catch) instead of hanging the device.
#include "common.h"
#include <cstring>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

#define GRU_U 128
#define GRU_G 384
#define DF_CH 16            // steps per staged chunk of input rows
#define DF_SPIN_LIMIT (1 << 17)

// -DDF_TRACE (diagnostic build, tools/trace_gru_df.py): s_memtime stamps of every wave of workgroup 0 over 16 consecutive steps, collected by the
// step's own LDS waits (no wait of their own), + the number of counter polls that found a group not ready.
//   0 step start | 1 group X's half landed | 2 phase A's FMAs issued | 3 group Y's half landed | 4 h' published | 5 polls A | 6 polls B
#ifdef DF_STATS
__device__ unsigned long long g_df_stats[2][8][4];      // [fwd | bwd][wave]: polls A, polls B, steps with a B re-read, kernel cycles
#endif
#ifdef DF_TRACE
#define DF_TR_STEP0 96
__device__ unsigned long long g_df_trace[2][8][16][8];      // [fwd | bwd][wave][step - DF_TR_STEP0][stamp]
#define DFT(k_) asm volatile("s_memtime %0" : "=s"(tr[k_]));
#define DFT_DECL unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DFT_DRAIN asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(tr[0]), "s"(tr[1]), "s"(tr[2]), "s"(tr[3]), "s"(tr[4]) : "memory");
#define DFT_STORE(w_, st_)                                                                                                 \
    {                                                                                                                      \
        DFT_DRAIN                                                                                                          \
        if (blockIdx.x == 0 && (st_) >= DF_TR_STEP0 && (st_) < DF_TR_STEP0 + 16 && (threadIdx.x & 63) == 0)                \
            for (int k_ = 0; k_ < 8; ++k_) g_df_trace[w_][threadIdx.x >> 6][(st_) - DF_TR_STEP0][k_] = tr[k_];            \
    }
#define DFT_SET(k_, v_) tr[k_] = (v_);
#else
#define DFT(k_)
#define DFT_DECL
#define DFT_DRAIN
#define DFT_STORE(w_, st_)
#define DFT_SET(k_, v_)
#endif

// One LDS batch: the four step counters of a producer group (lane i reads counter i & 3) and this lane's 16 values of that group's half of h.
// The counter read is issued FIRST: LDS executes a wave's operations in order, so data read behind a counter that shows `target` is that
// step's data.  Returns with everything landed (lgkmcnt(0)).
__device__ __forceinline__ void df_read_group(unsigned flag_addr, unsigned data_addr, unsigned& fv, f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
    asm volatile(
        "ds_read_b32 %0, %5\n\t"
        "ds_read_b128 %1, %6\n\t"
        "ds_read_b128 %2, %6 offset:16\n\t"
        "ds_read_b128 %3, %6 offset:32\n\t"
        "ds_read_b128 %4, %6 offset:48\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(fv), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(flag_addr), "v"(data_addr)
        : "memory");
}
// Both groups in ONE batch at the top of a step (the counters first): a wave of the group that runs behind finds both halves of h there and pays one
// LDS round trip per step; a wave of the group that runs ahead finds the other group's counter short and re-reads that half later.
__device__ __forceinline__ void df_read_both(unsigned flag_addr, unsigned data_addr, unsigned& fx, unsigned& fy, f32x4& x0, f32x4& x1, f32x4& x2, f32x4& x3,
                                             f32x4& y0, f32x4& y1, f32x4& y2, f32x4& y3) {
    asm volatile(
        "ds_read_b32 %0, %10\n\t"
        "ds_read_b32 %1, %10 offset:16\n\t"
        "ds_read_b128 %2, %11\n\t"
        "ds_read_b128 %3, %11 offset:16\n\t"
        "ds_read_b128 %4, %11 offset:32\n\t"
        "ds_read_b128 %5, %11 offset:48\n\t"
        "ds_read_b128 %6, %11 offset:256\n\t"
        "ds_read_b128 %7, %11 offset:272\n\t"
        "ds_read_b128 %8, %11 offset:288\n\t"
        "ds_read_b128 %9, %11 offset:304\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(fx), "=&v"(fy), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
        : "v"(flag_addr), "v"(data_addr)
        : "memory");
}
// the same batch, ordered BEHIND the FMAs that produced the three accumulator pairs (in / out operands: without the tie the scheduler sinks phase
// A's FMAs below phase B's wait and nothing overlaps)
__device__ __forceinline__ void df_read_group_after(unsigned flag_addr, unsigned data_addr, unsigned& fv, f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3,
                                                    f32x2& t0, f32x2& t1, f32x2& t2) {
    asm volatile(
        "ds_read_b32 %0, %8\n\t"
        "ds_read_b128 %1, %9\n\t"
        "ds_read_b128 %2, %9 offset:16\n\t"
        "ds_read_b128 %3, %9 offset:32\n\t"
        "ds_read_b128 %4, %9 offset:48\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(fv), "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "+v"(t0), "+v"(t1), "+v"(t2)
        : "v"(flag_addr), "v"(data_addr)
        : "memory");
}
#ifndef DF_SLEEP
#define DF_SLEEP 1
#endif
#define DF_STR_(x) #x
#define DF_STR(x) DF_STR_(x)
// one poll of a group's counters; the wave sleeps ~64 x DF_SLEEP cycles first: a spinning wave takes issue slots from its SIMD partner
__device__ __forceinline__ unsigned df_read_flag(unsigned flag_addr) {
    unsigned fv;
    asm volatile("s_sleep " DF_STR(DF_SLEEP) "\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(fv) : "v"(flag_addr) : "memory");
    return fv;
}
// all four counters of the group have reached `target`?
__device__ __forceinline__ bool df_ready(unsigned fv, unsigned target) { return __builtin_amdgcn_ballot_w64(fv < target) == 0ull; }

template <bool SAVE>
__global__ __launch_bounds__(512) void gru_fwd_df_kernel(const float* __restrict__ gx_f, const float* __restrict__ gx_b,
                                                         const float* __restrict__ U_f, const float* __restrict__ U_b,
                                                         const float* __restrict__ brec_f, const float* __restrict__ brec_b,
                                                         float* __restrict__ h_f, float* __restrict__ h_b,
                                                         float* __restrict__ sv_f, float* __restrict__ sv_b, int S) {
    constexpr float L2E = 1.4426950408889634f;
    const int b = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* gx = (dir ? gx_b : gx_f) + (size_t)b * S * GRU_G;
    const float* U = dir ? U_b : U_f;
    const float* brec = dir ? brec_b : brec_f;
    float* H = (dir ? h_b : h_f) + (size_t)b * S * GRU_U;
    float* sv = nullptr;
    if constexpr (SAVE) sv = (dir ? sv_b : sv_f) + (size_t)b * S * 4 * GRU_U;
    const int tid = threadIdx.x, j = tid >> 2, q = tid & 3, wave = tid >> 6, lane = tid & 63;
    __shared__ __attribute__((aligned(16))) float gxl[2][DF_CH * GRU_G];
    __shared__ __attribute__((aligned(16))) float hl[2][GRU_U];      // h(s) in buffer s & 1, plain order
    __shared__ unsigned flags[8];                                     // flags[w] = steps wave w has published (h(flags[w]) is in LDS)
    __shared__ unsigned sink[512];                                    // where lanes 1..63 of a publishing wave put their copy of the counter
    const bool odd = q & 1;
    // u[g][p], p < 8: rows k = 16 q + 2 p, + 1 (group X's units);  p >= 8: rows k = 64 + 16 q + 2 (p - 8), + 1 (group Y's units)
    // gates 0 / 1 swapped in odd lanes (mine | other), as in gru.hip VAR 1
    f32x2 u[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int gs = (g < 2 && odd) ? 1 - g : g;
            const int k = (p < 8 ? 0 : 64) + 16 * q + 2 * (p & 7);
            u[g][p].x = U[(size_t)k * GRU_G + gs * GRU_U + j];
            u[g][p].y = U[(size_t)(k + 1) * GRU_G + gs * GRU_U + j];
        }
    const int zr_off = (odd ? GRU_U : 0) + j;
    const float bzr = brec[zr_off], bh = brec[2 * GRU_U + j];
    if (tid < 2 * GRU_U) (&hl[0][0])[tid] = 0.f;
    if (tid < 8) flags[tid] = 0u;
    float h_own = 0.f, pre_n = 0.f, gxh2_n = 0.f;
    const unsigned h_off = 4u * j, sv_off = 4u * (4 * j + q);
    // LDS byte addresses (32-bit: the asm reads take them as they are)
    const unsigned hl_base = (unsigned)(size_t)&hl[0][0], fl_base = (unsigned)(size_t)&flags[0];
    const unsigned my_data = hl_base + 64u * q;                       // + 512 (step & 1) + 256 (group)
    const unsigned my_flags = fl_base + 4u * (lane & 3);              // + 16 (group)
    // lane 0 publishes the wave's counter, the other lanes write the same value to words of their own (one ds_write_b32, no exec-mask region)
    const unsigned pub = lane == 0 ? fl_base + 4u * wave : (unsigned)(size_t)&sink[0] + 4u * tid;
    const int nchunks = (S + DF_CH - 1) / DF_CH;
    float4 stg0, stg1, stg2;
#define DF_CHUNK_ROWS(c, n, tlo)                       \
    {                                                  \
        const int s0_ = (c) * DF_CH;                   \
        n = min(DF_CH, S - s0_);                       \
        tlo = dir ? S - s0_ - n : s0_;                 \
    }
#define DF_ISSUE(c)                                                                                \
    {                                                                                              \
        int n_, tlo_;                                                                              \
        DF_CHUNK_ROWS(c, n_, tlo_)                                                                 \
        const float4* src_ = reinterpret_cast<const float4*>(gx + (size_t)tlo_ * GRU_G);           \
        const int lim_ = n_ * (GRU_G / 4);                                                         \
        stg0 = src_[tid < lim_ ? tid : 0];                                                         \
        stg1 = src_[tid + 512 < lim_ ? tid + 512 : 0];                                             \
        stg2 = src_[tid + 1024 < lim_ ? tid + 1024 : 0];                                           \
    }
#define DF_COMMIT(buf)                                                 \
    {                                                                  \
        float4* d_ = reinterpret_cast<float4*>(gxl[buf]);              \
        d_[tid] = stg0; d_[tid + 512] = stg1; d_[tid + 1024] = stg2;   \
    }
    DF_ISSUE(0)
    DF_COMMIT(0)
    __syncthreads();
    unsigned step = 0;
#ifdef DF_STATS
    unsigned long long st_a = 0, st_b = 0, st_r = 0, st_t0 = __builtin_amdgcn_s_memtime();
#endif
    DFT_DECL
    for (int c = 0; c < nchunks; ++c) {
        int n, tlo;
        DF_CHUNK_ROWS(c, n, tlo)
        DF_ISSUE(min(c + 1, nchunks - 1))       // unconditional (see gru.hip)
        const float* gb = gxl[c & 1];
        {   // input terms of the chunk's first step
            const int row = dir ? n - 1 : 0;
            pre_n = (gb[row * GRU_G + zr_off] + bzr) * -L2E;
            gxh2_n = gb[row * GRU_G + 2 * GRU_U + j] * (2.f * L2E);
        }
        for (int i = 0; i < n; ++i) {
            const int row = dir ? n - 1 - i : i;
            const int t = tlo + row;
            const float pre = pre_n, gxh2 = gxh2_n;
            const unsigned par = (step & 1u) * 512u;
            // the NEXT step's two input terms (the chunk's last step re-reads its own row, unused): requested first, landed long before the tail
            const int rown = i + 1 < n ? (dir ? n - 2 - i : i + 1) : row;
            const float gzr_n = gb[rown * GRU_G + zr_off], gxh_n = gb[rown * GRU_G + 2 * GRU_U + j];
            unsigned fv;
            f32x4 a0, a1, a2, a3;
            // ---------------- both halves of h(step) requested at once; phase A on group X's
            DFT(0)
            unsigned fy;
            f32x4 b0, b1, b2, b3;
            df_read_both(my_flags, my_data + par, fv, fy, a0, a1, a2, a3, b0, b1, b2, b3);
            int guard = 0;
            if (!df_ready(fv, step)) {
                while (!df_ready(df_read_flag(my_flags), step) && ++guard < DF_SPIN_LIMIT) {}
                df_read_group(my_flags, my_data + par, fv, a0, a1, a2, a3);
                ++guard;
            }
            DFT_SET(5, guard)
#ifdef DF_STATS
            st_a += guard;
#endif
            DFT(1)
            f32x2 am2 = {0.f, 0.f}, ao2 = {0.f, 0.f}, aha = {0.f, 0.f};
            {
                const f32x4 hv[4] = {a0, a1, a2, a3};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const f32x2 h01 = {hv[k4].x, hv[k4].y}, h23 = {hv[k4].z, hv[k4].w};
                    am2 = pk_fma(h01, u[0][2 * k4], am2); ao2 = pk_fma(h01, u[1][2 * k4], ao2); aha = pk_fma(h01, u[2][2 * k4], aha);
                    am2 = pk_fma(h23, u[0][2 * k4 + 1], am2); ao2 = pk_fma(h23, u[1][2 * k4 + 1], ao2); aha = pk_fma(h23, u[2][2 * k4 + 1], aha);
                }
            }
            // ---------------- phase B: group Y's half (re-read behind phase A's FMAs if its counters were short at the top of the step);
            // z | r first, their sigmoid among the candidate gate's FMAs
            DFT(2)
            guard = 0;
            if (!df_ready(fy, step)) {
                unsigned f1;
                df_read_group_after(my_flags + 16u, my_data + par + 256u, f1, b0, b1, b2, b3, am2, ao2, aha);
                ++guard;
                if (!df_ready(f1, step)) {
                    while (!df_ready(df_read_flag(my_flags + 16u), step) && ++guard < DF_SPIN_LIMIT) {}
                    df_read_group(my_flags + 16u, my_data + par + 256u, f1, b0, b1, b2, b3);
                    ++guard;
                }
            }
            DFT_SET(6, guard)
#ifdef DF_STATS
            st_b += guard; st_r += guard ? 1 : 0;
#endif
            DFT(3)
            a0 = b0; a1 = b1; a2 = b2; a3 = b3;
            const f32x4 hv[4] = {a0, a1, a2, a3};
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const f32x2 h01 = {hv[k4].x, hv[k4].y}, h23 = {hv[k4].z, hv[k4].w};
                am2 = pk_fma(h01, u[0][8 + 2 * k4], am2); ao2 = pk_fma(h01, u[1][8 + 2 * k4], ao2);
                am2 = pk_fma(h23, u[0][8 + 2 * k4 + 1], am2); ao2 = pk_fma(h23, u[1][8 + 2 * k4 + 1], ao2);
            }
            __builtin_amdgcn_sched_barrier(0);
            float zr = (am2.x + am2.y) +
                       __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ao2.x + ao2.y), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
            zr += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(zr), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
            const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(zr, -L2E, pre)));
            const float z = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0xA0 /*quad_perm [0,0,2,2]*/, 0xF, 0xF, true));
            const float r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sg), 0xF5 /*quad_perm [1,1,3,3]*/, 0xF, 0xF, true));
            const float r2 = r * (2.f * L2E), omz = 1.f - z;
            const float ba = fmaf(z, h_own, omz), bb = -2.f * omz;
            f32x2 ahb = {0.f, 0.f};
#pragma unroll
            for (int k4 = 0; k4 < 4; k4 += 2) {
                const f32x2 h01 = {hv[k4].x, hv[k4].y}, h23 = {hv[k4].z, hv[k4].w};
                const f32x2 g01 = {hv[k4 + 1].x, hv[k4 + 1].y}, g23 = {hv[k4 + 1].z, hv[k4 + 1].w};
                aha = pk_fma(h01, u[2][8 + 2 * k4], aha); ahb = pk_fma(g01, u[2][8 + 2 * k4 + 2], ahb);
                aha = pk_fma(h23, u[2][8 + 2 * k4 + 1], aha); ahb = pk_fma(g23, u[2][8 + 2 * k4 + 3], ahb);
            }
            const f32x2 ah2 = aha + ahb;
            float ahs = ah2.x + ah2.y;
            ahs += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ahs), 0xB1, 0xF, 0xF, true));
            ahs += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ahs), 0x4E, 0xF, 0xF, true));
            const float ghh = ahs + bh;
            const float rc = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(r2, ghh, gxh2)));
            const float hn = fmaf(bb, rc, ba);      // z h + (1 - z) (1 - 2 rc)
            h_own = hn;
            // ---------------- publish: h' (the four lanes of a quad write the same word), then this wave's step counter
            hl[(step + 1) & 1][j] = hn;
            asm volatile("ds_write_b32 %0, %1" ::"v"(pub), "v"(step + 1u) : "memory");      // behind the h' write in this wave's LDS queue
            DFT(4)
            // outputs and the next step's input terms: off the chain
            *reinterpret_cast<float*>(reinterpret_cast<char*>(H) + ((unsigned)t * (GRU_U * 4u) + h_off)) = hn;
            pre_n = (gzr_n + bzr) * -L2E;
            gxh2_n = gxh_n * (2.f * L2E);
            if constexpr (SAVE) {
                const float hh = fmaf(rc, -2.f, 1.f);
                const float hi2 = q == 2 ? hh : ghh;
                *reinterpret_cast<float*>(reinterpret_cast<char*>(sv) + ((unsigned)t * (GRU_U * 16u) + sv_off)) = q < 2 ? sg : hi2;      // [t][unit][z r hh gh]
            }
            DFT_STORE(0, step)
            ++step;
        }
        DF_COMMIT((c + 1) & 1)      // the only wait on the staged loads
        lds_barrier();              // once per chunk: the committed rows are visible to every wave (and the groups re-align)
    }
#ifdef DF_STATS
    if (blockIdx.x == 0 && lane == 0) {
        g_df_stats[0][wave][0] = st_a; g_df_stats[0][wave][1] = st_b; g_df_stats[0][wave][2] = st_r;
        g_df_stats[0][wave][3] = __builtin_amdgcn_s_memtime() - st_t0;
    }
#endif
#undef DF_CHUNK_ROWS
#undef DF_ISSUE
#undef DF_COMMIT
}

int launch_gru_fwd_df(hipStream_t st, const float* gx_f, const float* gx_b, const float* U_f, const float* U_b,
                      const float* brec_f, const float* brec_b, float* h_f, float* h_b, float* sv_f, float* sv_b, int B, int S) {
    if (sv_f) hipLaunchKernelGGL((gru_fwd_df_kernel<true>), dim3(2 * B), dim3(512), 0, st, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, sv_f, sv_b, S);
    else hipLaunchKernelGGL((gru_fwd_df_kernel<false>), dim3(2 * B), dim3(512), 0, st, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, sv_f, sv_b, S);
    return 0;
}

// -DDF_TRACE: [wave][step][stamp] of workgroup 0 (which = 0 forward, 1 backward); -2 in a product build
int gru_df_trace_read(int which, unsigned long long* out) {
#ifdef DF_STATS
    hipDeviceSynchronize();
    memset(out, 0, 8 * 16 * 8 * sizeof(unsigned long long));
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_df_stats), 8 * 4 * sizeof(unsigned long long), (size_t)which * 8 * 4 * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost) == hipSuccess ? 1 : -3;
#endif
#ifdef DF_TRACE
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_df_trace), 8 * 16 * 8 * sizeof(unsigned long long), (size_t)which * 8 * 16 * 8 * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
#else
    return -2;
#endif
}
