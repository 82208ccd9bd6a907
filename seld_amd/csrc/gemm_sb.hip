// gemm_sb.hip — split-bf16 GEMM for the row-streamed products of the recurrent block and the heads:
//   GRU input projections   gx = feat * kernel + bias          (modules.py:311-316; M = B*T/5, K = 128, N = 2 x 384)
//   their input gradients   dfeat = dgx_f kernel_f^T + dgx_b kernel_b^T                     (K = 2 x 384, N = 128)
//   first Conv1D(128, 1) of the SED / DOA heads and its input gradient (modules.py:326-346)
// Same numerics as conv_sb.hip: every fp32 operand is split exactly into three bf16 values (x = hi + mid + lo)
// and the product is the 6 leading bf16 MFMA terms accumulated in fp32 (error ~2^-24 relative per product, i.e.
// fp32 level; the parity bar is 1e-4).  v_mfma_f32_32x32x16_bf16 runs 16x the rate of the fp32 MFMA, so 6 of
// them still cost 2.7x less than the exact-fp32 instruction, and a bf16 A fragment is 8 CONTIGUOUS k of one row:
// the fp32 rows of A go global -> registers in MFMA layout (64 B per lane per 32-k chunk) and are split there,
// with no LDS transpose at all.  Only the (small, pre-split) weight operand passes through LDS.
//
// Work split: wave = 32 rows x 128 columns (4 accumulator tiles), workgroup = 4 waves = 128 rows; K is walked in
// chunks of 32; the chunk of B (3 planes x 128 columns x 32 k = 24 KB) is double-buffered in LDS with one LDS-only
// barrier per chunk; the next chunk's A rows and B planes are in flight (registers) while the current one is
// multiplied.  Column groups are the fast grid index so the blocks that share rows of A run together.
#include "common.h"
#include "prep.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define GSB_KC 32
#define GSB_BN 128

// exact 3-way truncation split of two floats, packed as bf16 pairs (element 0 in the low half) — as conv_sb.hip
__device__ __forceinline__ void gsb_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
}

__device__ __forceinline__ unsigned gsb_rne_pair(float x0, float x1) { return bf16_rne_bits(x0) | (bf16_rne_bits(x1) << 16); }

// ---- weight pre-pass ---------------------------------------------------------------------------
// B fp32 ([K,N] if !transb, [N,K] if transb) -> planes [chunk c = k/32][plane][n][piece'][8] bf16 with
// piece' = piece ^ ((n >> 2) & 3): the image of one (chunk, plane, 128-column group) is the 8 KB the GEMM copies
// verbatim into LDS, already swizzled so that each 16-lane group of a ds_read_b128 (lanes {0-3,12-15,20-27}, ...: 64-B rows put
// column n on slot 4 (n & 3) + piece' of the 256-B bank row) covers all 16 slots.
__global__ __launch_bounds__(256) void gemm_split_b_kernel(GemmSplitJobs jobs) { gemm_split_b_body(jobs, blockIdx.y, blockIdx.x, gridDim.x); }

size_t gemm_sb_split_elems(int K, int N) { return (size_t)3 * K * N; }   // bf16 elements of one pre-split operand

int gemm_sb_usable(const void* A, int lda, int N, int K) {
    return (K % GSB_KC) == 0 && (N % GSB_BN) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
}

int launch_gemm_split_b(hipStream_t st, int njobs, const float* const* src, unsigned short* const* dst, const int* ldb,
                        const int* transb, const int* K, const int* N) {
    if (njobs <= 0 || njobs > GSB_MAX_JOBS) return -1;
    GemmSplitJobs j;
    j.njobs = njobs;
    j.one = g_mfma_one;
    for (int i = 0; i < njobs; ++i) {
        if (K[i] % GSB_KC) return -2;
        j.src[i] = src[i]; j.dst[i] = dst[i]; j.ldb[i] = ldb[i]; j.transb[i] = transb[i]; j.K[i] = K[i]; j.N[i] = N[i];
    }
    hipLaunchKernelGGL(gemm_split_b_kernel, dim3(144, njobs), dim3(256), 0, st, j);
    return 0;
}

// ---- the product -------------------------------------------------------------------------------
// mode 0: C0 = act(A0 B0 + bias0).
// mode 1 (two products sharing A): column groups >= N/128 compute C1 = act(A0 B1 + bias1).
// mode 2 (one product over a concatenated K axis): C0 = act(A0 B0 + A1 B1 + bias0).
// ONE: bf16 single-product mode (common.h g_mfma_one): A rounded to nearest bf16 in registers, B's plane 0 holds the rounded weights
// (gemm_split_b with jobs.one); one MFMA per k-step and column tile.  The B staging still moves all three planes (untouched code path).
// FOUR (option "bwd_four_products", set for a launch by BwdFourScope — backward products only): the two products with a lo factor are dropped, plane 2 of B is
// not staged, A is split in two planes with a rounded mid (common.h split2r_pair; gemm_split_b rounds the mid plane of every transposed / flipped B likewise)
template <int NG, bool ONE = false, bool FOUR = false>   // NG 4: K = 128, one product (or two sharing A): every A row of the tile is requested up front; 0: runtime loop
__global__ __launch_bounds__(256, NG == 4 ? 2 : 3) void gemm_sb_kernel(const float* __restrict__ A0, const float* __restrict__ A1, int lda,
                                                         const unsigned short* __restrict__ Bs0,
                                                         const unsigned short* __restrict__ Bs1,
                                                         const float* __restrict__ bias0, const float* __restrict__ bias1,
                                                         float* __restrict__ C0, float* __restrict__ C1, int ldc, int M, int N,
                                                         int K, int act, int mode, int dbg, int accum, GemmEpi epi) {
    __shared__ __attribute__((aligned(16))) unsigned short Bl[2][3 * GSB_BN * GSB_KC];   // 2 x 24 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 5, li = lane & 31;
    const int ngrp = N / GSB_BN;
    int grp = blockIdx.x;
    const float* bias = bias0;
    float* C = C0;
    if (mode == 1 && grp >= ngrp) { grp -= ngrp; Bs0 = Bs1; bias = bias1; C = C1; }
    const int n0 = grp * GSB_BN;
    const int m0 = blockIdx.y * 128 + wave * 32;
    const int nck = K / GSB_KC, ng = mode == 2 ? 2 * nck : nck;
    // this lane's row of A (rows past M read the last row; their results are not stored)
    const int arow = min(m0 + li, M - 1);
    const size_t aoff = (size_t)arow * lda + 16 * kg;

    f32x16 acc0 = zero16(), acc1 = zero16(), acc2 = zero16(), acc3 = zero16();
    float4 ca0, ca1, ca2, ca3, na0, na1, na2, na3;    // the chunk being multiplied / the chunk in flight (16 k each)
    u32x4 bq0, bq1, bq2, bq3, bq4, bq5;               // B chunk in flight: 3 planes x 2 pieces of 16 B per thread
    // the (chunk g, plane p) image of this column group: 128 columns x 64 B, contiguous
#define GSB_ISSUE(g_)                                                                                          \
    {                                                                                                          \
        const int h_ = (g_) >= nck ? 1 : 0, gc_ = (g_) - h_ * nck;                                             \
        const float4* ap_ = reinterpret_cast<const float4*>((h_ ? A1 : A0) + aoff + (size_t)gc_ * GSB_KC);     \
        na0 = ap_[0]; na1 = ap_[1]; na2 = ap_[2]; na3 = ap_[3];                                                \
        const u32x4* bp_ = reinterpret_cast<const u32x4*>((h_ ? Bs1 : Bs0) + ((size_t)gc_ * 3 * N + n0) * GSB_KC); \
        const size_t ps_ = (size_t)N * GSB_KC / 8;  /* plane stride in 16-B units */                           \
        bq0 = bp_[tid]; bq1 = bp_[tid + 256];                                                                  \
        bq2 = bp_[ps_ + tid]; bq3 = bp_[ps_ + tid + 256];                                                      \
        if (!FOUR) { bq4 = bp_[2 * ps_ + tid]; bq5 = bp_[2 * ps_ + tid + 256]; }                               \
    }
#define GSB_COMMIT(buf_)                                                          \
    {                                                                             \
        u32x4* d_ = reinterpret_cast<u32x4*>(Bl[buf_]);                           \
        d_[tid] = bq0; d_[tid + 256] = bq1;                                       \
        d_[512 + tid] = bq2; d_[512 + tid + 256] = bq3;                           \
        if (!FOUR) { d_[1024 + tid] = bq4; d_[1024 + tid + 256] = bq5; }          \
    }
    GSB_ISSUE(0)
    GSB_COMMIT(0)
    ca0 = na0; ca1 = na1; ca2 = na2; ca3 = na3;
    lds_barrier();
    // fragment of column (nt*32 + li), k-step s: piece (2 kg + s) ^ ((li >> 2) & 3) of its 64-B row
    const int swz = (li >> 2) & 3;
    const int boff0 = li * GSB_KC + (((2 * kg) ^ swz) << 3), boff1 = li * GSB_KC + (((2 * kg + 1) ^ swz) << 3);
#define GSB_MFMA(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC_, 0, 0, 0)
#define GSB_STEP(x0_, x1_, boff_)                                                                     \
    if (ONE) {                                                                                        \
        const u32x4 h_ = {gsb_rne_pair(x0_.x, x0_.y), gsb_rne_pair(x0_.z, x0_.w), gsb_rne_pair(x1_.x, x1_.y), gsb_rne_pair(x1_.z, x1_.w)}; \
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h_);                                             \
        const unsigned short* bb_ = bl + boff_;                                                       \
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(bb_), bh1 = *reinterpret_cast<const bf16x8*>(bb_ + 32 * GSB_KC), \
                     bh2 = *reinterpret_cast<const bf16x8*>(bb_ + 64 * GSB_KC), bh3 = *reinterpret_cast<const bf16x8*>(bb_ + 96 * GSB_KC); \
        GSB_MFMA(ah, bh0, acc0); GSB_MFMA(ah, bh1, acc1); GSB_MFMA(ah, bh2, acc2); GSB_MFMA(ah, bh3, acc3);   \
    } else if (FOUR) {                                                                                \
        unsigned h0_, h1_, h2_, h3_, m0_, m1_, m2_, m3_;                                              \
        split2r_pair(x0_.x, x0_.y, h0_, m0_);                                                         \
        split2r_pair(x0_.z, x0_.w, h1_, m1_);                                                         \
        split2r_pair(x1_.x, x1_.y, h2_, m2_);                                                         \
        split2r_pair(x1_.z, x1_.w, h3_, m3_);                                                         \
        const u32x4 h_ = {h0_, h1_, h2_, h3_}, m_ = {m0_, m1_, m2_, m3_};                             \
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h_), am = __builtin_bit_cast(bf16x8, m_);        \
        const unsigned short* bb_ = bl + boff_;                                                       \
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(bb_), bh1 = *reinterpret_cast<const bf16x8*>(bb_ + 32 * GSB_KC), \
                     bh2 = *reinterpret_cast<const bf16x8*>(bb_ + 64 * GSB_KC), bh3 = *reinterpret_cast<const bf16x8*>(bb_ + 96 * GSB_KC); \
        const unsigned short* bm_ = bb_ + GSB_BN * GSB_KC;                                            \
        const bf16x8 bm0 = *reinterpret_cast<const bf16x8*>(bm_), bm1 = *reinterpret_cast<const bf16x8*>(bm_ + 32 * GSB_KC), \
                     bm2 = *reinterpret_cast<const bf16x8*>(bm_ + 64 * GSB_KC), bm3 = *reinterpret_cast<const bf16x8*>(bm_ + 96 * GSB_KC); \
        GSB_MFMA(ah, bh0, acc0); GSB_MFMA(ah, bh1, acc1); GSB_MFMA(ah, bh2, acc2); GSB_MFMA(ah, bh3, acc3);   /* hi*hi  */ \
        GSB_MFMA(ah, bm0, acc0); GSB_MFMA(ah, bm1, acc1); GSB_MFMA(ah, bm2, acc2); GSB_MFMA(ah, bm3, acc3);   /* hi*mid */ \
        GSB_MFMA(am, bh0, acc0); GSB_MFMA(am, bh1, acc1); GSB_MFMA(am, bh2, acc2); GSB_MFMA(am, bh3, acc3);   /* mid*hi */ \
        GSB_MFMA(am, bm0, acc0); GSB_MFMA(am, bm1, acc1); GSB_MFMA(am, bm2, acc2); GSB_MFMA(am, bm3, acc3);   /* mid*mid */ \
    } else                                                                                            \
    {                                                                                                 \
        unsigned h0_, h1_, h2_, h3_, m0_, m1_, m2_, m3_, l0_, l1_, l2_, l3_;                          \
        gsb_split3_pair(x0_.x, x0_.y, h0_, m0_, l0_);                                                 \
        gsb_split3_pair(x0_.z, x0_.w, h1_, m1_, l1_);                                                 \
        gsb_split3_pair(x1_.x, x1_.y, h2_, m2_, l2_);                                                 \
        gsb_split3_pair(x1_.z, x1_.w, h3_, m3_, l3_);                                                 \
        const u32x4 h_ = {h0_, h1_, h2_, h3_}, m_ = {m0_, m1_, m2_, m3_}, l_ = {l0_, l1_, l2_, l3_};  \
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h_), am = __builtin_bit_cast(bf16x8, m_),        \
                     al = __builtin_bit_cast(bf16x8, l_);                                             \
        const unsigned short* bb_ = bl + boff_;                                                       \
        const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(bb_), bh1 = *reinterpret_cast<const bf16x8*>(bb_ + 32 * GSB_KC), \
                     bh2 = *reinterpret_cast<const bf16x8*>(bb_ + 64 * GSB_KC), bh3 = *reinterpret_cast<const bf16x8*>(bb_ + 96 * GSB_KC); \
        const unsigned short* bm_ = bb_ + GSB_BN * GSB_KC;                                            \
        const bf16x8 bm0 = *reinterpret_cast<const bf16x8*>(bm_), bm1 = *reinterpret_cast<const bf16x8*>(bm_ + 32 * GSB_KC), \
                     bm2 = *reinterpret_cast<const bf16x8*>(bm_ + 64 * GSB_KC), bm3 = *reinterpret_cast<const bf16x8*>(bm_ + 96 * GSB_KC); \
        const unsigned short* bl_ = bm_ + GSB_BN * GSB_KC;                                            \
        const bf16x8 bl0 = *reinterpret_cast<const bf16x8*>(bl_), bl1 = *reinterpret_cast<const bf16x8*>(bl_ + 32 * GSB_KC), \
                     bl2 = *reinterpret_cast<const bf16x8*>(bl_ + 64 * GSB_KC), bl3 = *reinterpret_cast<const bf16x8*>(bl_ + 96 * GSB_KC); \
        GSB_MFMA(ah, bh0, acc0); GSB_MFMA(ah, bh1, acc1); GSB_MFMA(ah, bh2, acc2); GSB_MFMA(ah, bh3, acc3);   /* hi*hi  */ \
        GSB_MFMA(ah, bm0, acc0); GSB_MFMA(ah, bm1, acc1); GSB_MFMA(ah, bm2, acc2); GSB_MFMA(ah, bm3, acc3);   /* hi*mid */ \
        GSB_MFMA(am, bh0, acc0); GSB_MFMA(am, bh1, acc1); GSB_MFMA(am, bh2, acc2); GSB_MFMA(am, bh3, acc3);   /* mid*hi */ \
        GSB_MFMA(ah, bl0, acc0); GSB_MFMA(ah, bl1, acc1); GSB_MFMA(ah, bl2, acc2); GSB_MFMA(ah, bl3, acc3);   /* hi*lo  */ \
        GSB_MFMA(al, bh0, acc0); GSB_MFMA(al, bh1, acc1); GSB_MFMA(al, bh2, acc2); GSB_MFMA(al, bh3, acc3);   /* lo*hi  */ \
        GSB_MFMA(am, bm0, acc0); GSB_MFMA(am, bm1, acc1); GSB_MFMA(am, bm2, acc2); GSB_MFMA(am, bm3, acc3);   /* mid*mid */ \
    }
    if constexpr (NG == 4) {
        // K = 128 = 4 chunks.  A chunk's 48 MFMAs (0.64 us) are shorter than a global load's round trip (~2 us under load),
        // so with one chunk of prefetch each of the 4 chunks waited for its rows.  All 16 float4 of the lane's row are in
        // flight from the start instead, and the B chunks two ahead (statically named sets; fully unrolled so that the
        // waits are counted vmcnt(N), not the vmcnt(0) hipcc emits across a loop back-edge).
        float4 pa[4][4];
        u32x4 pb[2][6];
#define GSB_LDA(c_) { const float4* ap_ = reinterpret_cast<const float4*>(A0 + aoff + (size_t)(c_) * GSB_KC); \
                      pa[c_][0] = ap_[0]; pa[c_][1] = ap_[1]; pa[c_][2] = ap_[2]; pa[c_][3] = ap_[3]; }
#define GSB_LDB(c_, set_) { const u32x4* bp_ = reinterpret_cast<const u32x4*>(Bs0 + ((size_t)(c_) * 3 * N + n0) * GSB_KC); \
                            const size_t ps_ = (size_t)N * GSB_KC / 8;                                                       \
                            pb[set_][0] = bp_[tid]; pb[set_][1] = bp_[tid + 256]; pb[set_][2] = bp_[ps_ + tid];              \
                            pb[set_][3] = bp_[ps_ + tid + 256]; if (!FOUR) { pb[set_][4] = bp_[2 * ps_ + tid]; pb[set_][5] = bp_[2 * ps_ + tid + 256]; } }
#define GSB_STB(buf_, set_) { u32x4* d_ = reinterpret_cast<u32x4*>(Bl[buf_]);                                  \
                              d_[tid] = pb[set_][0]; d_[tid + 256] = pb[set_][1]; d_[512 + tid] = pb[set_][2]; \
                              d_[512 + tid + 256] = pb[set_][3]; if (!FOUR) { d_[1024 + tid] = pb[set_][4]; d_[1024 + tid + 256] = pb[set_][5]; } }
        // chunk 0 of A and B came through the generic prologue (ca*, Bl[0]); request the rest
        GSB_LDB(1, 1)
        GSB_LDA(1) GSB_LDA(2) GSB_LDA(3)
        pa[0][0] = ca0; pa[0][1] = ca1; pa[0][2] = ca2; pa[0][3] = ca3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 2 < 4) GSB_LDB(g + 2, g & 1)        // its set was stored to LDS one iteration ago (chunk 0: in the prologue)
            __builtin_amdgcn_sched_barrier(0);
            const unsigned short* bl = Bl[g & 1];
            GSB_STEP(pa[g][0], pa[g][1], boff0)
            GSB_STEP(pa[g][2], pa[g][3], boff1)
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < 4) GSB_STB((g + 1) & 1, (g + 1) & 1)
            lds_barrier();
        }
#undef GSB_LDA
#undef GSB_LDB
#undef GSB_STB
    } else
    for (int g = 0; g < ng; ++g) {
        // always issue (the last chunk re-reads itself): a conditional issue would make the in-flight registers a phi and
        // put the wait for the loads right at the merge
        if (!(dbg & 1)) GSB_ISSUE(min(g + 1, ng - 1))
        __builtin_amdgcn_sched_barrier(0);   // keep the issue up here: the scheduler otherwise sinks the loads to their use
        const unsigned short* bl = Bl[g & 1];
        GSB_STEP(ca0, ca1, boff0)
        GSB_STEP(ca2, ca3, boff1)
        __builtin_amdgcn_sched_barrier(0);
        GSB_COMMIT((g + 1) & 1)
        ca0 = na0; ca1 = na1; ca2 = na2; ca3 = na3;
        lds_barrier();
    }
#undef GSB_ISSUE
#undef GSB_COMMIT
#undef GSB_STEP
    // epilogue.  The bias values are fetched before the first store: on gfx950 stores count in vmcnt too, so a load placed
    // between the store groups would wait for every store issued before it.
    const float bv0 = bias ? bias[n0 + li] : 0.f, bv1 = bias ? bias[n0 + 32 + li] : 0.f, bv2 = bias ? bias[n0 + 64 + li] : 0.f,
                bv3 = bias ? bias[n0 + 96 + li] : 0.f;
    if (dbg & 2) return;
    const bool full = m0 + 32 <= M;   // wave-uniform: every block but the last takes the unguarded stores
    float* crow = C + (size_t)(m0 + 4 * kg) * ldc + n0 + li;
    const int rows_left = M - m0 - 4 * kg;   // rows of this lane's stripe that exist
    if (epi.stat_part) {
        // BatchNorm statistics of this 128 x 128 tile (common.h GemmEpi): a lane holds 16 rows of one column per accumulator tile; the two
        // row halves of a column meet through a lane swap, the four waves through LDS (the B buffers are free: the last chunk's barrier is
        // behind us), fixed order.  One [sum | sum of squares] partial per (row block, column).
        float* red = reinterpret_cast<float*>(&Bl[0][0]);
#define GSB_STAT(ACC_, nt_)                                                                  \
        {                                                                                    \
            float s_ = 0.f, q_ = 0.f;                                                        \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                 \
                const int dr = (r & 3) + 8 * (r >> 2);                                       \
                const float v = (full || dr < rows_left) ? ACC_[r] : 0.f;                    \
                s_ += v; q_ = fmaf(v, v, q_);                                                \
            }                                                                                \
            s_ += __shfl_xor(s_, 32); q_ += __shfl_xor(q_, 32);                              \
            if (kg == 0) { red[(wave * 128 + (nt_) * 32 + li) * 2] = s_; red[(wave * 128 + (nt_) * 32 + li) * 2 + 1] = q_; } \
        }
        GSB_STAT(acc0, 0) GSB_STAT(acc1, 1) GSB_STAT(acc2, 2) GSB_STAT(acc3, 3)
#undef GSB_STAT
        lds_barrier();
        if (tid < 128) {
            const float S_ = (red[tid * 2] + red[(128 + tid) * 2]) + (red[(256 + tid) * 2] + red[(384 + tid) * 2]);
            const float Q_ = (red[tid * 2 + 1] + red[(128 + tid) * 2 + 1]) + (red[(256 + tid) * 2 + 1] + red[(384 + tid) * 2 + 1]);
            const int gc = n0 + tid;
            float* pp = epi.stat_part + ((size_t)(gc >> 6) * gridDim.y + blockIdx.y) * 128 + (gc & 63);
            pp[0] = S_; pp[64] = Q_;
        }
    }
    if (epi.addg) {
        // C += addg [gate bit]: the identity shortcut's gated gradient (one dword + one byte load per element, whole 128-B segments per row)
#define GSB_ADDG(ACC_, nt_)                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                     \
            const int dr = (r & 3) + 8 * (r >> 2);                                           \
            if (full || dr < rows_left) {                                                    \
                const size_t e_ = (size_t)(m0 + 4 * kg + dr) * ldc + n0 + (nt_) * 32 + li;   \
                const unsigned g_ = epi.gate4[e_ >> 2];                                      \
                ACC_[r] += ((g_ >> (e_ & 3)) & 1u) ? epi.addg[e_] : 0.f;                     \
            }                                                                                \
        }
        GSB_ADDG(acc0, 0) GSB_ADDG(acc1, 1) GSB_ADDG(acc2, 2) GSB_ADDG(acc3, 3)
#undef GSB_ADDG
    }
#define GSB_STORE(ACC_, nt_, BV_)                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
        const int dr = (r & 3) + 8 * (r >> 2);                                               \
        float v = ACC_[r] + BV_;                                                             \
        if (act == 1) v = 1.f / (1.f + expf(-v));                                            \
        else if (act == 2) v = tanhf(v);                                                     \
        else if (act == 3) v = fmaxf(v, 0.f);                                                \
        if (full || dr < rows_left) {                                                        \
            if (accum) v += crow[(size_t)dr * ldc + (nt_) * 32];                             \
            crow[(size_t)dr * ldc + (nt_) * 32] = v;                                         \
        }                                                                                    \
    }
    if (act == 0 && full && !accum) {
#define GSB_STORE_PLAIN(ACC_, nt_, BV_)                                                      \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) crow[(size_t)((r & 3) + 8 * (r >> 2)) * ldc + (nt_) * 32] = ACC_[r] + BV_;
        GSB_STORE_PLAIN(acc0, 0, bv0) GSB_STORE_PLAIN(acc1, 1, bv1) GSB_STORE_PLAIN(acc2, 2, bv2) GSB_STORE_PLAIN(acc3, 3, bv3)
#undef GSB_STORE_PLAIN
    } else {
        GSB_STORE(acc0, 0, bv0) GSB_STORE(acc1, 1, bv1) GSB_STORE(acc2, 2, bv2) GSB_STORE(acc3, 3, bv3)
    }
#undef GSB_STORE
}

// ---- variant with both operands in LDS ----------------------------------------------------------
// Workgroup = 16 waves = 128 rows x 128 columns, wave (wr, wc) = one 32x32 tile.  The A chunk (128 rows x 32 k) is read
// as one float4 per thread (8 rows x 128 B per wave: whole cache lines), split there and stored as three bf16 planes in
// the same swizzled [row][4 pieces x 16 B] image as B.  With N = 128 the model has only 600 row tiles for 1024 SIMDs, so
// the one-wave-per-row-tile form above leaves a lone wave per SIMD that exposes every load and every split; here each
// SIMD interleaves 4 waves and the staging work per thread is 4 elements per chunk.  Measured (tools/tune_gemm.py,
// M = 19200): K = 2 x 384, N = 128: 37 us against 52 us; K = 128, N = 2 x 384: 38 us against 35 us — the launcher
// picks by the number of column groups.  (Prefetching three chunks ahead and placing the next chunk's split / LDS stores
// in the MFMA gaps with sched_group_barrier were both measured slower than this plain form.)
template <int NG, bool CONV = false, bool ONE = false, bool FOUR = false>   // FOUR: see gemm_sb_kernel; chunks of 32 k when known at compile time (fully unrolled, 3 chunks of loads in flight), 0: runtime loop
__global__ __launch_bounds__(1024) void gemm_sb16_kernel(const float* __restrict__ A0, const float* __restrict__ A1, int lda,
                                                         const unsigned short* __restrict__ Bs0,
                                                         const unsigned short* __restrict__ Bs1,
                                                         const float* __restrict__ bias0, const float* __restrict__ bias1,
                                                         float* __restrict__ C0, float* __restrict__ C1, int ldc, int M, int N,
                                                         int K, int act, int mode, int accum, int cvs, int cvH, int cvW, GemmEpi epi) {
    __shared__ __attribute__((aligned(16))) unsigned short Al[2][3 * 128 * GSB_KC];      // 2 x 24 KB
    __shared__ __attribute__((aligned(16))) unsigned short Bl[2][3 * GSB_BN * GSB_KC];   // 2 x 24 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 5, li = lane & 31, wr = wave >> 2, wc = wave & 3;
    const int ngrp = N / GSB_BN;
    int grp = blockIdx.x;
    const float* bias = bias0;
    float* C = C0;
    if (mode == 1 && grp >= ngrp) { grp -= ngrp; Bs0 = Bs1; bias = bias1; C = C1; }
    const int n0 = grp * GSB_BN, m0 = blockIdx.y * 128;
    const int nck = K / GSB_KC, ng = mode == 2 ? 2 * nck : nck;
    // staging role: row srow, floats [4 sk, 4 sk + 4) of the chunk
    const int srow = tid >> 3, sk = tid & 7;
    const int srowg = min(m0 + srow, M - 1);
    const size_t aoff = (size_t)srowg * lda + 4 * sk;
    // CONV (cvs = log2 C): A0 is the NHWC image [M pixels][C], row m of the virtual im2col matrix [M][9 C] is the 3 x 3 neighbourhood of
    // pixel m (zeros outside the H x W image); a 32-k chunk lies inside one tap because C % 32 == 0
    const int cvt = CONV ? (srowg / cvW) % cvH : 0, cvf = CONV ? srowg % cvW : 0;
    const int a_dst = srow * GSB_KC + ((((sk >> 1) ^ (srow >> 2)) & 3) << 3) + 4 * (sk & 1);   // bf16 index inside a plane
    const size_t bplane = (size_t)N * GSB_KC / 8;   // plane stride of the pre-split B, in 16-B units
    float4 na;
    u32x4 nb0, nb1;
#define G16_ISSUE(g_)                                                                                           \
    {                                                                                                           \
        const int h_ = (g_) >= nck ? 1 : 0, gc_ = (g_) - h_ * nck;                                              \
        if (CONV) {                                                                                             \
            const int k0_ = gc_ * GSB_KC, tap_ = k0_ >> cvs, dy_ = tap_ / 3 - 1, dx_ = tap_ - 3 * (tap_ / 3) - 1;  \
            const bool ok_ = (unsigned)(cvt + dy_) < (unsigned)cvH && (unsigned)(cvf + dx_) < (unsigned)cvW;     \
            const float4 t_ = *reinterpret_cast<const float4*>(                                                 \
                A0 + (ok_ ? (((size_t)(srowg + dy_ * cvW + dx_)) << cvs) + (k0_ - (tap_ << cvs)) + 4 * sk : 0));    \
            na = ok_ ? t_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
        } else                                                                                                  \
        na = *reinterpret_cast<const float4*>((h_ ? A1 : A0) + aoff + (size_t)gc_ * GSB_KC);                    \
        const u32x4* bp_ = reinterpret_cast<const u32x4*>((h_ ? Bs1 : Bs0) + ((size_t)gc_ * 3 * N + n0) * GSB_KC); \
        nb0 = bp_[(tid >> 9) * bplane + (tid & 511)];            /* planes 0 and 1 */                           \
        if (!FOUR) nb1 = bp_[2 * bplane + (tid & 511)];          /* plane 2: used by the first 512 threads */   \
    }
#define G16_COMMIT(buf_)                                                                                        \
    {                                                                                                           \
        unsigned short* ad_ = Al[buf_] + a_dst;                                                                 \
        if (ONE) {                                                                                              \
            *reinterpret_cast<uint2*>(ad_) = make_uint2(gsb_rne_pair(na.x, na.y), gsb_rne_pair(na.z, na.w));    \
        } else if (FOUR) {                                                                                      \
        unsigned h0_, m0_, h1_, m1_;                                                                            \
        split2r_pair(na.x, na.y, h0_, m0_);                                                                     \
        split2r_pair(na.z, na.w, h1_, m1_);                                                                     \
        *reinterpret_cast<uint2*>(ad_) = make_uint2(h0_, h1_);                                                  \
        *reinterpret_cast<uint2*>(ad_ + 128 * GSB_KC) = make_uint2(m0_, m1_);                                   \
        } else {                                                                                                \
        unsigned h0_, m0_, l0_, h1_, m1_, l1_;                                                                  \
        gsb_split3_pair(na.x, na.y, h0_, m0_, l0_);                                                             \
        gsb_split3_pair(na.z, na.w, h1_, m1_, l1_);                                                             \
        *reinterpret_cast<uint2*>(ad_) = make_uint2(h0_, h1_);                                                  \
        *reinterpret_cast<uint2*>(ad_ + 128 * GSB_KC) = make_uint2(m0_, m1_);                                   \
        *reinterpret_cast<uint2*>(ad_ + 2 * 128 * GSB_KC) = make_uint2(l0_, l1_);                               \
        }                                                                                                       \
        u32x4* bd_ = reinterpret_cast<u32x4*>(Bl[buf_]);                                                        \
        bd_[tid] = nb0;                                                                                         \
        if (!FOUR && tid < 512) bd_[1024 + tid] = nb1;                                                          \
    }
    G16_ISSUE(0)
    G16_COMMIT(0)
    lds_barrier();
    f32x16 acc0 = zero16(), acc1 = zero16();   // two chains: 3 of the 6 products each
    // fragment (row or column r, k-step s): piece (2 s + kg) ^ ((r >> 2) & 3) of its 64-B row; r = tile * 32 + li
    const int swz = (li >> 2) & 3;
    const int foff0 = li * GSB_KC + ((kg ^ swz) << 3), foff1 = li * GSB_KC + (((2 + kg) ^ swz) << 3);
    const int aofs = wr * 32 * GSB_KC, bofs = wc * 32 * GSB_KC;
#define G16_STEP(foff_)                                                                                         \
    {                                                                                                           \
        const unsigned short* ap_ = al + aofs + foff_;                                                          \
        const unsigned short* bp_ = bl + bofs + foff_;                                                          \
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap_), bh = *reinterpret_cast<const bf16x8*>(bp_);    \
        GSB_MFMA(ah, bh, acc0);                                                                                 \
        if (FOUR) {                                                                                             \
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap_ + 128 * GSB_KC);                                 \
        const bf16x8 bm = *reinterpret_cast<const bf16x8*>(bp_ + GSB_BN * GSB_KC);                              \
        GSB_MFMA(ah, bm, acc1); GSB_MFMA(am, bh, acc0); GSB_MFMA(am, bm, acc1);                                 \
        } else if (!ONE) {                                                                                      \
        const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap_ + 128 * GSB_KC),                                 \
                     al_ = *reinterpret_cast<const bf16x8*>(ap_ + 2 * 128 * GSB_KC);                            \
        const bf16x8 bm = *reinterpret_cast<const bf16x8*>(bp_ + GSB_BN * GSB_KC),                              \
                     bl_ = *reinterpret_cast<const bf16x8*>(bp_ + 2 * GSB_BN * GSB_KC);                         \
        GSB_MFMA(ah, bm, acc1); GSB_MFMA(am, bh, acc0);                                                         \
        GSB_MFMA(ah, bl_, acc1); GSB_MFMA(al_, bh, acc0); GSB_MFMA(am, bm, acc1);                               \
        }                                                                                                       \
    }
    if constexpr (NG == 0) {
    for (int g = 0; g < ng; ++g) {
        G16_ISSUE(min(g + 1, ng - 1))
        __builtin_amdgcn_sched_barrier(0);
        const unsigned short* al = Al[g & 1];
        const unsigned short* bl = Bl[g & 1];
        G16_STEP(foff0)
        G16_STEP(foff1)
        __builtin_amdgcn_sched_barrier(0);
        G16_COMMIT((g + 1) & 1)
        lds_barrier();
    }
    } else {
        // One chunk's MFMAs (0.64 us) are shorter than a global load's round trip (~2 us under load): with one chunk of
        // prefetch every chunk waited for its loads (24 chunks x 2 us = the 48 us this kernel took at K = 2 x 384).  Here
        // three chunks are in flight in statically named register sets; the loop is fully unrolled because across a loop
        // back-edge hipcc's waitcnt bookkeeping falls back to vmcnt(0), which would wait for all of them.
        float4 pa[4];
        u32x4 pb0[4], pb1[4];
#define G16_ISSUE_S(g_, set_) { G16_ISSUE(g_) pa[set_] = na; pb0[set_] = nb0; pb1[set_] = nb1; }
#define G16_COMMIT_S(buf_, set_) { na = pa[set_]; nb0 = pb0[set_]; nb1 = pb1[set_]; G16_COMMIT(buf_) }
        // chunk 0 is already committed (above); request chunks 1..3
        if (NG > 1) G16_ISSUE_S(1, 1)
        if (NG > 2) G16_ISSUE_S(2, 2)
        if (NG > 3) G16_ISSUE_S(3, 3)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const unsigned short* al = Al[g & 1];
            const unsigned short* bl = Bl[g & 1];
            __builtin_amdgcn_sched_barrier(0);
            G16_STEP(foff0)
            G16_STEP(foff1)
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) G16_COMMIT_S((g + 1) & 1, (g + 1) & 3)       // waits for chunk g + 1 only: g + 2, g + 3 stay in flight
            if (g + 4 < NG) G16_ISSUE_S(g + 4, g & 3)                    // into the set chunk g + 0 ... was committed from
            lds_barrier();
        }
#undef G16_ISSUE_S
#undef G16_COMMIT_S
    }
#undef G16_ISSUE
#undef G16_COMMIT
#undef G16_STEP
    const int col = n0 + wc * 32 + li;
    const float bv = bias ? bias[col] : 0.f;
    const int rbase = m0 + wr * 32 + 4 * kg;
    float* crow = C + (size_t)rbase * ldc + col;
    const int rows_left = M - rbase;
    if (epi.stat_part) {
        // BatchNorm statistics of this 128 x 128 tile (see gemm_sb_kernel): wave (wr, wc) holds a 32 x 32 tile; row halves by a lane swap, the
        // four row waves of a column through LDS (the A buffers are free behind the last chunk's barrier), fixed order
        float* red = reinterpret_cast<float*>(&Al[0][0]);
        float s_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const float v = dr < rows_left ? acc0[r] + acc1[r] : 0.f;
            s_ += v; q_ = fmaf(v, v, q_);
        }
        s_ += __shfl_xor(s_, 32); q_ += __shfl_xor(q_, 32);
        if (kg == 0) { red[(wr * 128 + wc * 32 + li) * 2] = s_; red[(wr * 128 + wc * 32 + li) * 2 + 1] = q_; }
        lds_barrier();
        if (tid < 128) {
            const float S_ = (red[tid * 2] + red[(128 + tid) * 2]) + (red[(256 + tid) * 2] + red[(384 + tid) * 2]);
            const float Q_ = (red[tid * 2 + 1] + red[(128 + tid) * 2 + 1]) + (red[(256 + tid) * 2 + 1] + red[(384 + tid) * 2 + 1]);
            const int gc = n0 + tid;
            float* pp = epi.stat_part + ((size_t)(gc >> 6) * gridDim.y + blockIdx.y) * 128 + (gc & 63);
            pp[0] = S_; pp[64] = Q_;
        }
    }
    if (m0 + wr * 32 + 32 <= M && ((ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0)) {
        // 4 dwordx4 stores per wave instead of 16 dword stores (common.h: quad_transpose4)
        float* cq = C + (size_t)(rbase + (li & 3)) * ldc + n0 + wc * 32 + (li & ~3);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v_[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v_[j] = (acc0[4 * q + j] + acc1[4 * q + j]) + bv;
                if (act == 1) v_[j] = 1.f / (1.f + expf(-v_[j]));
                else if (act == 2) v_[j] = tanhf(v_[j]);
                else if (act == 3) v_[j] = fmaxf(v_[j], 0.f);
            }
            float4 o_ = quad_transpose4(v_[0], v_[1], v_[2], v_[3], li);
            if (accum) {
                const float4 p_ = *reinterpret_cast<const float4*>(cq + (size_t)(8 * q) * ldc);
                o_.x += p_.x; o_.y += p_.y; o_.z += p_.z; o_.w += p_.w;
            }
            if (epi.addg) {      // C += addg [gate bit]: four consecutive columns of one row = one float4 of addg and ONE gate byte
                const size_t e_ = (size_t)(cq - C) + (size_t)(8 * q) * ldc;
                const float4 g_ = *reinterpret_cast<const float4*>(epi.addg + e_);
                const unsigned b_ = epi.gate4[e_ >> 2];
                o_.x += (b_ & 1) ? g_.x : 0.f; o_.y += (b_ & 2) ? g_.y : 0.f; o_.z += (b_ & 4) ? g_.z : 0.f; o_.w += (b_ & 8) ? g_.w : 0.f;
            }
            *reinterpret_cast<float4*>(cq + (size_t)(8 * q) * ldc) = o_;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        float v = (acc0[r] + acc1[r]) + bv;
        if (act == 1) v = 1.f / (1.f + expf(-v));
        else if (act == 2) v = tanhf(v);
        
        else if (act == 3) v = fmaxf(v, 0.f);
        if (dr < rows_left && epi.addg) {
            const size_t e_ = (size_t)(rbase + dr) * ldc + col;
            v += ((epi.gate4[e_ >> 2] >> (e_ & 3)) & 1u) ? epi.addg[e_] : 0.f;
        }
        if (dr < rows_left) crow[(size_t)dr * ldc] = accum ? v + crow[(size_t)dr * ldc] : v;
    }
}

// ---- K = 128, B stationary (round 5) -----------------------------------------------------------------
// The K = 128 products (GRU input projections: M = 19 200, N = 2 x 384; resnet50_block's 128 -> N convolutions) spend their time around the
// MFMAs in gemm_sb_kernel<4>: every 128 x 128 tile re-stages its 96 KB of B through a double buffer (4 barriers), and a workgroup's prologue
// (first loads) and epilogue (64 dword stores per lane) overlap nothing of its own.  Here the WHOLE K extent of a column group's B — 4 chunks x
// 3 planes x 8 KB = 96 KB — is copied into LDS once per workgroup and stays; the workgroup (8 waves, one per CU) is persistent and its waves are
// independent from then on: each walks its own 32-row tiles — A rows global -> registers in MFMA layout, split there, 192 MFMAs against the
// resident B, 16 dwordx4 stores (quad transpose) — with the next tile's rows requested chunk by chunk into the registers the current tile has
// just consumed.  No barrier after the prologue.  The order of the products and of the k-steps is gemm_sb_kernel's, so the results are the same
// bits.  MEASURED (M = 19 200, N = 2 x 384): 31.6 us against the tiled form's 32.6 — with 600 row tiles per column group a wave gets 1.8 tiles, all
// waves start in phase, and the first tile's loads (7 us of A through L2, six column groups re-reading it) and the last tile's stores (59 MB, 11 us)
// overlap nothing: base 5.3 + B 1.6 + A 7 + MFMA 11 + stores 11 us add up almost serially in BOTH forms.  Kept as an opt-in (gsb_dbg bit 6).
template <bool FOUR>
__device__ __forceinline__ void sbp_step(const float4& x0, const float4& x1, const unsigned short* bb, f32x16& acc0, f32x16& acc1, f32x16& acc2, f32x16& acc3) {
#define SBP_MFMA(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC_, 0, 0, 0)
#define SBP_B4(p_, n0_, n1_, n2_, n3_)                                                                                           \
    const bf16x8 n0_ = *reinterpret_cast<const bf16x8*>(bb + (p_) * 4096), n1_ = *reinterpret_cast<const bf16x8*>(bb + (p_) * 4096 + 1024), \
                 n2_ = *reinterpret_cast<const bf16x8*>(bb + (p_) * 4096 + 2048), n3_ = *reinterpret_cast<const bf16x8*>(bb + (p_) * 4096 + 3072);
    if (FOUR) {
        unsigned h0, h1, h2, h3, m0, m1, m2, m3;
        split2r_pair(x0.x, x0.y, h0, m0); split2r_pair(x0.z, x0.w, h1, m1);
        split2r_pair(x1.x, x1.y, h2, m2); split2r_pair(x1.z, x1.w, h3, m3);
        const u32x4 h_ = {h0, h1, h2, h3}, m_ = {m0, m1, m2, m3};
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h_), am = __builtin_bit_cast(bf16x8, m_);
        SBP_B4(0, bh0, bh1, bh2, bh3)
        SBP_B4(1, bm0, bm1, bm2, bm3)
        SBP_MFMA(ah, bh0, acc0); SBP_MFMA(ah, bh1, acc1); SBP_MFMA(ah, bh2, acc2); SBP_MFMA(ah, bh3, acc3);
        SBP_MFMA(ah, bm0, acc0); SBP_MFMA(ah, bm1, acc1); SBP_MFMA(ah, bm2, acc2); SBP_MFMA(ah, bm3, acc3);
        SBP_MFMA(am, bh0, acc0); SBP_MFMA(am, bh1, acc1); SBP_MFMA(am, bh2, acc2); SBP_MFMA(am, bh3, acc3);
        SBP_MFMA(am, bm0, acc0); SBP_MFMA(am, bm1, acc1); SBP_MFMA(am, bm2, acc2); SBP_MFMA(am, bm3, acc3);
    } else {
        unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
        gsb_split3_pair(x0.x, x0.y, h0, m0, l0); gsb_split3_pair(x0.z, x0.w, h1, m1, l1);
        gsb_split3_pair(x1.x, x1.y, h2, m2, l2); gsb_split3_pair(x1.z, x1.w, h3, m3, l3);
        const u32x4 h_ = {h0, h1, h2, h3}, m_ = {m0, m1, m2, m3}, l_ = {l0, l1, l2, l3};
        const bf16x8 ah = __builtin_bit_cast(bf16x8, h_), am = __builtin_bit_cast(bf16x8, m_), al = __builtin_bit_cast(bf16x8, l_);
        SBP_B4(0, bh0, bh1, bh2, bh3)
        SBP_B4(1, bm0, bm1, bm2, bm3)
        SBP_B4(2, bl0, bl1, bl2, bl3)
        SBP_MFMA(ah, bh0, acc0); SBP_MFMA(ah, bh1, acc1); SBP_MFMA(ah, bh2, acc2); SBP_MFMA(ah, bh3, acc3);   // hi*hi
        SBP_MFMA(ah, bm0, acc0); SBP_MFMA(ah, bm1, acc1); SBP_MFMA(ah, bm2, acc2); SBP_MFMA(ah, bm3, acc3);   // hi*mid
        SBP_MFMA(am, bh0, acc0); SBP_MFMA(am, bh1, acc1); SBP_MFMA(am, bh2, acc2); SBP_MFMA(am, bh3, acc3);   // mid*hi
        SBP_MFMA(ah, bl0, acc0); SBP_MFMA(ah, bl1, acc1); SBP_MFMA(ah, bl2, acc2); SBP_MFMA(ah, bl3, acc3);   // hi*lo
        SBP_MFMA(al, bh0, acc0); SBP_MFMA(al, bh1, acc1); SBP_MFMA(al, bh2, acc2); SBP_MFMA(al, bh3, acc3);   // lo*hi
        SBP_MFMA(am, bm0, acc0); SBP_MFMA(am, bm1, acc1); SBP_MFMA(am, bm2, acc2); SBP_MFMA(am, bm3, acc3);   // mid*mid
    }
#undef SBP_B4
#undef SBP_MFMA
}

template <bool FOUR>
__global__ __launch_bounds__(512) void gemm_sbp_kernel(const float* __restrict__ A, int lda, const unsigned short* __restrict__ Bs0,
                                                       const unsigned short* __restrict__ Bs1, const float* __restrict__ bias0,
                                                       const float* __restrict__ bias1, float* __restrict__ C0, float* __restrict__ C1, int ldc, int M,
                                                       int N, int ngroups, int dbg) {
    constexpr int NPL = FOUR ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned short sbp_smem[];      // [4 chunks][3 planes][128 columns x 32 k] (FOUR: plane 2 unused)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 5, li = lane & 31;
    const int ngrp = N / GSB_BN;
    int grp = blockIdx.x % ngroups;
    const int wig = blockIdx.x / ngroups, wpg = gridDim.x / ngroups;      // this workgroup among those of its column group
    const unsigned short* Bs = Bs0;
    const float* bias = bias0;
    float* C = C0;
    if (grp >= ngrp) { grp -= ngrp; Bs = Bs1; bias = bias1; C = C1; }
    const int n0 = grp * GSB_BN;
    // B: (chunk c, plane p) of this column group is 8 KB contiguous in the pre-split image, already swizzled
    if (!(dbg & 8))
#pragma unroll
    for (int j = 0; j < 4 * NPL; ++j) {
        const int c = j / NPL, p = j - c * NPL;
        reinterpret_cast<u32x4*>(sbp_smem)[(c * 3 + p) * 512 + tid] =
            reinterpret_cast<const u32x4*>(Bs + ((size_t)(c * 3 + p) * N + n0) * GSB_KC)[tid];
    }
    const int ntile = (M + 31) / 32, slots = wpg * 8;
    int t = wave * wpg + wig;
    float4 pa[4][4];
#define SBP_LDA(tile_)                                                                                            \
    {                                                                                                             \
        const float4* ap_ = reinterpret_cast<const float4*>(A + (size_t)min((tile_) * 32 + li, M - 1) * lda + 16 * kg); \
        _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                                        \
            pa[c_][0] = ap_[8 * c_]; pa[c_][1] = ap_[8 * c_ + 1]; pa[c_][2] = ap_[8 * c_ + 2]; pa[c_][3] = ap_[8 * c_ + 3]; \
        }                                                                                                         \
    }
    if (t < ntile) SBP_LDA(t)
#undef SBP_LDA
    const float4 bv0 = bias ? *reinterpret_cast<const float4*>(bias + n0 + (li & ~3)) : make_float4(0.f, 0.f, 0.f, 0.f),
                 bv1 = bias ? *reinterpret_cast<const float4*>(bias + n0 + 32 + (li & ~3)) : make_float4(0.f, 0.f, 0.f, 0.f),
                 bv2 = bias ? *reinterpret_cast<const float4*>(bias + n0 + 64 + (li & ~3)) : make_float4(0.f, 0.f, 0.f, 0.f),
                 bv3 = bias ? *reinterpret_cast<const float4*>(bias + n0 + 96 + (li & ~3)) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int swz = (li >> 2) & 3;
    const int bo0 = li * GSB_KC + (((2 * kg) ^ swz) << 3), bo1 = li * GSB_KC + (((2 * kg + 1) ^ swz) << 3);
    for (; t < ntile; t += slots) {
        const int m0 = t * 32;
        const int tn = t + slots < ntile ? t + slots : t;      // always request (the last tile re-reads itself): no phi on the staged registers
        const float4* an = reinterpret_cast<const float4*>(A + (size_t)min(tn * 32 + li, M - 1) * lda + 16 * kg);
        f32x16 acc0 = zero16(), acc1 = zero16(), acc2 = zero16(), acc3 = zero16();
        // the resident B is loop-invariant: without this the compiler hoists all 96 fragment reads out of the tile loop (and spills them)
        int o0 = bo0, o1 = bo1;
        asm volatile("" : "+v"(o0), "+v"(o1));
        const unsigned short *b0 = sbp_smem + o0, *b1 = sbp_smem + o1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (!(dbg & 1)) sbp_step<FOUR>(pa[g][0], pa[g][1], b0 + g * 3 * 4096, acc0, acc1, acc2, acc3);
            __builtin_amdgcn_sched_barrier(0);      // one k-step's fragments at a time: two in flight do not fit beside the 64 + 64 registers of acc / A
            if (!(dbg & 1)) sbp_step<FOUR>(pa[g][2], pa[g][3], b1 + g * 3 * 4096, acc0, acc1, acc2, acc3);
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 4)) { pa[g][0] = an[8 * g]; pa[g][1] = an[8 * g + 1]; pa[g][2] = an[8 * g + 2]; pa[g][3] = an[8 * g + 3]; }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dbg & 2) continue;
        // a lane holds column li of rows 4 kg + (r & 3) + 8 (r >> 2): 4 x 4 transposes inside each quad of lanes -> one row, 4 columns per lane
        float* crow = C + (size_t)(m0 + 4 * kg + (li & 3)) * ldc + n0 + (li & ~3);
        const int rows_left = M - m0 - 4 * kg - (li & 3);
        const bool full = m0 + 32 <= M;      // wave-uniform: every tile but a ragged last one takes the unguarded stores
#define SBP_STORE(ACC_, nt_, BV_, GUARD_)                                                                         \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
            float4 o = quad_transpose4(ACC_[4 * q], ACC_[4 * q + 1], ACC_[4 * q + 2], ACC_[4 * q + 3], li);      \
            o.x += BV_.x; o.y += BV_.y; o.z += BV_.z; o.w += BV_.w;                                               \
            if (!(GUARD_) || 8 * q < rows_left) *reinterpret_cast<float4*>(crow + (size_t)(8 * q) * ldc + (nt_) * 32) = o; \
        }
        if (full) { SBP_STORE(acc0, 0, bv0, false) SBP_STORE(acc1, 1, bv1, false) SBP_STORE(acc2, 2, bv2, false) SBP_STORE(acc3, 3, bv3, false) }
        else { SBP_STORE(acc0, 0, bv0, true) SBP_STORE(acc1, 1, bv1, true) SBP_STORE(acc2, 2, bv2, true) SBP_STORE(acc3, 3, bv3, true) }
#undef SBP_STORE
    }
}

int g_gsb_dbg = 0;
thread_local GemmEpi g_gemm_epi;
int gemm_epi_row_blocks(int M, int split_bf16) { return split_bf16 ? (M + 127) / 128 : (M + 63) / 64; }
thread_local int g_gsb_four_now = 0;       // set for the duration of a backward launch (common.h BwdFourScope); per host thread: another thread's forward launch must not read it
int launch_gemm_sb(hipStream_t st, const float* A0, const float* A1, int lda, const unsigned short* Bs0, const unsigned short* Bs1,
                   const float* bias0, const float* bias1, float* C0, float* C1, int ldc, int M, int N, int K, int act, int mode,
                   int accum, int conv_C, int conv_H, int conv_W) {
    int cvs = 0;
    if (conv_C) {      // implicit 3x3: K = 9 C, C a power of two >= 32, M = B * H * W pixels, one product
        while ((1 << cvs) < conv_C) ++cvs;
        if ((1 << cvs) != conv_C || conv_C < 32 || K != 9 * conv_C || mode != 0 || conv_H <= 0 || conv_W <= 0 || M % (conv_H * conv_W)) return -1;
        lda = conv_C;
    }
    if (M <= 0 || N <= 0 || K <= 0 || !gemm_sb_usable(A0, lda, N, K)) return -1;
    if (mode == 2 && !gemm_sb_usable(A1, lda, N, K)) return -1;
    const GemmEpi epi = g_gemm_epi;
    if (epi.stat_part && (bias0 || act || mode || accum)) return -3;
    if (epi.addg && (!epi.gate4 || (ldc & 3) || mode)) return -3;
    dim3 grid(N / GSB_BN * (mode == 1 ? 2 : 1), (M + 127) / 128);
    // few column groups: not enough row tiles to give every SIMD more than one wave -> the 16-wave form
    const bool wide = grid.x >= 2;
    // (an implicit convolution always takes the 16-wave form: its per-chunk address work is shared by 4 column waves there — K = 9 x 256,
    // N = 256: 198 us against 264 for the 4-wave form)
    if (cvs ? false : (g_gsb_dbg & 4) ? true : (g_gsb_dbg & 8) ? false : wide) {
#define GSB_GO(NG_) { if (g_mfma_one) hipLaunchKernelGGL((gemm_sb_kernel<NG_, true>), grid, dim3(256), 0, st, A0, A1, lda, Bs0, Bs1, bias0, bias1, C0, C1, ldc, M, N, K, act, \
                                       mode, g_gsb_dbg & 3, accum, epi);                                                                                            \
                      else if (g_gsb_four_now) hipLaunchKernelGGL((gemm_sb_kernel<NG_, false, true>), grid, dim3(256), 0, st, A0, A1, lda, Bs0, Bs1, bias0, bias1, C0, C1, ldc, M, N, K, act, \
                                       mode, g_gsb_dbg & 3, accum, epi);                                                                                            \
                      else hipLaunchKernelGGL((gemm_sb_kernel<NG_, false>), grid, dim3(256), 0, st, A0, A1, lda, Bs0, Bs1, bias0, bias1, C0, C1, ldc, M, N, K, act, \
                                       mode, g_gsb_dbg & 3, accum, epi); }
        // K = 128, plain epilogue: B stationary in LDS, persistent workgroups (gemm_sbp_kernel) — an opt-in (gsb_dbg bit 6): measured equal to the tiled form
        // (31.6 against 32.6 us at the GRU input projections' shape, the same bits; profiles/r05_gru_experiments.txt)
        if (K == 128 && mode != 2 && !act && !accum && !epi.stat_part && !epi.addg && !g_mfma_one && (g_gsb_dbg & 64) && !(g_gsb_dbg & 3) && /* bits 8-11: ablations */ (int)grid.x <= 128 &&
            !(ldc & 3) && !(reinterpret_cast<uintptr_t>(C0) & 15) && !(reinterpret_cast<uintptr_t>(C1) & 15)) {
            const int ngroups = (int)grid.x, wpg = 256 / ngroups;
            const int need = ((M + 31) / 32 + 7) / 8;      // workgroups per column group that still have a tile for every wave
            const int w = wpg < need ? wpg : need;
            const size_t smem = (size_t)4 * 3 * 4096 * sizeof(unsigned short);
            if (g_gsb_four_now) {
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sbp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                hipLaunchKernelGGL((gemm_sbp_kernel<true>), dim3(ngroups * w), dim3(512), smem, st, A0, lda, Bs0, Bs1, bias0, bias1, C0, C1, ldc, M, N, ngroups, g_gsb_dbg >> 8);
            } else {
                hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_sbp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                hipLaunchKernelGGL((gemm_sbp_kernel<false>), dim3(ngroups * w), dim3(512), smem, st, A0, lda, Bs0, Bs1, bias0, bias1, C0, C1, ldc, M, N, ngroups, g_gsb_dbg >> 8);
            }
            return 0;
        }
        if (K == 128 && mode != 2 && !(g_gsb_dbg & 32)) GSB_GO(4)
        else GSB_GO(0)
#undef GSB_GO
    } else {
        const int ng = K / GSB_KC * (mode == 2 ? 2 : 1);
#define G16_GO(NG_, CV_) { if (g_gsb_four_now) hipLaunchKernelGGL((gemm_sb16_kernel<NG_, CV_, false, true>), grid, dim3(1024), 0, st, A0, A1, lda, Bs0, Bs1, bias0, bias1, C0, C1, ldc, M, N, K, act, mode, accum, cvs, conv_H, conv_W, epi); \
                           else hipLaunchKernelGGL((gemm_sb16_kernel<NG_, CV_>), grid, dim3(1024), 0, st, A0, A1, lda, Bs0, Bs1, bias0, bias1, C0, C1, ldc, M, N, K, act, mode, accum, cvs, conv_H, conv_W, epi); }
#define G16_GO1(NG_) hipLaunchKernelGGL((gemm_sb16_kernel<NG_, false, true>), grid, dim3(1024), 0, st, A0, A1, lda, Bs0, Bs1, bias0, bias1, C0, C1, ldc, M, N, K, act, mode, accum, cvs, conv_H, conv_W, epi)
        // bf16 single-product mode: instantiated for the shapes of the headline model (the GRU input gradients, K = 128, and the generic loop);
        // the implicit-convolution and resnet50 shapes keep the exact products
        if (g_mfma_one && !cvs && !(g_gsb_dbg & 16) && (ng == 24 || ng == 4)) { if (ng == 24) G16_GO1(24); else G16_GO1(4); return 0; }
        if (g_mfma_one && !cvs && ng != 36 && ng != 16 && ng != 12 && ng != 8) { G16_GO1(0); return 0; }
        if (cvs) { if (ng == 36) G16_GO(36, true) else if (ng == 72) G16_GO(72, true) else G16_GO(0, true) }      // implicit 3x3: K = 9 x 128 / 9 x 256 unrolled
        else if (g_gsb_dbg & 16) G16_GO(0, false)
        else if (ng == 36) G16_GO(36, false)        // resnet50_block stage 2: the 3x3 product on im2col rows (K = 9 x 128)
        else if (ng == 24) G16_GO(24, false)        // the GRU input gradients: K = 2 x 384
        else if (ng == 16) G16_GO(16, false)        // ... its 1x1 products with K = 512
        else if (ng == 12) G16_GO(12, false)
        else if (ng == 8) G16_GO(8, false)
        else if (ng == 4) G16_GO(4, false)
        else G16_GO(0, false)
#undef G16_GO
#undef G16_GO1
    }
    return 0;
}
