// conv_pool_sb.hip — the z-free form of conv_pool.hip (first-layer Conv2D, layers.py:27-32, with the (5,4) max-pool
// window reduction of the pre-normalisation output in its epilogue) on the bf16 matrix cores: every fp32 operand is
// split exactly into three bf16 values and a product is the 6 leading partial products, accumulated in fp32 — the
// scheme of conv_sb.hip / gemm_sb.hip (fp32-level accuracy, 2.7x the rate of v_mfma_f32_32x32x2_f32).
//
// Why this layer maps well: the reduction index is k = tap * CP + channel with the channels of a pixel contiguous
// (CP = 8 slots for the 7 FOA channels, 16 for the 10 MIC ones), so the 8 consecutive k of a bf16 A fragment are ONE
// pixel's channel vector: a lane reads 16 B of the halo patch at (row + dy, bin + dx), no im2col and no transpose.
// Slot CIN of every pixel holds 1.0 and carries the bias through the centre tap's weight row; the slots above it and
// the k-steps past tap 8 meet zero weights.
//
// Tile, wave roles and the epilogue (window extreme zext, its position amax, BN statistics) are those of
// conv_first_fwd_pool_kernel<CIN, false, *>: tile = 10 image rows x 64 bins, wave = pooling row (w >> 1) x 32-bin strip
// (w & 1), 5 row tiles x 2 channel halves = 10 accumulator tiles that share the 6 weight fragments of a k-step.
// The patch (3 bf16 planes [12][66][CP]) is single-buffered: two workgroups share a CU and cover each other's staging.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CPSB_MAX_PERSISTENT 512
int conv_pool_sb_stat_capacity() { return CPSB_MAX_PERSISTENT; }

// exact 3-way truncation split of two floats, packed as bf16 pairs (element 0 in the low half) — as conv_sb.hip
__device__ __forceinline__ void cpsb_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
}

template <int CIN>
struct PoolSbGeom {
    static constexpr int CP = CIN < 8 ? 8 : 16;              // channel slots per pixel (CIN values, 1.0, zeros)
    // PACK (CIN = 7, round 4): K = 64 instead of 72 (+ 8 of padding = five k-steps): the ninth tap's seven channels ride in the spare slot 7 of
    // taps 0 .. 6 (a v_perm puts element t of the tap-8 pixel's vector into the fragment of tap t), the bias in slot 7 of tap 7 (which the
    // patch already holds as 1.0): four k-steps, a fifth of the MFMAs gone
    static constexpr bool PACK = CIN == 7;
    static constexpr int KTOT = PACK ? 64 : 9 * CP;
    static constexpr int NS = (KTOT + 15) / 16;              // k-steps of 16
    // weight row stride in bf16: >= NS*16 and an odd number of 16-B slots, so the 16 lanes of a ds_read_b128 group
    // (co = lane) land on 16 different slots of the 256-B bank row
    static constexpr int KW = (((NS * 16 / 8) | 1)) * 8;
    static constexpr int PLANE = 12 * 66 * CP;               // bf16 per patch plane
    static constexpr int WPLANE = 64 * KW;
    static constexpr size_t SMEM = (size_t)(3 * PLANE + 3 * WPLANE) * 2 + 512 * sizeof(float);
};

// ONE: bf16 single-product mode (common.h g_mfma_one): operands rounded to nearest bf16, plane 0 only, one MFMA per (k-step, row, half)
template <int CIN, bool WRITE_AMAX, bool ONE>
__global__ __launch_bounds__(256, (CIN < 8 ? 2 : 1)) void conv_first_fwd_pool_sb_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                        const float* __restrict__ bias,
                                                                        const float* __restrict__ gamma, float* __restrict__ zext,
                                                                        unsigned char* __restrict__ amax,
                                                                        float* __restrict__ stat_partial, int B, int H) {
    using G = PoolSbGeom<CIN>;
    constexpr int CP = G::CP, NS = G::NS, KW = G::KW, PLANE = G::PLANE, WPLANE = G::WPLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned short cpsb_smem[];
    unsigned short* patch = cpsb_smem;                 // [3][12][66][CP]
    unsigned short* Wl = patch + 3 * PLANE;            // [3][64][KW]
    float* red = reinterpret_cast<float*>(Wl + 3 * WPLANE);   // [4][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = lane >> 5, li = lane & 31;
    const int grp = wave >> 1, strip = wave & 1;
    // ---- weights: w [9*CIN][64] fp32 (k = tap*CIN + c) -> planes [co][tap*CP + c]; bias on (centre tap, slot CIN)
    for (int idx = tid; idx < 3 * WPLANE / 2; idx += 256) reinterpret_cast<unsigned*>(Wl)[idx] = 0u;
    for (int idx = tid; idx < 3 * PLANE / 2; idx += 256) reinterpret_cast<unsigned*>(patch)[idx] = 0u;   // halo columns stay zero
    __syncthreads();
    for (int idx = tid; idx < (9 * CIN + 1) * 64; idx += 256) {
        const int k = idx >> 6, co = idx & 63;
        const bool isb = k == 9 * CIN;
        // output channels with gamma < 0 are computed NEGATED (weights and bias): their window extreme is then a maximum like
        // everyone else's, with no per-value sign flip in the reduction; the stored extreme and the sum are flipped back
        const float v0_ = isb ? (bias ? bias[co] : 0.f) : w[idx];
        const float v = gamma[co] < 0.f ? -v0_ : v0_;
        const int kk = G::PACK ? (isb ? 63 : (k / CIN < 8 ? (k / CIN) * CP + (k % CIN) : (k % CIN) * CP + 7))
                               : (isb ? 4 * CP + CIN : (k / CIN) * CP + (k % CIN));
        const unsigned u = __float_as_uint(v);
        const float r = v - __uint_as_float(u & 0xffff0000u);
        const unsigned vv = __float_as_uint(r);
        const float s = r - __uint_as_float(vv & 0xffff0000u);
        unsigned short* o = Wl + co * KW + kk;
        if (ONE) { o[0] = (unsigned short)bf16_rne_bits(v); continue; }
        o[0] = (unsigned short)(u >> 16);
        o[WPLANE] = (unsigned short)(vv >> 16);
        o[2 * WPLANE] = (unsigned short)(__float_as_uint(s) >> 16);
    }
    // gamma < 0: the window extreme that survives BN+ReLU+MaxPool is the minimum of z -> flip the sign, take the maximum
    const unsigned smask[2] = {gamma[li] < 0.f ? 0x80000000u : 0u, gamma[32 + li] < 0.f ? 0x80000000u : 0u};
    const int tiles_per_img = (H + 9) / 10;
    const int ntiles = B * tiles_per_img;
    const int HP = H / 5;
    // ---- patch staging: 12 rows x 64 bins = 768 pixels, 3 per thread (pixel = idx: row idx >> 6, bin idx & 63)
    float stg[3][CIN];
#define CPSB_ISSUE(tile_)                                                                               \
    {                                                                                                   \
        const int ib_ = (tile_) / tiles_per_img, it0_ = ((tile_) - ib_ * tiles_per_img) * 10;           \
        _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                                 \
            const int idx = tid + 256 * u;                                                              \
            const int t = it0_ - 1 + (idx >> 6);                                                        \
            const bool ok = t >= 0 && t < H;                                                            \
            const float* p_ = ok ? x + ((size_t)(ib_ * H + t) * 64 + (idx & 63)) * CIN : x;             \
            const unsigned keep_ = ok ? 0xffffffffu : 0u;   /* AND mask: a select on a loaded value becomes a branch */ \
            _Pragma("unroll") for (int c = 0; c < CIN; ++c) stg[u][c] = __uint_as_float(__float_as_uint(p_[c]) & keep_); \
        }                                                                                               \
    }
#define CPSB_SLOT(u_, c_) ((c_) < CIN ? stg[u_][(c_) < CIN ? (c_) : 0] : ((c_) == CIN ? 1.f : 0.f))
#define CPSB_COMMIT()                                                                                   \
    _Pragma("unroll") for (int u = 0; u < 3; ++u) {                                                     \
        const int idx = tid + 256 * u;                                                                  \
        unsigned short* d_ = patch + (((idx >> 6) * 66 + (idx & 63) + 1) * CP);                         \
        _Pragma("unroll") for (int q = 0; q < CP / 8; ++q) {                                            \
            if (ONE) {                                                                                  \
                const u32x4 rv_ = {bf16_rne_bits(CPSB_SLOT(u, 8 * q + 0)) | (bf16_rne_bits(CPSB_SLOT(u, 8 * q + 1)) << 16), \
                                   bf16_rne_bits(CPSB_SLOT(u, 8 * q + 2)) | (bf16_rne_bits(CPSB_SLOT(u, 8 * q + 3)) << 16), \
                                   bf16_rne_bits(CPSB_SLOT(u, 8 * q + 4)) | (bf16_rne_bits(CPSB_SLOT(u, 8 * q + 5)) << 16), \
                                   bf16_rne_bits(CPSB_SLOT(u, 8 * q + 6)) | (bf16_rne_bits(CPSB_SLOT(u, 8 * q + 7)) << 16)}; \
                *reinterpret_cast<u32x4*>(d_ + 8 * q) = rv_;                                            \
                continue;                                                                               \
            }                                                                                           \
            unsigned h0_, h1_, h2_, h3_, m0_, m1_, m2_, m3_, l0_, l1_, l2_, l3_;                        \
            cpsb_split3_pair(CPSB_SLOT(u, 8 * q + 0), CPSB_SLOT(u, 8 * q + 1), h0_, m0_, l0_);          \
            cpsb_split3_pair(CPSB_SLOT(u, 8 * q + 2), CPSB_SLOT(u, 8 * q + 3), h1_, m1_, l1_);          \
            cpsb_split3_pair(CPSB_SLOT(u, 8 * q + 4), CPSB_SLOT(u, 8 * q + 5), h2_, m2_, l2_);          \
            cpsb_split3_pair(CPSB_SLOT(u, 8 * q + 6), CPSB_SLOT(u, 8 * q + 7), h3_, m3_, l3_);          \
            const u32x4 hv_ = {h0_, h1_, h2_, h3_}, mv_ = {m0_, m1_, m2_, m3_}, lv_ = {l0_, l1_, l2_, l3_}; \
            *reinterpret_cast<u32x4*>(d_ + 8 * q) = hv_;                                                \
            *reinterpret_cast<u32x4*>(d_ + PLANE + 8 * q) = mv_;                                        \
            *reinterpret_cast<u32x4*>(d_ + 2 * PLANE + 8 * q) = lv_;                                    \
        }                                                                                               \
    }
    int tile = blockIdx.x;
    if (tile < ntiles) {
        CPSB_ISSUE(tile)
        CPSB_COMMIT()
    }
    __syncthreads();
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    f32x2s s1p[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2p[2] = {{0.f, 0.f}, {0.f, 0.f}};
    // A fragment of k-step s: lane (li, kg) reads the 8 slots [c0, c0 + 8) of pixel (row + dy, bin + dx) with
    // k0 = 16 s + 8 kg, tap = min(k0 / CP, 8) (past tap 8 the weights are zero: any finite pixel will do), c0 = k0 % CP
#define CPSB_TAP(s_, g_) ((16 * (s_) + 8 * (g_)) / CP > 8 ? 8 : (16 * (s_) + 8 * (g_)) / CP)
#define CPSB_AOFF(s_, g_) (((CPSB_TAP(s_, g_) / 3) * 66 + (CPSB_TAP(s_, g_) % 3)) * CP + ((16 * (s_) + 8 * (g_)) / CP > 8 ? 0 : (16 * (s_) + 8 * (g_)) % CP))
    const unsigned short* pbase = patch + ((5 * grp) * 66 + 32 * strip + li) * CP;
    const unsigned short* wbase = Wl + li * KW + 8 * kg;
    const int lane_e = kg * 64 + li;
#define CPSB_MFMA(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC_, 0, 0, 0)
    // PACK: slot 7 of the fragment <- element (2 s + kg) of the tap-8 pixel (row + 2, bin + 2): dword s of its vector, half kg
#define CPSB_FRAG(ptr_, p8_)                                                                            \
    ([&]() {                                                                                            \
        u32x4 f_ = *reinterpret_cast<const u32x4*>(ptr_);                                               \
        if (G::PACK) f_.w = __builtin_amdgcn_perm(*reinterpret_cast<const unsigned*>(p8_), f_.w, psel); \
        return __builtin_bit_cast(bf16x8, f_);                                                          \
    }())
#define CPSB_ROW(j_, ACCA_, ACCB_)                                                                      \
    {                                                                                                   \
        const unsigned short* ap_ = pa + (j_) * 66 * CP;                                                \
        const unsigned short* p8_ = p8 + (j_) * 66 * CP;                                                \
        const bf16x8 ah = CPSB_FRAG(ap_, p8_);                                                          \
        CPSB_MFMA(ah, bh0, ACCA_); CPSB_MFMA(ah, bh1, ACCB_);                                           \
        if (!ONE) {                                                                                     \
        const bf16x8 am = CPSB_FRAG(ap_ + PLANE, p8_ + PLANE), al = CPSB_FRAG(ap_ + 2 * PLANE, p8_ + 2 * PLANE);   \
        CPSB_MFMA(ah, bm0, ACCA_); CPSB_MFMA(ah, bm1, ACCB_);                                           \
        CPSB_MFMA(am, bh0, ACCA_); CPSB_MFMA(am, bh1, ACCB_); CPSB_MFMA(ah, bl0, ACCA_); CPSB_MFMA(ah, bl1, ACCB_);   \
        CPSB_MFMA(al, bh0, ACCA_); CPSB_MFMA(al, bh1, ACCB_); CPSB_MFMA(am, bm0, ACCA_); CPSB_MFMA(am, bm1, ACCB_);   \
        }                                                                                               \
    }
    // window reduction of one accumulator register of one channel half: conv_pool.hip's CP_DRAIN without the z store
#define CPSB_DRAIN(r_, c_, Y0, Y1, Y2, Y3, Y4)                                                          \
    {                                                                                                   \
        const float w0 = Y0[r_], w1 = Y1[r_], w2 = Y2[r_], w3 = Y3[r_], w4 = Y4[r_];                    \
        const float hi5 = fmaxf(fmaxf(fmaxf(w0, w1), fmaxf(w2, w3)), w4);                               \
        const bool take_ = ((r_) & 3) == 0 || hi5 > best;     /* strict: the first extreme in scan order wins ties */ \
        if (WRITE_AMAX) {    /* training: remember WHERE the extreme is: position row * 4 + column of the window */ \
            int row_ = 4;                                                                               \
            row_ = (w3 == hi5) ? 3 : row_; row_ = (w2 == hi5) ? 2 : row_;                               \
            row_ = (w1 == hi5) ? 1 : row_; row_ = (w0 == hi5) ? 0 : row_;                               \
            bpos = take_ ? row_ * 4 + ((r_) & 3) : bpos;                                                \
        }                                                                                               \
        best = take_ ? hi5 : best;                                                                      \
        if (((r_) & 3) == 3) {                                                                          \
            (zext + (er + (size_t)((2 * ((r_) >> 2)) * 64 + 32 * (c_))))[lane_e] = __uint_as_float(__float_as_uint(best) ^ smask[c_]); \
            if (WRITE_AMAX) (amax + (er + (size_t)((2 * ((r_) >> 2)) * 64 + 32 * (c_))))[lane_e] = (unsigned char)bpos; \
        }                                                                                               \
    }
    // BN statistics of one accumulator tile: packed adds / FMAs over register pairs (any pairing sums to the same totals)
#define CPSB_STATS(c_, Y)                                                                               \
    _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                                 \
        const f32x2s y2 = {Y[r], Y[r + 1]};                                                             \
        s1p[c_] += y2;                                                                                  \
        s2p[c_] = __builtin_elementwise_fma(y2, y2, s2p[c_]);                                           \
    }
#ifdef CPSB_TIMING
    long long tm_top = 0, tm_mfma = 0, tm_drain = 0, tm_commit = 0;
#endif
    for (; tile < ntiles; tile += gridDim.x) {
#ifdef CPSB_TIMING
        const long long c0 = clock64();
#endif
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * 10;
        const int nxt = tile + gridDim.x;
        CPSB_ISSUE(nxt < ntiles ? nxt : tile)
        __builtin_amdgcn_sched_barrier(0);
        const int tg = t0 + 5 * grp;             // first image row of this wave's pooling row
        const bool live = tg < H;                // H % 5 == 0: a pooling row is entirely inside or outside
        const size_t er = ((size_t)(b * HP + tg / 5) * 16 + 8 * strip) * 64;
        float best = 0.f;
        int bpos = 0;
#ifdef CPSB_TIMING
        const long long c1 = clock64();
#endif
        f32x16 accA0 = zero16(), accA1 = zero16(), accA2 = zero16(), accA3 = zero16(), accA4 = zero16();
        f32x16 accB0 = zero16(), accB1 = zero16(), accB2 = zero16(), accB3 = zero16(), accB4 = zero16();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned short* pa = pbase + (kg ? CPSB_AOFF(s, 1) : CPSB_AOFF(s, 0));
            // elements 2 s, 2 s + 1 of the tap-8 pixel.  The offset is laundered through an empty asm: seen as consecutive, the four k-steps' dwords
            // were merged into one ds_read_b128 per (row, plane) that stayed live for the whole tile — 60 registers, 84 spill instructions, 462 us
            int p8o = 2 * s;
            asm volatile("" : "+s"(p8o));
            const unsigned short* p8 = pbase + (2 * 66 + 2) * CP + p8o;
            const unsigned psel = s == 3 ? (kg ? 0x03020100u : 0x05040100u) : (kg ? 0x07060100u : 0x05040100u);   // tap 7 keeps its 1.0 (the bias)
            const unsigned short* wp = wbase + 16 * s;
            const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(wp), bh1 = *reinterpret_cast<const bf16x8*>(wp + 32 * KW);
            bf16x8 bm0 = bh0, bm1 = bh1, bl0 = bh0, bl1 = bh1;       // (single-product mode: unused)
            if (!ONE) {
                bm0 = *reinterpret_cast<const bf16x8*>(wp + WPLANE); bm1 = *reinterpret_cast<const bf16x8*>(wp + WPLANE + 32 * KW);
                bl0 = *reinterpret_cast<const bf16x8*>(wp + 2 * WPLANE); bl1 = *reinterpret_cast<const bf16x8*>(wp + 2 * WPLANE + 32 * KW);
            }
            CPSB_ROW(0, accA0, accB0)
            CPSB_ROW(1, accA1, accB1)
            CPSB_ROW(2, accA2, accB2)
            CPSB_ROW(3, accA3, accB3)
            CPSB_ROW(4, accA4, accB4)
        }
#ifdef CPSB_TIMING
        const long long c2 = clock64();
#endif
        if (live) {
            CPSB_STATS(0, accA0) CPSB_STATS(0, accA1) CPSB_STATS(0, accA2) CPSB_STATS(0, accA3) CPSB_STATS(0, accA4)
            CPSB_STATS(1, accB0) CPSB_STATS(1, accB1) CPSB_STATS(1, accB2) CPSB_STATS(1, accB3) CPSB_STATS(1, accB4)
#pragma unroll
            for (int r = 0; r < 16; ++r) CPSB_DRAIN(r, 0, accA0, accA1, accA2, accA3, accA4)
#pragma unroll
            for (int r = 0; r < 16; ++r) CPSB_DRAIN(r, 1, accB0, accB1, accB2, accB3, accB4)
        }
#ifdef CPSB_TIMING
        const long long c3 = clock64();
#endif
        // single patch buffer: everyone is done reading it, then the prefetched tile replaces it
        lds_barrier();
        CPSB_COMMIT()
        lds_barrier();
#ifdef CPSB_TIMING
        const long long c4 = clock64();
        tm_top += c1 - c0; tm_mfma += c2 - c1; tm_drain += c3 - c2; tm_commit += c4 - c3;
#endif
    }
#undef CPSB_DRAIN
#undef CPSB_STATS
#undef CPSB_ROW
#undef CPSB_FRAG
#undef CPSB_MFMA
#undef CPSB_AOFF
#undef CPSB_TAP
#undef CPSB_ISSUE
#undef CPSB_COMMIT
#undef CPSB_SLOT
    if (stat_partial) {
        float s1[2], s2[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            s1[c] = __uint_as_float(__float_as_uint(s1p[c].x + s1p[c].y) ^ smask[c]);     // negated channels: flip the sum back
            s2[c] = s2p[c].x + s2p[c].y;
            s1[c] += __shfl_xor(s1[c], 32);
            s2[c] += __shfl_xor(s2[c], 32);
        }
        if (kg == 0) {
            red[wave * 128 + li] = s1[0];
            red[wave * 128 + 32 + li] = s1[1];
            red[wave * 128 + 64 + li] = s2[0];
            red[wave * 128 + 96 + li] = s2[1];
        }
        __syncthreads();
        if (tid < 128) stat_partial[(size_t)blockIdx.x * 128 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
#ifdef CPSB_TIMING
        __syncthreads();
        if (tid == 0) {      // diagnostic build (tools/tune_conv1.py): the first statistics slots carry wave 0's phase cycles instead
            float* o = stat_partial + (size_t)blockIdx.x * 128;
            o[0] = (float)tm_top; o[1] = (float)tm_mfma; o[2] = (float)tm_drain; o[3] = (float)tm_commit;
        }
#endif
    }
}

template <int CIN>
static int launch_cpsb(hipStream_t st, const float* x, const float* w, const float* bias, const float* gamma, float* zext,
                       unsigned char* amax, float* stat_partial, int* n_partial, int B, int H) {
    using G = PoolSbGeom<CIN>;
    const int ntiles = B * ((H + 9) / 10);
    const int grid = ntiles < CPSB_MAX_PERSISTENT ? ntiles : CPSB_MAX_PERSISTENT;
#define CPSB_GO(A_)                                                                                                  \
    {                                                                                                                \
        if (g_mfma_one) {                                                                                            \
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_fwd_pool_sb_kernel<CIN, A_, true>),             \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM);                               \
        hipLaunchKernelGGL((conv_first_fwd_pool_sb_kernel<CIN, A_, true>), dim3(grid), dim3(256), G::SMEM, st, x, w, bias, \
                           gamma, zext, amax, stat_partial, B, H);                                                   \
        } else {                                                                                                     \
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_fwd_pool_sb_kernel<CIN, A_, false>),            \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::SMEM);                               \
        hipLaunchKernelGGL((conv_first_fwd_pool_sb_kernel<CIN, A_, false>), dim3(grid), dim3(256), G::SMEM, st, x, w, bias, \
                           gamma, zext, amax, stat_partial, B, H);                                                   \
        }                                                                                                            \
    }
    if (amax) CPSB_GO(true)
    else CPSB_GO(false)
#undef CPSB_GO
    if (n_partial) *n_partial = grid;
    return 0;
}

// same contract as launch_conv_first_fwd_pool with z == nullptr (conv_pool.hip)
int launch_conv_first_fwd_pool_sb(hipStream_t st, const float* x, const float* w, const float* bias, const float* gamma,
                                  float* zext, unsigned char* amax, float* stat_partial, int* n_partial, int B, int H, int Cin) {
    if (H % 5 || H <= 0 || B <= 0) return -2;
    if (Cin == 7) return launch_cpsb<7>(st, x, w, bias, gamma, zext, amax, stat_partial, n_partial, B, H);
    if (Cin == 10) return launch_cpsb<10>(st, x, w, bias, gamma, zext, amax, stat_partial, n_partial, B, H);
    return -2;
}
