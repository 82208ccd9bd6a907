// conv_sb.hip — Cin = Cout = 64 3x3 'same' convolution (layers.py:27-32) on the bf16 matrix cores at fp32
// accuracy: "split-bf16" operands.
//
// Every fp32 operand is split EXACTLY into three bf16 values by truncation, x = hi + mid + lo (8 + 8 + 8
// significant bits), and a product a*b is accumulated in fp32 from the six partial products
//   hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid
// on v_mfma_f32_32x32x16_bf16.  The three dropped terms (mid*lo, lo*mid, lo*lo) are <= 2^-24 relative, the
// same size as one fp32 rounding, so the result meets the fp32 parity bar, while the MFMA work per MAC is
// 6/16 of v_mfma_f32_32x32x2_f32 (the f32-input MFMA runs at 1/16 of the bf16 rate).
//
// Tile: 256 pixels x 64 output channels per 512-thread block (8 waves, wave w = pixel rows 32w..32w+31 x both
// co-tiles).  Per tap: the fp32 A tile [256 px][64 ci] is prefetched into registers during the previous tap's
// MFMAs, split at commit time into three bf16 planes in LDS (row stride 144 B: conflict-free 16-B fragment
// reads); the tap's weights come pre-split and pre-transposed ([co][ci]) from split_weights_kernel.
#include "common.h"
#include "prep.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SB_LD 72                    // bf16 elements per LDS row: 64 + 8 pad (144 B)
#define SB_A_ELEMS (3 * 256 * SB_LD)
#define SB_W_ELEMS (3 * 64 * SB_LD)
#define SB_MAX_PERSISTENT 512

int conv_sb_partial_capacity() { return SB_MAX_PERSISTENT; }

// exact 3-way truncation split of two floats, packed as bf16 pairs (element 0 in the low half)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
}

// bf16 single-product mode: the two floats rounded to nearest-even bf16, packed like the planes above
__device__ __forceinline__ unsigned rne_pair(float x0, float x1) { return bf16_rne_bits(x0) | (bf16_rne_bits(x1) << 16); }

int g_mfma_one = 0;

// w [9][in 64][out 64] fp32 -> planes [9][3][out 64][in 64] bf16 (transposed so a B fragment is 8 contiguous k)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ wsp, int one) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 9 * 4096) return;
    const int tap = idx >> 12, rem = idx & 4095, out = rem >> 6, in = rem & 63;
    const float x = w[tap * 4096 + in * 64 + out];
    const unsigned u = one ? bf16_rne_bits(x) << 16 : __float_as_uint(x) & 0xffff0000u;      // all three planes in both modes: prep.h
    const float r = x - __uint_as_float(u);
    const unsigned v = __float_as_uint(r);
    const float s = r - __uint_as_float(v & 0xffff0000u);
    unsigned short* o = wsp + (size_t)tap * 3 * 4096 + out * 64 + in;
    o[0] = (unsigned short)(u >> 16);
    o[4096] = (unsigned short)(v >> 16);
    o[2 * 4096] = (unsigned short)(__float_as_uint(s) >> 16);
}

// several weight tensors in one launch; flip = the input-gradient form: tap 8 - tap with in / out swapped
// (wt[tap][co][ci] = w[8 - tap][ci][co], conv.hip flip_weights_kernel) split directly from the unflipped tensor
__global__ __launch_bounds__(256) void split_weights_batch_kernel(SplitWeightJobs jobs) { split_weights_body(jobs, blockIdx.y, blockIdx.x); }

int launch_split_weights_batch(hipStream_t st, int n, const float* const* w, unsigned short* const* dst, const int* flip) {
    if (n <= 0 || n > 8) return -1;
    SplitWeightJobs j;
    j.one = g_mfma_one;
    for (int i = 0; i < n; ++i) { j.w[i] = w[i]; j.dst[i] = dst[i]; j.flip[i] = flip[i]; }
    hipLaunchKernelGGL(split_weights_batch_kernel, dim3(9 * 4096 / 256, n), dim3(256), 0, st, j);
    return 0;
}

int launch_split_weights(hipStream_t st, const float* w, unsigned short* wsp) {
    hipLaunchKernelGGL(split_weights_kernel, dim3(9 * 4096 / 256), dim3(256), 0, st, w, wsp, g_mfma_one);
    return 0;
}

template <bool STATS>
__global__ __launch_bounds__(512) void conv64_fwd_sb_kernel(const float* __restrict__ x, const unsigned short* __restrict__ wsp,
                                                            const float* __restrict__ bias, float* __restrict__ z,
                                                            float* __restrict__ stat_partial, int npix, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned short sb_smem[];
    unsigned short* Ap = sb_smem;                  // [3][256][SB_LD]
    unsigned short* Wp = sb_smem + SB_A_ELEMS;     // [3][64][SB_LD]
    float* red = reinterpret_cast<float*>(sb_smem + SB_A_ELEMS + SB_W_ELEMS);   // [8][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    const int ntiles = (npix + 255) >> 8;
    const int g4 = (tid & 15) * 4;   // channel group of this thread's staging loads
    const int pxs = tid >> 4;        // staging pixel slot (0..31), pixels pxs + 32u
    const int wrow = tid >> 3, wchunk = (tid & 7) * 8;   // weight staging: row co, 8 bf16 per uint4
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    float4 ar[8];
    uint4 wr[3];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int p0 = tile << 8;
        unsigned vmask[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + pxs + 32 * u;
            unsigned m = 0;
            if (p < npix) {
                const int f = p % W, t = (p / W) % H;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dt = tap / 3 - 1, df = tap % 3 - 1;
                    if (t + dt >= 0 && t + dt < H && f + df >= 0 && f + df < W) m |= 1u << tap;
                }
            }
            vmask[u] = m;
        }
#define SB_ISSUE(tap_)                                                                                          \
    {                                                                                                           \
        const int tp_ = (tap_);                                                                                 \
        const int shift_ = (tp_ / 3 - 1) * W + (tp_ % 3 - 1);                                                   \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                         \
            const bool ok_ = (vmask[u] >> tp_) & 1u;                                                            \
            const float4 v_ = *reinterpret_cast<const float4*>(x + (size_t)(ok_ ? p0 + pxs + 32 * u + shift_ : 0) * 64 + g4); \
            ar[u] = ok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                                 \
        }                                                                                                       \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                        \
            wr[pl] = reinterpret_cast<const uint4*>(wsp + ((size_t)tp_ * 3 + pl) * 4096)[tid];                  \
    }
#define SB_COMMIT()                                                                                             \
    {                                                                                                           \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                         \
            unsigned h0, m0, l0, h1, m1, l1;                                                                    \
            split3_pair(ar[u].x, ar[u].y, h0, m0, l0);                                                          \
            split3_pair(ar[u].z, ar[u].w, h1, m1, l1);                                                          \
            unsigned short* d_ = Ap + (pxs + 32 * u) * SB_LD + g4;                                              \
            *reinterpret_cast<uint2*>(d_) = make_uint2(h0, h1);                                                 \
            *reinterpret_cast<uint2*>(d_ + 256 * SB_LD) = make_uint2(m0, m1);                                   \
            *reinterpret_cast<uint2*>(d_ + 2 * 256 * SB_LD) = make_uint2(l0, l1);                               \
        }                                                                                                       \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                        \
            *reinterpret_cast<uint4*>(Wp + pl * 64 * SB_LD + wrow * SB_LD + wchunk) = wr[pl];                   \
    }
        SB_ISSUE(0)
        lds_barrier();   // previous tile's readers are done
        SB_COMMIT()
        lds_barrier();
        f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            SB_ISSUE(tap < 8 ? tap + 1 : 8)   // unconditional (tap 8 re-reads itself): no phi on the staged registers
            __builtin_amdgcn_sched_barrier(0);
            const unsigned short* arow = Ap + (wave * 32 + li) * SB_LD + 8 * hi;
            const unsigned short* wrow0 = Wp + li * SB_LD + 8 * hi;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 a[3], b[3][2];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[pl] = *reinterpret_cast<const bf16x8*>(arow + pl * 256 * SB_LD + 16 * s);
                    b[pl][0] = *reinterpret_cast<const bf16x8*>(wrow0 + pl * 64 * SB_LD + 16 * s);
                    b[pl][1] = *reinterpret_cast<const bf16x8*>(wrow0 + pl * 64 * SB_LD + 32 * SB_LD + 16 * s);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0][c], acc[c], 0, 0, 0);   // hi*hi
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1][c], acc[c], 0, 0, 0);   // hi*mid
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0][c], acc[c], 0, 0, 0);   // mid*hi
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2][c], acc[c], 0, 0, 0);   // hi*lo
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0][c], acc[c], 0, 0, 0);   // lo*hi
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1][c], acc[c], 0, 0, 0);   // mid*mid
                }
            }
            lds_barrier();      // unconditional commit (see conv64_fwd_sbr_kernel): tap 8 re-commits itself
            SB_COMMIT()
            lds_barrier();
        }
#undef SB_ISSUE
#undef SB_COMMIT
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float bv = bias ? bias[c * 32 + li] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wave * 32 + mfma_row(r, hi);
                if (p < npix) {
                    const float v = acc[c][r] + bv;
                    z[(size_t)p * 64 + c * 32 + li] = v;
                    s1[c] += v;
                    s2[c] = fmaf(v, v, s2[c]);
                }
            }
        }
    }
    if (STATS) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            s1[c] += __shfl_xor(s1[c], 32);
            s2[c] += __shfl_xor(s2[c], 32);
        }
        __syncthreads();
        if (hi == 0) {
            red[wave * 128 + li] = s1[0];
            red[wave * 128 + 32 + li] = s1[1];
            red[wave * 128 + 64 + li] = s2[0];
            red[wave * 128 + 96 + li] = s2[1];
        }
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) t += red[w8 * 128 + tid];
            stat_partial[(size_t)blockIdx.x * 128 + tid] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Halo-region variant: a tile is R full image rows x W columns (TP = R*W pixels, one wave per 32 pixels).
// The (R+2) x (W+2) input region is loaded ONCE per tile (prefetched into registers under the previous
// tile's MFMAs), split into the three bf16 planes once, and all 9 taps read their A fragments from it at
// shifted pixel rows; only the 27 KB of pre-split weights change per tap.
template <int WLOG2, int R, bool SWZ, bool STATS>
__global__ __launch_bounds__(32 * ((R << WLOG2) / 32) * 2) void conv64_fwd_sbr_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wsp, const float* __restrict__ bias,
    float* __restrict__ z, float* __restrict__ stat_partial, int B, int H) {
    constexpr int W = 1 << WLOG2, TP = R * W, NW = TP / 32, NT = 64 * NW;
    constexpr int RW = W + 2, RR = R + 2, NPIX = RR * RW;
    constexpr int NREG4 = NPIX * 16, NPF = (NREG4 + NT - 1) / NT;      // region float4 slots, per thread
    constexpr int NW4 = 3 * 64 * 8, NWF = (NW4 + NT - 1) / NT;         // weight uint4 slots per tap, per thread
    static_assert(TP % 32 == 0, "tile must be whole 32-pixel MFMA row tiles");
    // LDS rows of 64 bf16 (128 B).  Padded layout: row stride 72 (SB_LD).  Swizzled layout (SWZ): stride 64 and
    // the 16-byte chunk c of row p stored at chunk c ^ ((p >> 1) & 7): 16 consecutive rows x one chunk index
    // still cover all 64 banks, with no padding — which is what lets a 256-pixel tile (8 waves, two per SIMD)
    // fit the 160 KB of LDS next to a tap's weights.
    constexpr int LD = SWZ ? 64 : SB_LD;
    extern __shared__ __attribute__((aligned(16))) unsigned short sb_smem[];
    unsigned short* Rp = sb_smem;                          // [3][NPIX][LD]
    unsigned short* Wp = sb_smem + 3 * NPIX * LD;          // [3][64][LD]
    float* red = reinterpret_cast<float*>(Wp + 3 * 64 * LD);  // [NW][128]
#define SBR_CH(p_, c_) (SWZ ? ((c_) ^ (((p_) >> 1) & 7)) : (c_))      /* where chunk c_ of row p_ lives */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    const int tiles_per_img = (H + R - 1) / R;
    const int ntiles = B * tiles_per_img;
    // static part of this thread's region slots
    int roff[NPF];        // float offset relative to the tile's first pixel (may be negative: halo)
    int rrow[NPF];  // region row (image row = t0 - 1 + rrow), -1 = slot unused / outside the image columns
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = tid + NT * u;
        const int pix = idx >> 4, g = idx & 15;
        const int rr = pix / RW, cc = pix - rr * RW;
        const bool ok = idx < NREG4 && cc >= 1 && cc <= W;
        rrow[u] = ok ? rr : -1;
        roff[u] = ((rr - 1) * W + (cc - 1)) * 64 + g * 4;
    }
    float4 rreg[NPF];
    static_assert(NWF == 4 || NWF == 3, "weight staging is three or four named registers");
    u32x4 wreg0, wreg1, wreg2, wreg3;      // named (an indexed array of these was demoted to scratch)
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#define SBR_ISSUE_REGION(tile_)                                                                         \
    {                                                                                                   \
        const int tb_ = (tile_) / tiles_per_img, tt0_ = ((tile_) - tb_ * tiles_per_img) * R;            \
        const float* org_ = x + (size_t)(tb_ * H + tt0_) * W * 64;                                      \
        _Pragma("unroll") for (int u = 0; u < NPF; ++u) {                                               \
            const int t_ = tt0_ - 1 + rrow[u];                                                          \
            const bool ok_ = rrow[u] >= 0 && t_ >= 0 && t_ < H;                                         \
            const float4 v_ = *reinterpret_cast<const float4*>(ok_ ? org_ + roff[u] : x);               \
            rreg[u] = ok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                       \
        }                                                                                               \
    }
#define SBR_COMMIT_REGION()                                                                             \
    _Pragma("unroll") for (int u = 0; u < NPF; ++u) {                                                   \
        const int idx = tid + NT * u;                                                                   \
        if (idx < NREG4) {                                                                              \
            unsigned h0, m0, l0, h1, m1, l1;                                                            \
            split3_pair(rreg[u].x, rreg[u].y, h0, m0, l0);                                              \
            split3_pair(rreg[u].z, rreg[u].w, h1, m1, l1);                                              \
            unsigned short* d_ = Rp + (idx >> 4) * LD + SBR_CH(idx >> 4, (idx & 15) >> 1) * 8 + (idx & 1) * 4; \
            *reinterpret_cast<uint2*>(d_) = make_uint2(h0, h1);                                         \
            *reinterpret_cast<uint2*>(d_ + NPIX * LD) = make_uint2(m0, m1);                             \
            *reinterpret_cast<uint2*>(d_ + 2 * NPIX * LD) = make_uint2(l0, l1);                         \
        }                                                                                               \
    }
#define SBR_W_SRC(tap_, u_) \
    reinterpret_cast<const u32x4*>(wsp + (size_t)(tap_) * 3 * 4096)[(tid + NT * (u_)) < NW4 ? (tid + NT * (u_)) : 0]
#define SBR_ISSUE_W(tap_)                                                                               \
    {                                                                                                   \
        wreg0 = SBR_W_SRC(tap_, 0);                                                                     \
        wreg1 = SBR_W_SRC(tap_, 1);                                                                     \
        wreg2 = SBR_W_SRC(tap_, 2);                                                                     \
        if (NWF > 3) wreg3 = SBR_W_SRC(tap_, 3);                                                        \
    }
#define SBR_W_DST(u_, v_)                                                                               \
    {                                                                                                   \
        const int idx = tid + NT * (u_);                                                                \
        if (idx < NW4) {                                                                                \
            const int pl = idx >> 9, rem = idx & 511;      /* 512 uint4 per plane: row = rem>>3 */      \
            *reinterpret_cast<u32x4*>(Wp + pl * 64 * LD + (rem >> 3) * LD + SBR_CH(rem >> 3, rem & 7) * 8) = v_; \
        }                                                                                               \
    }
#define SBR_COMMIT_W()                                                                                  \
    {                                                                                                   \
        SBR_W_DST(0, wreg0)                                                                             \
        SBR_W_DST(1, wreg1)                                                                             \
        SBR_W_DST(2, wreg2)                                                                             \
        if (NWF > 3) SBR_W_DST(3, wreg3)                                                                \
    }
    int tile = blockIdx.x;
    // region row of this lane's pixel for tap (0,0): pixel p = 32*wave + li of the tile -> (r, c)
    const int pl_ = wave * 32 + li;
    const int pbase = (pl_ >> WLOG2) * RW + (pl_ & (W - 1));       // region row of this lane's pixel at tap (0,0)
    if (tile < ntiles) {
        SBR_ISSUE_REGION(tile)
        SBR_ISSUE_W(0)
        SBR_COMMIT_REGION()
        SBR_COMMIT_W()
    }
    lds_barrier();
#ifdef SBR_TIMING
    long long tm_mfma = 0, tm_b1 = 0, tm_cw = 0, tm_b2 = 0, tm_epi = 0, tm_top = 0, ttop = clock64();
    int tm_n = 0;
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * R;
        const int nxt = tile + gridDim.x;
        SBR_ISSUE_REGION(nxt < ntiles ? nxt : tile)     // unconditional: no phi on the staged registers
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[2] = {zero16(), zero16()}, accs[2] = {zero16(), zero16()};
        const unsigned short* wrow0 = Wp + li * LD;
        const int wsw = (li >> 1) & 7;        // swizzle key of weight rows li and li + 32 (the same)
#ifdef SBR_TIMING
        const long long tc0 = clock64();
#endif
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
#ifdef SBR_TIMING
            const long long ta = clock64();
#endif
            SBR_ISSUE_W(tap < 8 ? tap + 1 : 0)          // tap 8 prefetches tap 0 of the next tile
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch loads ahead of the MFMA steps
            const int prow = pbase + (tap / 3) * RW + (tap % 3);
            const unsigned short* arow = Rp + prow * LD;
            const int asw = (prow >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 a[3], bb[3][2];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const int ca = SWZ ? ((2 * s + hi) ^ asw) : (2 * s + hi), cw = SWZ ? ((2 * s + hi) ^ wsw) : (2 * s + hi);
                    a[pl] = *reinterpret_cast<const bf16x8*>(arow + pl * NPIX * LD + 8 * ca);
                    bb[pl][0] = *reinterpret_cast<const bf16x8*>(wrow0 + pl * 64 * LD + 8 * cw);
                    bb[pl][1] = *reinterpret_cast<const bf16x8*>(wrow0 + pl * 64 * LD + 32 * LD + 8 * cw);
                }
                // Four accumulator chains (channel half x {large, small} products), visited round-robin: a dependent
                // v_mfma_f32_32x32x16_bf16 cannot issue before its predecessor has left the pipe, and runs of six
                // MFMAs into one accumulator (the natural order) held the matrix pipe at about half rate.
#define SBR_MFMA(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC_, 0, 0, 0)
                SBR_MFMA(a[0], bb[0][0], acc[0]);  SBR_MFMA(a[0], bb[0][1], acc[1]);     // hi*hi
                SBR_MFMA(a[0], bb[1][0], accs[0]); SBR_MFMA(a[0], bb[1][1], accs[1]);    // hi*mid
                SBR_MFMA(a[1], bb[0][0], acc[0]);  SBR_MFMA(a[1], bb[0][1], acc[1]);     // mid*hi
                SBR_MFMA(a[0], bb[2][0], accs[0]); SBR_MFMA(a[0], bb[2][1], accs[1]);    // hi*lo
                SBR_MFMA(a[2], bb[0][0], acc[0]);  SBR_MFMA(a[2], bb[0][1], acc[1]);     // lo*hi
                SBR_MFMA(a[1], bb[1][0], accs[0]); SBR_MFMA(a[1], bb[1][1], accs[1]);    // mid*mid
#undef SBR_MFMA
            }
            // The commit is UNCONDITIONAL (after tap 8 it stores tap 0 of the next tile): when the staged
            // registers were only used inside an `if (tap < 8)` block the compiler sank the global loads
            // into that block, i.e. behind the barrier, and every tap paid a full L2 round trip.
#ifdef SBR_TIMING
            const long long tb = clock64();
#endif
            lds_barrier();
#ifdef SBR_TIMING
            const long long tcc = clock64();
#endif
            SBR_COMMIT_W()
#ifdef SBR_TIMING
            const long long td = clock64();
#endif
            lds_barrier();
#ifdef SBR_TIMING
            const long long te = clock64();
            tm_mfma += tb - ta; tm_b1 += tcc - tb; tm_cw += td - tcc; tm_b2 += te - td;
#endif
        }
#ifdef SBR_TIMING
        const long long tf = clock64();
#endif
        SBR_COMMIT_REGION()     // next tile's region (every wave passed the barrier after tap 8's reads)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float bv = bias ? bias[c * 32 + li] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = wave * 32 + mfma_row(r, hi);          // pixel of the tile
                const int t = t0 + (p >> WLOG2);
                if (t < H) {
                    const float v = (acc[c][r] + accs[c][r]) + bv;
                    z[((size_t)(b * H + t) * W + (p & (W - 1))) * 64 + c * 32 + li] = v;
                    s1[c] += v;
                    s2[c] = fmaf(v, v, s2[c]);
                }
            }
        }
        lds_barrier();          // the committed region is visible to every wave
#ifdef SBR_TIMING
        tm_epi += clock64() - tf; tm_top += tc0 - ttop; ttop = clock64(); ++tm_n;
#endif
    }
#ifdef SBR_TIMING
    if (tid == 0 && (blockIdx.x == 3 || blockIdx.x == 200))
        printf("sbr W=%d block %d tiles %d: per tile top %lld  mfma+issue %lld  barrier1 %lld  commitW %lld  barrier2 %lld  epilogue+commitR %lld\n",
               W, blockIdx.x, tm_n, tm_top / tm_n, tm_mfma / tm_n, tm_b1 / tm_n, tm_cw / tm_n, tm_b2 / tm_n, tm_epi / tm_n);
#endif
#undef SBR_ISSUE_REGION
#undef SBR_COMMIT_REGION
#undef SBR_ISSUE_W
#undef SBR_COMMIT_W
#undef SBR_CH
#undef SBR_W_SRC
#undef SBR_W_DST
    if (STATS) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            s1[c] += __shfl_xor(s1[c], 32);
            s2[c] += __shfl_xor(s2[c], 32);
        }
        __syncthreads();
        if (hi == 0) {
            red[wave * 128 + li] = s1[0];
            red[wave * 128 + 32 + li] = s1[1];
            red[wave * 128 + 64 + li] = s2[0];
            red[wave * 128 + 96 + li] = s2[1];
        }
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) t += red[w8 * 128 + tid];
            stat_partial[(size_t)blockIdx.x * 128 + tid] = t;
        }
    }
}

template <int WLOG2, int R, bool SWZ>
static int launch_sbr(hipStream_t st, const float* x, const unsigned short* wsp, const float* bias, float* z,
                      float* stat_partial, int* n_partial, int B, int H) {
    constexpr int W = 1 << WLOG2, NW = (R * W) / 32, NT = 64 * NW, NPIX = (R + 2) * (W + 2), LD = SWZ ? 64 : SB_LD;
    const int ntiles = B * ((H + R - 1) / R);
    const int grid = ntiles < 256 ? ntiles : 256;      // one block per CU (LDS-limited), persistent
    const size_t smem = (size_t)(3 * NPIX * LD + 3 * 64 * LD) * sizeof(unsigned short) + (size_t)NW * 128 * sizeof(float);
    if (stat_partial) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_fwd_sbr_kernel<WLOG2, R, SWZ, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((conv64_fwd_sbr_kernel<WLOG2, R, SWZ, true>), dim3(grid), dim3(NT), smem, st, x, wsp, bias, z, stat_partial, B, H);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_fwd_sbr_kernel<WLOG2, R, SWZ, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((conv64_fwd_sbr_kernel<WLOG2, R, SWZ, false>), dim3(grid), dim3(NT), smem, st, x, wsp, bias, z, stat_partial, B, H);
    }
    if (n_partial) *n_partial = grid;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Double-buffered-weights variant (the default for W = 16 and W = 4).  Two changes against conv64_fwd_sbr_kernel, both aimed at
// the 18 barriers + 9 serialised weight commits per tile that held its matrix pipe at ~54 % busy:
//  * the image is exactly W columns wide, so the region's two halo COLUMNS are always zero padding: they are not stored — a lane
//    whose shifted column falls outside reads ONE shared all-zero row instead — and the LDS they occupied pays for
//  * a SECOND weight buffer: tap t + 1's pre-split weights are committed while tap t's MFMAs run (they were loaded to registers a
//    tap earlier), so a tap needs one barrier instead of commit-between-two-barriers.
// Region = (R + 2) rows x W columns, three bf16 planes, rows XOR-swizzled; weights [2][3][64][64].
// ONE: bf16 single-product mode — operands rounded to nearest bf16, plane 0 only (region, weights, fragments), one MFMA per k-step and
// column tile instead of six.
// FOUR (round 4, backward products only: option "bwd_four_products"): four of the six products — hi*hi, hi*mid, mid*hi, mid*mid; the two with a lo
// factor (2^-17 relative each) are dropped, the lo planes are neither staged nor read.  An input gradient feeds no MaxPool / ReLU decision, so the
// error (~1.5e-5 relative per product, random in sign over a 576-term sum) stays one decade inside the 1e-4 bar; forward products keep all six.
// PRE (round 4): x is the PREVIOUS block's window-extreme tensor (conv_pool_sb.hip's zext) and the block's BatchNormalization + ReLU
// are applied while the region is loaded — max(0, fmaf(v, scale[c], shift[c])), the arithmetic of bn_relu_ext, so the result is the same bits —
// and every tile writes the activated values of the rows it owns to pre_out (the pooled tensor the backward pass reads): the separate
// elementwise pass over the pooled tensor (24 us per step) is gone.  pre_out must not alias x (a neighbour's halo read would see activated values).
// EXT (round 4, W = 16 with a (1,4) pool behind the block): the epilogue also stores, per pooling window of four bins and channel, the
// EXTREME of z that survives BatchNorm + ReLU + MaxPool — the maximum where gamma >= 0, the minimum where gamma < 0: BN is monotone in z
// (fmaf is), so pool(relu(bn(z))) = relu(bn(extreme)) bit for bit (conv_pool.hip's argument) — into ext_out [B][H][W/4][64]: the NEXT block's
// PRE loader turns them into this block's pooled tensor, and the separate pooling pass over z (16 us per step) is gone.  In the accumulator
// layout a lane's registers 4q .. 4q + 3 are exactly one window of one channel.
template <int WLOG2, int R, bool STATS, bool ONE, bool FOUR = false, bool PRE = false, bool EXT = false>
__global__ __launch_bounds__(32 * ((R << WLOG2) / 32) * 2) void conv64_fwd_sbd_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wsp, const float* __restrict__ bias,
    float* __restrict__ z, float* __restrict__ stat_partial, int B, int H,
    const float* __restrict__ pre_scale = nullptr, const float* __restrict__ pre_shift = nullptr, float* __restrict__ pre_out = nullptr,
    const float* __restrict__ ext_gamma = nullptr, float* __restrict__ ext_out = nullptr) {
    static_assert(!EXT || WLOG2 == 4, "window extremes: W = 16, windows of four bins");
    constexpr int W = 1 << WLOG2, TP = R * W, NW = TP / 32, NT = 64 * NW;
    constexpr int RR = R + 2, NPIX = RR * W, ZROW = NPIX;               // region pixels; index of the all-zero row
    constexpr int NREG4 = NPIX * 16, NPF = (NREG4 + NT - 1) / NT;      // region float4 slots, per thread
    constexpr int NPL = ONE ? 1 : (FOUR ? 2 : 3);                       // bf16 planes in use
    constexpr int NW4 = NPL * 64 * 8, NWF = ONE ? (512 + NT - 1) / NT : (NW4 + NT - 1) / NT;         // weight uint4 slots per tap, per thread
    constexpr int LD = 64, PLANE = (NPIX + 1) * LD, WBUF = NPL * 64 * LD;      // NPL planes of region and weights: the four-product form's workgroup is a third smaller
    static_assert(TP % 32 == 0, "tile must be whole 32-pixel MFMA row tiles");
    static_assert(ONE || NWF == 4 || NWF == 3 || (FOUR && NWF == 2), "weight staging is two to four named registers");
    static_assert(!ONE || NWF <= 2, "single-product weight staging is one or two named registers");
    extern __shared__ __attribute__((aligned(16))) unsigned short sb_smem[];
    unsigned short* Rp = sb_smem;                          // [NPL][NPIX + 1][LD]
    unsigned short* Wp = sb_smem + NPL * PLANE;            // [2][NPL][64][LD]
    float* red = reinterpret_cast<float*>(sb_smem);        // [NW][128], aliases the region after the last tile
#define SBD_CH(p_, c_) ((c_) ^ (((p_) >> 1) & 7))          /* where 16-byte chunk c_ of row p_ lives */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    const int tiles_per_img = (H + R - 1) / R;
    const int ntiles = B * tiles_per_img;
    int roff[NPF], rrow[NPF];    // float offset relative to the tile's first pixel (negative in the top halo row); region row or -1
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int idx = tid + NT * u;
        const int pix = idx >> 4, g = idx & 15;
        const int rr = pix >> WLOG2, cc = pix & (W - 1);
        rrow[u] = idx < NREG4 ? rr : -1;
        roff[u] = ((rr - 1) * W + cc) * 64 + g * 4;
    }
    for (int i = tid; i < NPL * LD / 2; i += NT) {         // the zero row of every plane (never written again)
        const int pl = i / (LD / 2), e = i - pl * (LD / 2);
        reinterpret_cast<unsigned*>(Rp + pl * PLANE + ZROW * LD)[e] = 0u;
    }
    float4 rreg[NPF];
    u32x4 wreg0, wreg1, wreg2, wreg3;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    // PRE: a thread's region slots all carry the same 4 channels (NT % 16 == 0); where the staged tile's rows go in pre_out
    static_assert(!PRE || NT % 16 == 0, "one channel group per thread");
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PRE) { psc = reinterpret_cast<const float4*>(pre_scale)[tid & 15]; psh = reinterpret_cast<const float4*>(pre_shift)[tid & 15]; }
    float* pre_org = nullptr;
    int pre_t0 = 0;
    bool ext_neg[2] = {false, false};
    if (EXT) { ext_neg[0] = ext_gamma[li] < 0.f; ext_neg[1] = ext_gamma[32 + li] < 0.f; }
#define SBD_ISSUE_REGION(tile_)                                                                         \
    {                                                                                                   \
        const int tb_ = (tile_) / tiles_per_img, tt0_ = ((tile_) - tb_ * tiles_per_img) * R;            \
        const float* org_ = x + (size_t)(tb_ * H + tt0_) * W * 64;                                      \
        _Pragma("unroll") for (int u = 0; u < NPF; ++u) {                                               \
            const int t_ = tt0_ - 1 + rrow[u];                                                          \
            const bool ok_ = rrow[u] >= 0 && t_ >= 0 && t_ < H;                                         \
            float4 v_ = *reinterpret_cast<const float4*>(ok_ ? org_ + roff[u] : x);                     \
            if (PRE) v_ = make_float4(fmaxf(0.f, fmaf(v_.x, psc.x, psh.x)), fmaxf(0.f, fmaf(v_.y, psc.y, psh.y)),  \
                                      fmaxf(0.f, fmaf(v_.z, psc.z, psh.z)), fmaxf(0.f, fmaf(v_.w, psc.w, psh.w)));  \
            rreg[u] = ok_ ? v_ : make_float4(0.f, 0.f, 0.f, 0.f);                                       \
        }                                                                                               \
        if (PRE) { pre_org = pre_out + (size_t)(tb_ * H + tt0_) * W * 64; pre_t0 = tt0_; }              \
    }
#define SBD_COMMIT_REGION()                                                                             \
    _Pragma("unroll") for (int u = 0; u < NPF; ++u) {                                                   \
        const int idx = tid + NT * u;                                                                   \
        /* PRE: the rows this tile OWNS (region rows 1 .. R inside the image) leave as the previous block's pooled tensor */ \
        if (PRE && pre_out && idx < NREG4 && rrow[u] >= 1 && rrow[u] <= R && pre_t0 + rrow[u] - 1 < H)  \
            *reinterpret_cast<float4*>(pre_org + roff[u]) = rreg[u];                                    \
        if (idx < NREG4) {                                                                              \
            unsigned short* d_ = Rp + (idx >> 4) * LD + SBD_CH(idx >> 4, (idx & 15) >> 1) * 8 + (idx & 1) * 4; \
            if (ONE) {                                                                                  \
                *reinterpret_cast<uint2*>(d_) = make_uint2(rne_pair(rreg[u].x, rreg[u].y), rne_pair(rreg[u].z, rreg[u].w)); \
            } else if (FOUR) {                                                                          \
            unsigned h0, m0, h1, m1;                                                                    \
            split2r_pair(rreg[u].x, rreg[u].y, h0, m0);                                                 \
            split2r_pair(rreg[u].z, rreg[u].w, h1, m1);                                                 \
            *reinterpret_cast<uint2*>(d_) = make_uint2(h0, h1);                                         \
            *reinterpret_cast<uint2*>(d_ + PLANE) = make_uint2(m0, m1);                                 \
            } else {                                                                                    \
            unsigned h0, m0, l0, h1, m1, l1;                                                            \
            split3_pair(rreg[u].x, rreg[u].y, h0, m0, l0);                                              \
            split3_pair(rreg[u].z, rreg[u].w, h1, m1, l1);                                              \
            *reinterpret_cast<uint2*>(d_) = make_uint2(h0, h1);                                         \
            *reinterpret_cast<uint2*>(d_ + PLANE) = make_uint2(m0, m1);                                 \
            *reinterpret_cast<uint2*>(d_ + 2 * PLANE) = make_uint2(l0, l1);                             \
            }                                                                                           \
        }                                                                                               \
    }
#define SBD_W_SRC(tap_, u_) \
    reinterpret_cast<const u32x4*>(wsp + (size_t)(tap_) * 3 * 4096)[(tid + NT * (u_)) < NW4 ? (tid + NT * (u_)) : 0]
#define SBD_ISSUE_W(tap_)                                                                               \
    {                                                                                                   \
        wreg0 = SBD_W_SRC(tap_, 0);                                                                     \
        if (NWF > 1) wreg1 = SBD_W_SRC(tap_, 1);                                                        \
        if (NWF > 2) wreg2 = SBD_W_SRC(tap_, 2);                                                        \
        if (NWF > 3) wreg3 = SBD_W_SRC(tap_, 3);                                                        \
    }
#define SBD_W_DST(buf_, u_, v_)                                                                         \
    {                                                                                                   \
        const int idx = tid + NT * (u_);                                                                \
        if (idx < NW4) {                                                                                \
            const int pl = idx >> 9, rem = idx & 511;      /* 512 uint4 per plane: row = rem>>3 */      \
            *reinterpret_cast<u32x4*>(Wp + (buf_) * WBUF + pl * 64 * LD + (rem >> 3) * LD + SBD_CH(rem >> 3, rem & 7) * 8) = v_; \
        }                                                                                               \
    }
#define SBD_COMMIT_W(buf_)                                                                              \
    {                                                                                                   \
        SBD_W_DST(buf_, 0, wreg0)                                                                       \
        if (NWF > 1) SBD_W_DST(buf_, 1, wreg1)                                                          \
        if (NWF > 2) SBD_W_DST(buf_, 2, wreg2)                                                          \
        if (NWF > 3) SBD_W_DST(buf_, 3, wreg3)                                                          \
    }
    int tile = blockIdx.x;
    const int pl_ = wave * 32 + li;                       // this lane's pixel of the tile
    const int pr = pl_ >> WLOG2, pc = pl_ & (W - 1);
    // region row of this lane's A fragment per column shift dx = 0, 1, 2 at tap row 0 (ZROW: outside the image)
    const int prow0 = pc >= 1 ? pr * W + pc - 1 : -1, prow1 = pr * W + pc, prow2 = pc + 1 < W ? pr * W + pc + 1 : -1;
    if (tile < ntiles) {
        SBD_ISSUE_REGION(tile)
        SBD_ISSUE_W(0)
        SBD_COMMIT_REGION()
        SBD_COMMIT_W(0)
        SBD_ISSUE_W(1)                                    // tap 1 rides in the registers until tap 0's loop body commits it
    }
    lds_barrier();
    int wb = 0;                                           // weight buffer that holds the tap being computed
    for (; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * R;
        const int nxt = tile + gridDim.x;
        SBD_ISSUE_REGION(nxt < ntiles ? nxt : tile)     // unconditional: no phi on the staged registers
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[2] = {zero16(), zero16()}, accs[2] = {zero16(), zero16()};
        const int wsw = (li >> 1) & 7;        // swizzle key of weight rows li and li + 32 (the same)
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            // the registers hold tap + 1 (tap 8: tap 0 of the next tile): into the buffer nobody reads in this tap, then fetch tap + 2
            SBD_COMMIT_W(wb ^ 1)
            SBD_ISSUE_W(tap < 7 ? tap + 2 : tap - 7)
            __builtin_amdgcn_sched_barrier(0);
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int pbase_ = dx == 0 ? prow0 : (dx == 1 ? prow1 : prow2);
            const int prow = pbase_ >= 0 ? pbase_ + dy * W : ZROW;
            const unsigned short* arow = Rp + prow * LD;
            const unsigned short* wrow0 = Wp + wb * WBUF + li * LD;
            const int asw = (prow >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 a[3], bb[3][2];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const int ca = (2 * s + hi) ^ asw, cw = (2 * s + hi) ^ wsw;
                    a[pl] = *reinterpret_cast<const bf16x8*>(arow + pl * PLANE + 8 * ca);
                    bb[pl][0] = *reinterpret_cast<const bf16x8*>(wrow0 + pl * 64 * LD + 8 * cw);
                    bb[pl][1] = *reinterpret_cast<const bf16x8*>(wrow0 + pl * 64 * LD + 32 * LD + 8 * cw);
                }
#define SBD_MFMA(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC_, 0, 0, 0)
                SBD_MFMA(a[0], bb[0][0], acc[0]);  SBD_MFMA(a[0], bb[0][1], acc[1]);     // hi*hi
                if (!ONE) {
                SBD_MFMA(a[0], bb[1][0], accs[0]); SBD_MFMA(a[0], bb[1][1], accs[1]);    // hi*mid
                SBD_MFMA(a[1], bb[0][0], acc[0]);  SBD_MFMA(a[1], bb[0][1], acc[1]);     // mid*hi
                if (!FOUR) {
                SBD_MFMA(a[0], bb[2][0], accs[0]); SBD_MFMA(a[0], bb[2][1], accs[1]);    // hi*lo
                SBD_MFMA(a[2], bb[0][0], acc[0]);  SBD_MFMA(a[2], bb[0][1], acc[1]);     // lo*hi
                }
                SBD_MFMA(a[1], bb[1][0], accs[0]); SBD_MFMA(a[1], bb[1][1], accs[1]);    // mid*mid
                }
#undef SBD_MFMA
            }
            lds_barrier();      // tap + 1's weights are visible; everyone is done with this tap's buffer
            wb ^= 1;
        }
        SBD_COMMIT_REGION()     // next tile's region (every wave passed the barrier after tap 8's reads)
        // statistics in the accumulator layout (a lane = one channel), then a 4 x 4 transpose inside each quad of lanes so that the
        // tile leaves as 8 dwordx4 stores per wave instead of 64 dword stores (common.h: quad_transpose4)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float bv = bias ? bias[c * 32 + li] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = (acc[c][4 * q + j] + accs[c][4 * q + j]) + bv;
                    const int t = t0 + ((wave * 32 + 8 * q + 4 * hi + j) >> WLOG2);
                    if (t < H) {
                        s1[c] += v[j];
                        s2[c] = fmaf(v[j], v[j], s2[c]);
                    }
                }
                if (EXT) {      // pixels 8 q + 4 hi .. + 3 of this wave = four consecutive bins of one image row: one window
                    const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), mn = fminf(fminf(v[0], v[1]), fminf(v[2], v[3]));
                    const int p0 = wave * 32 + 8 * q + 4 * hi;
                    const int te = t0 + (p0 >> WLOG2);
                    if (te < H) ext_out[((size_t)(b * H + te) * (W / 4) + ((p0 & (W - 1)) >> 2)) * 64 + c * 32 + li] = ext_neg[c] ? mn : mx;
                }
                const float4 o = quad_transpose4(v[0], v[1], v[2], v[3], li);
                const int p = wave * 32 + 8 * q + 4 * hi + (li & 3);      // pixel of the tile this lane stores
                const int t = t0 + (p >> WLOG2);
                if (t < H) *reinterpret_cast<float4*>(z + ((size_t)(b * H + t) * W + (p & (W - 1))) * 64 + c * 32 + (li & ~3)) = o;
            }
        }
        lds_barrier();          // the committed region is visible to every wave
    }
#undef SBD_ISSUE_REGION
#undef SBD_COMMIT_REGION
#undef SBD_ISSUE_W
#undef SBD_COMMIT_W
#undef SBD_CH
#undef SBD_W_SRC
#undef SBD_W_DST
    if (STATS) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            s1[c] += __shfl_xor(s1[c], 32);
            s2[c] += __shfl_xor(s2[c], 32);
        }
        __syncthreads();
        if (hi == 0) {
            red[wave * 128 + li] = s1[0];
            red[wave * 128 + 32 + li] = s1[1];
            red[wave * 128 + 64 + li] = s2[0];
            red[wave * 128 + 96 + li] = s2[1];
        }
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) t += red[w8 * 128 + tid];
            stat_partial[(size_t)blockIdx.x * 128 + tid] = t;
        }
    }
}

int g_bwd_four = 1;        // option "bwd_four_products": backward-only products (input / kernel gradients) on four of the six split-bf16 terms
static thread_local int s_sbd_four = 0; // set around an input-gradient launch (launch_conv64_dgrad_sb); per host thread: two contexts driven by two threads must not see each other's
int g_sbd_dgrad_r8 = 1;    // option "dgrad_r8": the W = 16 four-product input gradient on 8-row tiles (4 waves, 74 KB of LDS instead of 8 waves, 107 KB: two workgroups per
                           // CU, or one beside a kernel-gradient workgroup of the side stream; the same taps and k-steps per pixel: the same bits)
int g_conv64_dbuf = 1;     // 1: conv64_fwd_sbd_kernel (double-buffered weights) for W = 16 / 4; 0: conv64_fwd_sbr_kernel

template <int WLOG2, int R>
static int launch_sbd(hipStream_t st, const float* x, const unsigned short* wsp, const float* bias, float* z,
                      float* stat_partial, int* n_partial, int B, int H, const float* pre_scale = nullptr, const float* pre_shift = nullptr,
                      float* pre_out = nullptr, const float* ext_gamma = nullptr, float* ext_out = nullptr) {
    constexpr int W = 1 << WLOG2, NW = (R * W) / 32, NT = 64 * NW, NPIX = (R + 2) * W;
    const int ntiles = B * ((H + R - 1) / R);
    const int grid = ntiles < 256 ? ntiles : 256;      // one block per CU (LDS-limited), persistent
    const size_t smem = (size_t)(3 * (NPIX + 1) * 64 + 2 * 3 * 64 * 64) * sizeof(unsigned short);
    static_assert((size_t)(3 * ((R + 2) * (1 << WLOG2) + 1) * 64 + 2 * 3 * 64 * 64) * 2 <= 163840, "LDS");
    static_assert((size_t)NW * 128 * 4 <= (size_t)3 * (NPIX + 1) * 64 * 2, "the statistics buffer aliases the region");
#define SBD_GO(S_, O_)                                                                                                                    \
    {                                                                                                                                     \
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_fwd_sbd_kernel<WLOG2, R, S_, O_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
        hipLaunchKernelGGL((conv64_fwd_sbd_kernel<WLOG2, R, S_, O_>), dim3(grid), dim3(NT), smem, st, x, wsp, bias, z, stat_partial, B, H);   \
    }
    if (pre_scale) {      // BatchNorm + ReLU of the previous block on load (six-product forward; launch_conv64_fwd_sb checks the shapes)
        if (g_mfma_one || !pre_shift || pre_out == x) return -3;      // pre_out == nullptr: inference (nobody reads the activated tensor)
#define SBD_PRE_GO(S_, E_)                                                                                                                  \
        {                                                                                                                                   \
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_fwd_sbd_kernel<WLOG2, R, S_, false, false, true, E_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            hipLaunchKernelGGL((conv64_fwd_sbd_kernel<WLOG2, R, S_, false, false, true, E_>), dim3(grid), dim3(NT), smem, st, x, wsp, bias, z, stat_partial, B, H, pre_scale, pre_shift, \
                               pre_out, ext_gamma, ext_out);                                                                                \
        }
        if constexpr (WLOG2 == 4) {
            if (ext_out && !ext_gamma) return -3;
            if (ext_out) { if (stat_partial) SBD_PRE_GO(true, true) else SBD_PRE_GO(false, true) }
            else if (stat_partial) SBD_PRE_GO(true, false)
            else SBD_PRE_GO(false, false)
        } else if constexpr (WLOG2 == 2) {
            if (ext_out) return -3;
            if (stat_partial) SBD_PRE_GO(true, false) else SBD_PRE_GO(false, false)
        } else
            return -3;
#undef SBD_PRE_GO
        if (n_partial) *n_partial = grid;
        return 0;
    }
    if (ext_out) return -3;      // window extremes come with the PRE loader only (the training step's second block)
    if (g_mfma_one) { if (stat_partial) SBD_GO(true, true) else SBD_GO(false, true) }
    else if (s_sbd_four && !stat_partial) {
        // two planes of region and weights: 106 KB at W = 16, 84 KB at W = 4 — at W = 4 a kernel-gradient workgroup of the side stream (70 KB, option
        // conv_wgrad_side) fits on the same CU beside it
        const size_t smem4 = (size_t)(2 * (NPIX + 1) * 64 + 2 * 2 * 64 * 64) * sizeof(unsigned short);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_fwd_sbd_kernel<WLOG2, R, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4);
        hipLaunchKernelGGL((conv64_fwd_sbd_kernel<WLOG2, R, false, false, true>), dim3(grid), dim3(NT), smem4, st, x, wsp, bias, z, stat_partial, B, H);
    } else { if (stat_partial) SBD_GO(true, false) else SBD_GO(false, false) }
#undef SBD_GO
    if (n_partial) *n_partial = grid;
    return 0;
}

// the four-product form alone on R-row tiles (option "dgrad_r8", round 5: 2.510 -> 2.496 ms per step same box; W = 16, R = 8 -> 4 waves, 74 KB of LDS: two workgroups per CU, or one beside a
// kernel-gradient workgroup of the side stream)
template <int WLOG2, int R>
static int launch_sbd_four(hipStream_t st, const float* x, const unsigned short* wsp, const float* bias, float* z, int* n_partial, int B, int H) {
    constexpr int W = 1 << WLOG2, NW = (R * W) / 32, NT = 64 * NW, NPIX = (R + 2) * W;
    const int ntiles = B * ((H + R - 1) / R);
    const int grid = ntiles < 512 ? ntiles : 512;
    const size_t smem4 = (size_t)(2 * (NPIX + 1) * 64 + 2 * 2 * 64 * 64) * sizeof(unsigned short);
    hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_fwd_sbd_kernel<WLOG2, R, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem4);
    hipLaunchKernelGGL((conv64_fwd_sbd_kernel<WLOG2, R, false, false, true>), dim3(grid), dim3(NT), smem4, st, x, wsp, bias, z, nullptr, B, H);
    if (n_partial) *n_partial = grid;
    return 0;
}

int conv64_fwd_sb_takes_pre(int W) { return (W == 16 || W == 4) && g_conv64_dbuf && !g_mfma_one; }
int launch_conv64_fwd_sb(hipStream_t st, const float* x, const unsigned short* wsp, const float* bias, float* z,
                         float* stat_partial, int* n_partial, int B, int H, int W, const float* pre_scale, const float* pre_shift, float* pre_out,
                         const float* ext_gamma, float* ext_out) {
    if ((pre_scale || ext_out) && !conv64_fwd_sb_takes_pre(W)) return -3;
    if (W == 16 && g_conv64_dbuf && s_sbd_four && g_sbd_dgrad_r8 && !stat_partial && !pre_scale && !ext_out && !g_mfma_one)
        return launch_sbd_four<4, 8>(st, x, wsp, bias, z, n_partial, B, H);      // option "dgrad_r8"
    // (W = 4 on 40-row tiles — 76 KB, two per CU, 480 tiles in one round — measured: 2.331 -> 2.346 ms per step, not kept: profiles/r05_gru_experiments.txt)
    if (W == 16 && g_conv64_dbuf) return launch_sbd<4, 16>(st, x, wsp, bias, z, stat_partial, n_partial, B, H, pre_scale, pre_shift, pre_out, ext_gamma, ext_out);   // 8 waves, 156 KB
    if (W == 4 && g_conv64_dbuf) return launch_sbd<2, 48>(st, x, wsp, bias, z, stat_partial, n_partial, B, H, pre_scale, pre_shift, pre_out);    // 6 waves
    if (W == 8 && g_conv64_dbuf) return launch_sbd<3, 32>(st, x, wsp, bias, z, stat_partial, n_partial, B, H);    // 8 waves (resnet50_block stage 1)
    if (W == 16) return launch_sbr<4, 16, true>(st, x, wsp, bias, z, stat_partial, n_partial, B, H);    // 8 waves, 153 KB
    if (W == 4) return launch_sbr<2, 48, false>(st, x, wsp, bias, z, stat_partial, n_partial, B, H);   // 6 waves (halo columns: no room for 8)
    const int npix = B * H * W;
    const int ntiles = (npix + 255) / 256;
    const int grid = ntiles < SB_MAX_PERSISTENT ? ntiles : SB_MAX_PERSISTENT;
    const size_t smem = (size_t)(SB_A_ELEMS + SB_W_ELEMS) * sizeof(unsigned short) + 8 * 128 * sizeof(float);
    if (stat_partial) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_fwd_sb_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(conv64_fwd_sb_kernel<true>, dim3(grid), dim3(512), smem, st, x, wsp, bias, z, stat_partial, npix, H, W);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_fwd_sb_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(conv64_fwd_sb_kernel<false>, dim3(grid), dim3(512), smem, st, x, wsp, bias, z, stat_partial, npix, H, W);
    }
    if (n_partial) *n_partial = grid;
    return 0;
}

// the input gradient of a 64 -> 64 3x3 convolution = the forward kernel on dz with the flipped, channel-swapped weights (wsp_flip); four products when
// the option allows (the forward itself never takes that path: s_sbd_four is set here only)
int launch_conv64_dgrad_sb(hipStream_t st, const float* dz, const unsigned short* wsp_flip, float* dx, int B, int H, int W) {
    s_sbd_four = g_bwd_four && !g_mfma_one;
    const int rc = launch_conv64_fwd_sb(st, dz, wsp_flip, nullptr, dx, nullptr, nullptr, B, H, W);
    s_sbd_four = 0;
    return rc;
}
