// conv_wgrad_sb.hip — kernel / bias gradient of the 64 -> 64 Conv2D blocks (layers.py:27-32, second and third block of
// seldnet.json) on the bf16 matrix cores with exactly split operands (the scheme of conv_sb.hip):
//     dW[tap][ci][co] = sum_px x[px + tap][ci] * dz[px][co]          db[co] = sum_px dz[px][co]
// The reduction index of this product is the PIXEL, which is the slow axis of both NHWC operands, and a bf16 MFMA fragment
// wants 8 consecutive k per lane.  gfx950's transposed LDS read does that for free: the LDS images stay [pixel][channel]
// (what the global loads deliver, written with 8-byte stores) and ds_read_b64_tr_b16 hands lane i the channel-i column of
// a 4-pixel x 16-channel block, i.e. 4 consecutive k of row i; two of them make a fragment.
//
// Chunk = 64 pixels of one image (4 rows at W = 16, 8 at W = 8, 16 at W = 4) plus the halo: up to 108 pixels of x and 64 of dz, three
// bf16 planes each, 66 KB of LDS -> two workgroups per CU.  Wave (cih, coh) owns the 32 ci x 32 co tile of all 9 taps
// (144 accumulator registers), exactly as conv64_wgrad_kernel; a k-step is 16 pixels: 6 transposed reads for the dz
// fragments, 6 per tap for x, 54 MFMAs.
// Bank conflicts: a transposed read covers 4 pixel rows x 64 B per 32-lane half; with 128-B pixel rows the rows two
// apart would share banks, so the two 64-B halves of a pixel row are swapped when bit 1 of the pixel index is set
// (any 4 consecutive pixels then cover the 64 banks exactly).
// Output: the slab layout of conv64_wgrad_kernel ([9][64 ci][64 co] + [64] per workgroup), reduced by reduce_slabs.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define WGSB_MAX_BLOCKS 512
#define WGSB_SLAB (9 * 4096 + 64)
#define WGSB_PXC 64            // pixels per chunk
#define WGSB_HALO 108          // (R + 2) * (W + 2) for W = 16 (6 x 18) and W = 4 (18 x 6); W = 8 uses 100 (10 x 10) of it

__device__ __forceinline__ void wgsb_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
}

// 4 pixels x 16 channels, transposed: this lane's address is (pixel row q of the block, 4 of its channels); it receives its
// own channel column for the 4 pixels
__device__ __forceinline__ s16x4 wgsb_tr(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 wgsb_frag(const char* p0, const char* p1) {
    const s16x4 a = wgsb_tr(p0), b = wgsb_tr(p1);
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ unsigned wgsb_rne_pair(float x0, float x1) { return bf16_rne_bits(x0) | (bf16_rne_bits(x1) << 16); }

// ONE: bf16 single-product mode (common.h g_mfma_one): operands rounded to nearest bf16, plane 0 only, one MFMA per (k-step, tap)
// FOUR (option "bwd_four_products", conv_sb.hip g_bwd_four): the two products with a lo factor are dropped and the lo planes neither staged nor read
template <int WLOG2, bool ONE, bool FOUR = false>
__global__ __launch_bounds__(256, 2) void conv64_wgrad_sb_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                 float* __restrict__ slab, int B, int H) {
    constexpr int W = 1 << WLOG2, R = WGSB_PXC / W, RW = W + 2, RR = R + 2;
    static_assert(RR * RW <= WGSB_HALO, "halo size");
    static_assert(W == 16 || W == 8 || W == 4, "k-step geometry below is written for these widths");
    constexpr int XPL = WGSB_HALO * 128, DPL = WGSB_PXC * 128;        // plane sizes in bytes
    extern __shared__ __attribute__((aligned(16))) char wgsb_smem[];
    char* xl = wgsb_smem;                 // [3][108 px][128 B]
    char* dl = wgsb_smem + 3 * XPL;       // [3][64 px][128 B]
    float* red = reinterpret_cast<float*>(dl + 3 * DPL);   // [16][64] bias partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cih = wave >> 1, coh = wave & 1;
    const int kg = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int chunks_per_img = (H + R - 1) / R;
    const int nchunks = B * chunks_per_img;
    // ---- per-lane read offsets.  Pixel P of an image lives at P*128 + (chb ^ 64*((P>>1)&1)).  A read's pixel is
    // P0 + q with P0 = (compile-time part) + (kg part): bit 1 of a sum depends on the addends mod 4 only, so the swizzle bit is that
    // of ((P0c & 3) + kg part + q) and the lane keeps one offset per value of P0c & 3.
    // A k-step is 16 pixels: W = 16 one image row (kg = its halves), W = 8 two rows (kg = the row), W = 4 four rows (kg = the row pair).
    constexpr int KGPIX = W == 16 ? 8 : W == 8 ? RW : 2 * RW;           // pixels between the kg = 0 and kg = 1 halves of a k-step (x image)
    const int chbA = 64 * cih + 32 * g1 + 8 * p, chbB = 64 * coh + 32 * g1 + 8 * p;
    int offA[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) offA[m] = (kg * KGPIX + q) * 128 + (chbA ^ (64 * (((m + q + kg * KGPIX) >> 1) & 1)));
    const int offB = (8 * kg + q) * 128 + (chbB ^ (64 * ((q >> 1) & 1)));
    f32x16 acc0 = zero16(), acc1 = zero16(), acc2 = zero16(), acc3 = zero16(), acc4 = zero16(), acc5 = zero16(), acc6 = zero16(),
           acc7 = zero16(), acc8 = zero16();
    float4 brun = make_float4(0.f, 0.f, 0.f, 0.f);     // this thread's 4 channels (tid & 15) of the bias gradient
#define WGSB_MFMA(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC_, 0, 0, 0)
    // compile-time pixel offset of the x block of k-step s, half h (pixels 8 kg + 4 h ...), tap (dy, dx), kg part excluded
#define WGSB_P0(s_, h_, dy_, dx_) (W == 16 ? ((s_) + (dy_)) * RW + 4 * (h_) + (dx_) : W == 8 ? (2 * (s_) + (dy_)) * RW + 4 * (h_) + (dx_) \
                                           : (4 * (s_) + (h_) + (dy_)) * RW + (dx_))
#define WGSB_AADDR(s_, h_, dy_, dx_, pl_) (xl + (pl_) * XPL + WGSB_P0(s_, h_, dy_, dx_) * 128 + offA[WGSB_P0(s_, h_, dy_, dx_) & 3])
    // One (k-step, tap) = 6 MFMAs on that tap's accumulator.  Operands are pipelined one step ahead in named registers and
    // the order is pinned with sched_barrier: left alone, the scheduler hoists a whole k-step of transposed reads (108
    // registers) above the MFMAs and spills the accumulators.
#define WGSB_LDA(s_, tap_, H_, M_, L_)                                                                                \
    H_ = wgsb_frag(WGSB_AADDR(s_, 0, (tap_) / 3, (tap_) % 3, 0), WGSB_AADDR(s_, 1, (tap_) / 3, (tap_) % 3, 0));       \
    if (!ONE) {                                                                                                       \
    M_ = wgsb_frag(WGSB_AADDR(s_, 0, (tap_) / 3, (tap_) % 3, 1), WGSB_AADDR(s_, 1, (tap_) / 3, (tap_) % 3, 1));       \
    if (!FOUR) L_ = wgsb_frag(WGSB_AADDR(s_, 0, (tap_) / 3, (tap_) % 3, 2), WGSB_AADDR(s_, 1, (tap_) / 3, (tap_) % 3, 2)); \
    }
#define WGSB_LDB(s_, H_, M_, L_)                                                                                      \
    {                                                                                                                 \
        const char* bp_ = dl + (16 * (s_)) * 128 + offB;                                                              \
        H_ = wgsb_frag(bp_, bp_ + 4 * 128);                                                                           \
        if (!ONE) {                                                                                                   \
        M_ = wgsb_frag(bp_ + DPL, bp_ + DPL + 4 * 128);                                                               \
        if (!FOUR) L_ = wgsb_frag(bp_ + 2 * DPL, bp_ + 2 * DPL + 4 * 128);                                            \
        }                                                                                                             \
    }
#define WGSB_TAP(s_, tap_, ACC_)                                                                                      \
    {                                                                                                                 \
        constexpr bool last_ = (s_) == 3 && (tap_) == 8;                                                              \
        constexpr int ns_ = (tap_) == 8 ? ((s_) < 3 ? (s_) + 1 : 3) : (s_), nt_ = (tap_) == 8 ? 0 : (tap_) + 1;       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        if (!last_) { WGSB_LDA(ns_, nt_, nh, nm, nl) }                                                                \
        if (!last_ && (tap_) == 8) WGSB_LDB(ns_, nbh, nbm, nbl)                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        if (!ONE && !FOUR) WGSB_MFMA(al, bh, ACC_);                                                                   \
        if (!ONE) { WGSB_MFMA(am, bh, ACC_); WGSB_MFMA(am, bm, ACC_); }                                               \
        WGSB_MFMA(ah, bh, ACC_);                                                                                      \
        if (!ONE) WGSB_MFMA(ah, bm, ACC_);                                                                            \
        if (!ONE && !FOUR) WGSB_MFMA(ah, bl, ACC_);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        ah = nh; am = nm; al = nl;                                                                                    \
        if ((tap_) == 8) { bh = nbh; bm = nbm; bl = nbl; }                                                            \
    }
#define WGSB_KSTEP(s_)                                                                                                \
    WGSB_TAP(s_, 0, acc0) WGSB_TAP(s_, 1, acc1) WGSB_TAP(s_, 2, acc2) WGSB_TAP(s_, 3, acc3) WGSB_TAP(s_, 4, acc4)     \
    WGSB_TAP(s_, 5, acc5) WGSB_TAP(s_, 6, acc6) WGSB_TAP(s_, 7, acc7) WGSB_TAP(s_, 8, acc8)
    // ---- staging: float4 = 4 channels of a pixel -> split -> 8 B per plane.  All loads of a chunk are issued back to back
    // (unconditionally: clamped pointer + AND mask, so no branch separates them), then committed.
    constexpr int NXV = (WGSB_HALO * 16 + 255) / 256;     // 7 float4 of x per thread (the last one partial)
    float4 xv[NXV], dv[WGSB_PXC / 16];
#define WGSB_ISSUE(chunk_)                                                                                            \
    {                                                                                                                 \
        const int b_ = (chunk_) / chunks_per_img, t0_ = ((chunk_) - b_ * chunks_per_img) * R;                         \
        _Pragma("unroll") for (int u = 0; u < NXV; ++u) {                                                             \
            const int idx = tid + 256 * u;                                                                            \
            const int g = idx & 15, P = idx >> 4;                                                                     \
            const int rr = P / RW, cc = P - rr * RW;                                                                  \
            const int t = t0_ - 1 + rr, f = cc - 1;                                                                   \
            const bool ok = P < RR * RW && t >= 0 && t < H && f >= 0 && f < W;                                      \
            const float4 v = *reinterpret_cast<const float4*>(ok ? x + ((size_t)(b_ * H + t) * W + f) * 64 + g * 4 : x); \
            const unsigned k_ = ok ? 0xffffffffu : 0u;                                                                \
            xv[u] = make_float4(__uint_as_float(__float_as_uint(v.x) & k_), __uint_as_float(__float_as_uint(v.y) & k_), \
                                __uint_as_float(__float_as_uint(v.z) & k_), __uint_as_float(__float_as_uint(v.w) & k_)); \
        }                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < WGSB_PXC / 16; ++u) {                                                   \
            const int idx = tid + 256 * u;      /* float4 index: pixel idx >> 4, channels 4 (idx & 15) ... */         \
            const bool ok = t0_ + ((idx >> 4) >> WLOG2) < H;                                                          \
            const float4 v = *(ok ? reinterpret_cast<const float4*>(dz + (size_t)(b_ * H + t0_) * W * 64) + idx       \
                                  : reinterpret_cast<const float4*>(dz));                                             \
            const unsigned k_ = ok ? 0xffffffffu : 0u;                                                                \
            dv[u] = make_float4(__uint_as_float(__float_as_uint(v.x) & k_), __uint_as_float(__float_as_uint(v.y) & k_), \
                                __uint_as_float(__float_as_uint(v.z) & k_), __uint_as_float(__float_as_uint(v.w) & k_)); \
        }                                                                                                             \
    }
#define WGSB_PUT(v_, base_, PL_, P_, g_)                                                                              \
    {                                                                                                                 \
        char* d = (base_) + (P_) * 128 + ((8 * (g_)) ^ (64 * (((P_) >> 1) & 1)));                                     \
        if (ONE) {                                                                                                    \
            *reinterpret_cast<uint2*>(d) = make_uint2(wgsb_rne_pair(v_.x, v_.y), wgsb_rne_pair(v_.z, v_.w));          \
        } else if (FOUR) {                                                                                            \
        unsigned h0, m0, h1, m1;                                                                                      \
        split2r_pair(v_.x, v_.y, h0, m0);                                                                             \
        split2r_pair(v_.z, v_.w, h1, m1);                                                                             \
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);                                                            \
        *reinterpret_cast<uint2*>(d + (PL_)) = make_uint2(m0, m1);                                                    \
        } else {                                                                                                      \
        unsigned h0, m0, l0, h1, m1, l1;                                                                              \
        wgsb_split3_pair(v_.x, v_.y, h0, m0, l0);                                                                     \
        wgsb_split3_pair(v_.z, v_.w, h1, m1, l1);                                                                     \
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);                                                            \
        *reinterpret_cast<uint2*>(d + (PL_)) = make_uint2(m0, m1);                                                    \
        *reinterpret_cast<uint2*>(d + 2 * (PL_)) = make_uint2(l0, l1);                                                \
        }                                                                                                             \
    }
#define WGSB_COMMIT()                                                                                                 \
    {                                                                                                                 \
        _Pragma("unroll") for (int u = 0; u < NXV; ++u) {                                                             \
            const int idx = tid + 256 * u;                                                                            \
            if (idx < WGSB_HALO * 16) WGSB_PUT(xv[u], xl, XPL, idx >> 4, idx & 15)                                    \
        }                                                                                                             \
        _Pragma("unroll") for (int u = 0; u < WGSB_PXC / 16; ++u) {                                                   \
            const int idx = tid + 256 * u;                                                                            \
            brun.x += dv[u].x; brun.y += dv[u].y; brun.z += dv[u].z; brun.w += dv[u].w;                               \
            WGSB_PUT(dv[u], dl, DPL, idx >> 4, idx & 15)                                                              \
        }                                                                                                             \
    }
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        // the staged values are not held across the multiplication (144 accumulator registers leave no room: prefetching
        // them spilled 93 VGPRs); the CU's other workgroup multiplies while this one loads
        WGSB_ISSUE(chunk)
        __syncthreads();                      // everyone is done reading the previous chunk
        WGSB_COMMIT()
        __syncthreads();
        bf16x8 ah = {}, am = {}, al = {}, bh = {}, bm = {}, bl = {}, nh, nm, nl, nbh, nbm, nbl;
        WGSB_LDB(0, bh, bm, bl)
        WGSB_LDA(0, 0, ah, am, al)
        nh = ah; nm = am; nl = al; nbh = bh; nbm = bm; nbl = bl;
        WGSB_KSTEP(0) WGSB_KSTEP(1) WGSB_KSTEP(2) WGSB_KSTEP(3)
    }
#undef WGSB_ISSUE
#undef WGSB_COMMIT
#undef WGSB_PUT
#undef WGSB_KSTEP
#undef WGSB_TAP
#undef WGSB_LDA
#undef WGSB_LDB
#undef WGSB_AADDR
#undef WGSB_P0
#undef WGSB_MFMA
    float* out = slab + (size_t)blockIdx.x * WGSB_SLAB;
    const int hi = kg, li = lane & 31;
#define WGSB_OUT(tap_, ACC_)                                                                   \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                             \
        out[(tap_) * 4096 + (cih * 32 + mfma_row(r, hi)) * 64 + coh * 32 + li] = ACC_[r];
    WGSB_OUT(0, acc0) WGSB_OUT(1, acc1) WGSB_OUT(2, acc2) WGSB_OUT(3, acc3) WGSB_OUT(4, acc4)
    WGSB_OUT(5, acc5) WGSB_OUT(6, acc6) WGSB_OUT(7, acc7) WGSB_OUT(8, acc8)
#undef WGSB_OUT
    // bias gradient: thread (tid >> 4, tid & 15) holds channels 4 (tid & 15) ... of its pixels; 16 partials per channel
    __syncthreads();
    reinterpret_cast<float4*>(red)[tid] = brun;      // red[tid >> 4][4 (tid & 15) + e]
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += red[i * 64 + tid];
        out[9 * 4096 + tid] = s;
    }
}

int conv64_wgrad_sb_usable(int W) { return W == 16 || W == 8 || W == 4; }

int launch_conv64_wgrad_sb(hipStream_t st, const float* x, const float* dz, float* slab, int* n_slab, int B, int H, int W) {
    if (!conv64_wgrad_sb_usable(W) || B <= 0 || H <= 0) return -2;
    const int R = WGSB_PXC / W;
    const int nchunks = B * ((H + R - 1) / R);
    // W = 4 (third block, 39 MB of operands): one workgroup per CU — 256 slabs of 148 KB instead of 512 halve the slab traffic
    // (75.6 -> 37.8 MB written, and read again by the combine) and the kernel is 10 % FASTER for it (0.0495 -> 0.0445 ms, same box);
    // W = 16 (157 MB of operands) keeps two workgroups per CU: with one it is 11 % slower
    const int cap_ = W == 4 ? WGSB_MAX_BLOCKS / 2 : WGSB_MAX_BLOCKS;
    const int grid = nchunks < cap_ ? nchunks : cap_;
    const size_t smem = (size_t)3 * WGSB_HALO * 128 + (size_t)3 * WGSB_PXC * 128 + 16 * 64 * sizeof(float);
#define WGSB_GO(L)                                                                                            \
    {                                                                                                         \
        if (g_mfma_one) {                                                                                     \
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_wgrad_sb_kernel<L, true>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                       \
            hipLaunchKernelGGL((conv64_wgrad_sb_kernel<L, true>), dim3(grid), dim3(256), smem, st, x, dz, slab, B, H);  \
        } else if (g_bwd_four) {                                                                              \
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_wgrad_sb_kernel<L, false, true>),        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                       \
            hipLaunchKernelGGL((conv64_wgrad_sb_kernel<L, false, true>), dim3(grid), dim3(256), smem, st, x, dz, slab, B, H); \
        } else {                                                                                              \
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_wgrad_sb_kernel<L, false>),              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                       \
            hipLaunchKernelGGL((conv64_wgrad_sb_kernel<L, false>), dim3(grid), dim3(256), smem, st, x, dz, slab, B, H); \
        }                                                                                                     \
    }
    if (W == 16) WGSB_GO(4) else if (W == 8) WGSB_GO(3) else WGSB_GO(2)
#undef WGSB_GO
    *n_slab = grid;
    return 0;
}
