// gemm_tn_sb.hip — weight gradients of the GRU kernels on split-bf16 MFMA:  C[128][N] = sum_m A[m][0..127]^T B[m][0..N-1]
// (+ colsum(B), the bias gradient), A = the layer input or the shifted hidden state, B = the gate gradients dgx / dgh
// (modules.py:311-316 under tape.gradient).  Like the conv kernel gradients (conv_wgrad_sb.hip) the reduction index is the
// slow axis (rows) of both operands: LDS keeps [row][channel] bf16 planes (three per operand, split on the way in) and
// ds_read_b64_tr_b16 supplies the transposed fragments.  The f32-input TN kernel (gemm.hip) ran at half its MFMA floor and
// — for the FIRST GRU layer, whose weight gradients cannot hide under a recurrence — every microsecond of it lands on the
// main stream's conv backward.
//
// Block = 4 waves, tile = 128 (k1) x 128 (n), wave (kh, nh) = 64 x 64 = four accumulator tiles; chunk = 32 rows: 48 KB of
// LDS (3 blocks per CU), the next chunk's 8 float4 in flight under the current chunk's 48 MFMAs per wave.  256-B image rows;
// the 64-B unit u of row m is stored at u ^ (m & 3): the 4 rows of a transposed-read block then cover the 64 banks.
// Splits over rows write slabs in gemm_tn's layout ([128*N | N] per split), combined by reduce_slabs2 in a fixed order.
#include "common.h"
#include <algorithm>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tnsb_split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302);
}
__device__ __forceinline__ bf16x8 tnsb_frag(const char* p0, const char* p1) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

#define TNSB_PL (32 * 256)     // bytes per plane: 32 rows x 128 bf16

// blockIdx.y = job: up to TN_MAX_JOBS products of one shape (the four weight gradients of a GRU layer) in one launch; job j's slabs
// follow job j-1's (gridDim.z slabs each)
// tile_stride > 0 (the convolution kernel gradients of resnet50_block, K1 = a multiple of 128): ONE product, blockIdx.y = the
// 128-row tile of C (columns 128 y .. of A); a split's slab is the whole [K1][N] matrix, tile_stride floats apart
// cvs = log2 C > 0 (tile mode only): A is an NHWC image [M pixels][C] and the product's A matrix its virtual im2col [M][9 C] (3 x 3
// 'same', zeros outside the H x W image): tile y covers columns 128 y .. of it, i.e. ONE tap (C % 128 == 0) at channel offset
// (128 y) % C — the row of the image is the pixel shifted by the tap, masked at the image border.
// ONE: bf16 single-product mode (common.h g_mfma_one): both operands rounded to nearest bf16, plane 0 only, one MFMA per tile pair
// FOUR (option "bwd_four_products", conv_sb.hip g_bwd_four): weight gradients on four products — the two with a lo factor dropped, two planes per image
template <bool CONV, bool ONE = false, bool FOUR = false>      // a template parameter: the GRU's instantiation (the headline step's side stream) must not carry the convolution's index work
__global__ __launch_bounds__(256, 3) void gemm_tn_sb_kernel(TnJobs jobs, int ldb, float* __restrict__ slab, int M, int N, int rows_per_split,
                                                            int S, int want_bias, long long tile_stride, int cvs, int cvH, int cvW) {
    const int job = tile_stride ? 0 : blockIdx.y;
    const int cv_tap = CONV ? (128 * (int)blockIdx.y) >> cvs : 0;
    const int cv_dy = cv_tap / 3 - 1, cv_dx = cv_tap - 3 * (cv_tap / 3) - 1;
    const float* __restrict__ A = jobs.A[job] + (CONV ? 128 * (int)blockIdx.y - (cv_tap << cvs) : tile_stride ? 128 * blockIdx.y : 0);
    const float* __restrict__ Bm = jobs.B[job];
    const int lda = CONV ? 1 << cvs : jobs.lda[job], shift = jobs.shift[job];
    __shared__ __attribute__((aligned(16))) char Al[(FOUR ? 2 : 3) * TNSB_PL];
    __shared__ __attribute__((aligned(16))) char Bl[(FOUR ? 2 : 3) * TNSB_PL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = wave >> 1, nh = wave & 1;
    const int kg = lane >> 5, g1 = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    const int n0 = blockIdx.x * 128;
    const int mbeg = blockIdx.z * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    // transposed-read offsets: block row q = image row (8 kg + 4 h + q) of a k-step, 4 columns at (32 tile + 16 g1 + 4 p)
    const int lrow = (8 * kg + q) * 256 + 32 * g1 + 8 * p;
    const int oa0 = lrow + (((2 * kh) ^ q) << 6), oa1 = lrow + (((2 * kh + 1) ^ q) << 6);
    const int ob0 = lrow + (((2 * nh) ^ q) << 6), ob1 = lrow + (((2 * nh + 1) ^ q) << 6);
    // staging role: 4 float4 of A and of B per chunk: slot = tid + 256 u -> row slot >> 5, float4 (slot & 31) = tid & 31
    const int c4 = tid & 31, r0s = tid >> 5;
    float4 va[4], vb[4];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);      // this thread's 4 columns of colsum(B)
#define TNSB_LOAD(mm0_)                                                                                      \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                          \
        const int gm = (mm0_) + r0s + 8 * u;                                                                 \
        const bool inb = gm < mend;                                                                          \
        int am = gm;                                                                                         \
        bool oka = inb;                                                                                      \
        if (shift != 0) {   /* A row with the time shift (H_prev for the recurrent-kernel gradient) */       \
            const int t = gm % S;                                                                            \
            oka = inb && (t + shift >= 0) && (t + shift < S);                                                \
            am = gm + shift;                                                                                 \
        }                                                                                                    \
        if (CONV) {         /* A row = the pixel shifted by this tile's tap */                               \
            const int pr_ = gm / cvW, f_ = gm - pr_ * cvW, t_ = pr_ % cvH;                                   \
            oka = inb && (unsigned)(t_ + cv_dy) < (unsigned)cvH && (unsigned)(f_ + cv_dx) < (unsigned)cvW;   \
            am = gm + cv_dy * cvW + cv_dx;                                                                   \
        }                                                                                                    \
        const float4 ta = *reinterpret_cast<const float4*>(A + (size_t)(oka ? am : 0) * lda + 4 * c4);       \
        const float4 tb = *reinterpret_cast<const float4*>(Bm + (size_t)(inb ? gm : 0) * ldb + n0 + 4 * c4); \
        const unsigned ka = oka ? 0xffffffffu : 0u, kb = inb ? 0xffffffffu : 0u;                             \
        va[u] = make_float4(__uint_as_float(__float_as_uint(ta.x) & ka), __uint_as_float(__float_as_uint(ta.y) & ka), \
                            __uint_as_float(__float_as_uint(ta.z) & ka), __uint_as_float(__float_as_uint(ta.w) & ka)); \
        vb[u] = make_float4(__uint_as_float(__float_as_uint(tb.x) & kb), __uint_as_float(__float_as_uint(tb.y) & kb), \
                            __uint_as_float(__float_as_uint(tb.z) & kb), __uint_as_float(__float_as_uint(tb.w) & kb)); \
    }
#define TNSB_PUT(v_, img_, m_)                                                                               \
    {                                                                                                        \
        char* d = (img_) + (m_) * 256 + ((((c4 >> 3) ^ ((m_) & 3)) << 6) | (8 * (c4 & 7)));                  \
        if (ONE) {                                                                                           \
            *reinterpret_cast<uint2*>(d) = make_uint2(bf16_rne_bits(v_.x) | (bf16_rne_bits(v_.y) << 16), bf16_rne_bits(v_.z) | (bf16_rne_bits(v_.w) << 16)); \
        } else if (FOUR) {                                                                                   \
        unsigned h0, m0, h1, m1;                                                                             \
        split2r_pair(v_.x, v_.y, h0, m0);                                                                    \
        split2r_pair(v_.z, v_.w, h1, m1);                                                                    \
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);                                                   \
        *reinterpret_cast<uint2*>(d + TNSB_PL) = make_uint2(m0, m1);                                         \
        } else {                                                                                             \
        unsigned h0, m0, l0, h1, m1, l1;                                                                     \
        tnsb_split3_pair(v_.x, v_.y, h0, m0, l0);                                                            \
        tnsb_split3_pair(v_.z, v_.w, h1, m1, l1);                                                            \
        *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);                                                   \
        *reinterpret_cast<uint2*>(d + TNSB_PL) = make_uint2(m0, m1);                                         \
        *reinterpret_cast<uint2*>(d + 2 * TNSB_PL) = make_uint2(l0, l1);                                     \
        }                                                                                                    \
    }
#define TNSB_COMMIT()                                                                                        \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                          \
        TNSB_PUT(va[u], Al, r0s + 8 * u)                                                                     \
        TNSB_PUT(vb[u], Bl, r0s + 8 * u)                                                                     \
        bsum.x += vb[u].x; bsum.y += vb[u].y; bsum.z += vb[u].z; bsum.w += vb[u].w;                          \
    }
    f32x16 c00 = zero16(), c01 = zero16(), c10 = zero16(), c11 = zero16();      // [k1 tile][n tile] of this wave
#define TNSB_MFMA(A_, B_, ACC_) ACC_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, ACC_, 0, 0, 0)
#define TNSB_PAIR(AH, AM, AL, BH, BM_, BL, ACC_)                                                             \
    if (!FOUR) TNSB_MFMA(AL, BH, ACC_);                                                                      \
    TNSB_MFMA(AM, BH, ACC_); TNSB_MFMA(AM, BM_, ACC_);                                                       \
    TNSB_MFMA(AH, BH, ACC_); TNSB_MFMA(AH, BM_, ACC_);                                                       \
    if (!FOUR) TNSB_MFMA(AH, BL, ACC_);
#define TNSB_KSTEP(s_)                                                                                       \
    if (ONE) {                                                                                               \
        constexpr int o_ = (16 * (s_)) * 256;                                                                \
        const bf16x8 a0h = tnsb_frag(Al + o_ + oa0, Al + o_ + oa0 + 4 * 256), a1h = tnsb_frag(Al + o_ + oa1, Al + o_ + oa1 + 4 * 256); \
        const bf16x8 b0h = tnsb_frag(Bl + o_ + ob0, Bl + o_ + ob0 + 4 * 256), b1h = tnsb_frag(Bl + o_ + ob1, Bl + o_ + ob1 + 4 * 256); \
        TNSB_MFMA(a0h, b0h, c00); TNSB_MFMA(a0h, b1h, c01); TNSB_MFMA(a1h, b0h, c10); TNSB_MFMA(a1h, b1h, c11);                        \
    } else                                                                                                   \
    {                                                                                                        \
        constexpr int o_ = (16 * (s_)) * 256;                                                                \
        const bf16x8 a0h = tnsb_frag(Al + o_ + oa0, Al + o_ + oa0 + 4 * 256), a0m = tnsb_frag(Al + TNSB_PL + o_ + oa0, Al + TNSB_PL + o_ + oa0 + 4 * 256), \
                     a0l = FOUR ? a0m : tnsb_frag(Al + 2 * TNSB_PL + o_ + oa0, Al + 2 * TNSB_PL + o_ + oa0 + 4 * 256);    \
        const bf16x8 a1h = tnsb_frag(Al + o_ + oa1, Al + o_ + oa1 + 4 * 256), a1m = tnsb_frag(Al + TNSB_PL + o_ + oa1, Al + TNSB_PL + o_ + oa1 + 4 * 256), \
                     a1l = FOUR ? a1m : tnsb_frag(Al + 2 * TNSB_PL + o_ + oa1, Al + 2 * TNSB_PL + o_ + oa1 + 4 * 256);    \
        const bf16x8 b0h = tnsb_frag(Bl + o_ + ob0, Bl + o_ + ob0 + 4 * 256), b0m = tnsb_frag(Bl + TNSB_PL + o_ + ob0, Bl + TNSB_PL + o_ + ob0 + 4 * 256), \
                     b0l = FOUR ? b0m : tnsb_frag(Bl + 2 * TNSB_PL + o_ + ob0, Bl + 2 * TNSB_PL + o_ + ob0 + 4 * 256);    \
        const bf16x8 b1h = tnsb_frag(Bl + o_ + ob1, Bl + o_ + ob1 + 4 * 256), b1m = tnsb_frag(Bl + TNSB_PL + o_ + ob1, Bl + TNSB_PL + o_ + ob1 + 4 * 256), \
                     b1l = FOUR ? b1m : tnsb_frag(Bl + 2 * TNSB_PL + o_ + ob1, Bl + 2 * TNSB_PL + o_ + ob1 + 4 * 256);    \
        TNSB_PAIR(a0h, a0m, a0l, b0h, b0m, b0l, c00) TNSB_PAIR(a0h, a0m, a0l, b1h, b1m, b1l, c01)            \
        TNSB_PAIR(a1h, a1m, a1l, b0h, b0m, b0l, c10) TNSB_PAIR(a1h, a1m, a1l, b1h, b1m, b1l, c11)            \
    }
    TNSB_LOAD(mbeg)
    for (int mm0 = mbeg; mm0 < mend; mm0 += 32) {
        __syncthreads();                 // every wave is done with the previous chunk's images
        TNSB_COMMIT()
        __syncthreads();
        TNSB_LOAD(mm0 + 32)              // rows past mend load row 0 and are masked to zero
        __builtin_amdgcn_sched_barrier(0);
        TNSB_KSTEP(0)
        TNSB_KSTEP(1)
    }
#undef TNSB_KSTEP
#undef TNSB_PAIR
#undef TNSB_MFMA
#undef TNSB_COMMIT
#undef TNSB_PUT
#undef TNSB_LOAD
    float* out = tile_stride ? slab + (size_t)blockIdx.z * tile_stride + (size_t)blockIdx.y * 128 * N
                             : slab + ((size_t)blockIdx.y * gridDim.z + blockIdx.z) * ((size_t)128 * N + N);
    const int li = lane & 31;
    // 4 dwordx4 stores per accumulator tile (common.h: quad_transpose4): row 8 q + 4 kg + (li & 3), columns 4 (li >> 2) .. + 3
#define TNSB_OUT(ACC_, kt_, nt_)                                                                      \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                     \
        *reinterpret_cast<float4*>(out + (size_t)(64 * kh + 32 * (kt_) + 8 * q + 4 * kg + (li & 3)) * N + n0 + 64 * nh + 32 * (nt_) + (li & ~3)) = \
            quad_transpose4(ACC_[4 * q], ACC_[4 * q + 1], ACC_[4 * q + 2], ACC_[4 * q + 3], li);
    TNSB_OUT(c00, 0, 0) TNSB_OUT(c01, 0, 1) TNSB_OUT(c10, 1, 0) TNSB_OUT(c11, 1, 1)
#undef TNSB_OUT
    if (want_bias) {      // colsum(B): 8 row groups x 32 float4 columns -> fixed-order sum through LDS
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(Al);
        red[tid] = bsum;
        __syncthreads();
        if (tid < 32) {
            float4 s = red[tid];
#pragma unroll
            for (int i = 1; i < 8; ++i) { const float4 t = red[tid + 32 * i]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
            *reinterpret_cast<float4*>(out + (size_t)128 * N + n0 + 4 * tid) = s;
        }
    }
}

// K1 = 128, N % 128 == 0, 16-byte aligned operands with leading dimensions % 4 == 0; same slab layout / n_slab as launch_gemm_tn
int g_tn_lds_floor_kb = 0;
int g_tn_tile_blocks = 384;      // tile mode: workgroups per launch the split count aims at (option "tn_tile_blocks"; resnet50_gru same box: 14.709 / 14.752 ms per step at 768, 14.646 at 384, 14.691 at 512, 14.87 at 256 and 1024)
int gemm_tn_sb_usable(const void* A, int lda, const void* Bm, int ldb, int K1, int N) {
    return K1 == 128 && (N % 128) == 0 && (lda & 3) == 0 && (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(Bm) & 15) == 0;
}
int launch_gemm_tn_sb_batch(hipStream_t st, const TnJobs& jobs, int njobs, int ldb, float* slab, int* nslab, int M, int N, int S, int want_bias) {
    if (M <= 0 || njobs < 1 || njobs > TN_MAX_JOBS) return -1;
    for (int j = 0; j < njobs; ++j)
        if (!gemm_tn_sb_usable(jobs.A[j], jobs.lda[j], jobs.B[j], ldb, 128, N)) return -1;
    int splits = (M + 159) / 160;
    if (splits > gemm_tn_max_splits()) splits = gemm_tn_max_splits();
    int rps = (M + splits - 1) / splits;
    rps = (rps + 31) / 32 * 32;
    splits = (M + rps - 1) / rps;
    // g_tn_lds_floor_kb (option "tn_lds_floor", experiment): extra dynamic LDS per workgroup so that these side-stream blocks (48 KB static) cannot land on a
    // CU that holds a GRU recurrence block (49 / 57 KB of the 160): the batch launches run under the next layer's BPTT
    const size_t dyn = (size_t)g_tn_lds_floor_kb * 1024;
    if (g_mfma_one) {
        if (dyn) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_sb_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((gemm_tn_sb_kernel<false, true>), dim3(N / 128, njobs, splits), dim3(256), dyn, st, jobs, ldb, slab, M, N, rps, S > 0 ? S : M, want_bias, 0LL, 0, 0, 0);
    } else if (g_bwd_four) {
        if (dyn) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_sb_kernel<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((gemm_tn_sb_kernel<false, false, true>), dim3(N / 128, njobs, splits), dim3(256), dyn, st, jobs, ldb, slab, M, N, rps, S > 0 ? S : M, want_bias, 0LL, 0, 0, 0);
    } else {
        if (dyn) hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_sb_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((gemm_tn_sb_kernel<false, false>), dim3(N / 128, njobs, splits), dim3(256), dyn, st, jobs, ldb, slab, M, N, rps, S > 0 ? S : M, want_bias, 0LL, 0, 0, 0);
    }
    *nslab = splits;
    return 0;
}
// C[K1][N] = A^T B with K1 % 128 == 0: one launch, (K1/128) x (N/128) tiles x splits blocks; slabs of K1 * N floats each (no bias part).
// The split count aims at g_tn_tile_blocks workgroups per launch (384: fewer slabs to write and combine than the 768 that fill every CU three times) within the slab buffer's capacity.
int launch_gemm_tn_sb_tiles(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, float* slab, int64_t slab_cap, int* nslab,
                            int M, int K1, int N, int conv_C, int conv_H, int conv_W) {
    int cvs = 0;
    if (conv_C) {      // A = NHWC image, K1 = 9 C, C a power of two and a multiple of 128
        while ((1 << cvs) < conv_C) ++cvs;
        if ((1 << cvs) != conv_C || (conv_C % 128) || K1 != 9 * conv_C || conv_H <= 0 || conv_W <= 0 || M % (conv_H * conv_W)) return -1;
        lda = conv_C;
    }
    if (M <= 0 || K1 <= 0 || (K1 % 128) || !gemm_tn_sb_usable(A, lda, Bm, ldb, 128, N)) return -1;
    const int tiles = (K1 / 128) * (N / 128);
    int64_t splits = (g_tn_tile_blocks + tiles - 1) / tiles;
    splits = std::min<int64_t>(splits, slab_cap / ((int64_t)K1 * N));
    splits = std::min<int64_t>(splits, 128);
    splits = std::min<int64_t>(splits, (M + 127) / 128);         // at least four 32-row chunks per split
    if (splits < 1) return -1;
    int rps = (int)((M + splits - 1) / splits);
    rps = (rps + 31) / 32 * 32;
    splits = (M + rps - 1) / rps;
    TnJobs jobs = {};
    jobs.A[0] = A; jobs.B[0] = Bm; jobs.lda[0] = lda; jobs.shift[0] = 0;
    const bool four = g_bwd_four && !g_mfma_one;      // a kernel gradient: backward only
    if (cvs && four) hipLaunchKernelGGL((gemm_tn_sb_kernel<true, false, true>), dim3(N / 128, K1 / 128, (unsigned)splits), dim3(256), 0, st, jobs, ldb, slab, M, N, rps, M, 0,
                                (long long)K1 * N, cvs, conv_H, conv_W);
    else if (cvs) hipLaunchKernelGGL(gemm_tn_sb_kernel<true>, dim3(N / 128, K1 / 128, (unsigned)splits), dim3(256), 0, st, jobs, ldb, slab, M, N, rps, M, 0,
                                (long long)K1 * N, cvs, conv_H, conv_W);
    else if (four) hipLaunchKernelGGL((gemm_tn_sb_kernel<false, false, true>), dim3(N / 128, K1 / 128, (unsigned)splits), dim3(256), 0, st, jobs, ldb, slab, M, N, rps, M, 0,
                            (long long)K1 * N, 0, 0, 0);
    else hipLaunchKernelGGL(gemm_tn_sb_kernel<false>, dim3(N / 128, K1 / 128, (unsigned)splits), dim3(256), 0, st, jobs, ldb, slab, M, N, rps, M, 0,
                            (long long)K1 * N, 0, 0, 0);
    *nslab = (int)splits;
    return 0;
}
int launch_gemm_tn_sb(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, float* slab, int* nslab, int M, int N, int S,
                      int shift, int want_bias) {
    TnJobs jobs = {};
    jobs.A[0] = A; jobs.B[0] = Bm; jobs.lda[0] = lda; jobs.shift[0] = shift;
    return launch_gemm_tn_sb_batch(st, jobs, 1, ldb, slab, nslab, M, N, S, want_bias);
}
