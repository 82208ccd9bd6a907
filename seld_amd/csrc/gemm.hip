// gemm.hip — fp32 MFMA GEMMs for the Dense / Conv1D(k=1) heads and the GRU input projections
// (modules.py:368-371, models.py:28-30, GRU kernel/recurrent_kernel products), forward and backward.
//   gemm_f32_kernel<TRANSB>   C[M,N] = act(A[M,K] * op(B) + bias) (+C)        64x64 tile, K-chunks of 32
//   gemm_tn_kernel            slab[z][K1,N] = sum_{m in split z} A[m,K1]^T B[m,N]   (weight gradients)
//   colsum_kernel             slab[z][N]    = sum_{m in split z} X[m,N]             (bias gradients)
//   reduce_slabs_kernel       out[i] (+)= sum_z slab[z][i]                          (deterministic combine)
// v_mfma_f32_32x32x2_f32 is exact fp32, so these meet the 1e-4 parity bar without a split scheme.
#include "common.h"
#include "prep.h"

#define GT_LDA 65  // transposed A tile: [k][row], +1 pad (transposing b32 writes are <=2-way conflicted)

// Software pipeline per K-chunk of 32: [commit the prefetched chunk to LDS] [LDS barrier] [issue the next
// chunk's global loads -> registers] [16 MFMAs on two accumulator chains] [LDS barrier].  Barriers are
// LDS-only so the prefetch stays in flight across them.
template <int TRANSB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                       const float* __restrict__ Bm, int ldb,
                                                       const float* __restrict__ bias, float* __restrict__ C,
                                                       int ldc, int M, int N, int K, int act, int accumulate,
                                                       const float* __restrict__ A1, const float* __restrict__ B1,
                                                       const float* __restrict__ bias1, float* __restrict__ C1, int mode,
                                                       float* __restrict__ M0, float* __restrict__ M1, int nsplit, GemmEpi epi) {
    // mode 0: C = act(A B + bias) (+C).   mode 1 (two products sharing A): column tiles >= ceil(N/64) compute
    // C1 = act(A B1 + bias1).   mode 2 (one product over a concatenated K): C = act(A B + A1 B1 + bias) (+C), K % 32 == 0.
    // mode 3: mode 0, and the result is also stored to C1 (same leading dimension): the caller's copy of a head output.
    // mode 4 (both heads' outputs from one product, launch_gemm_heads): columns < nsplit get activation act & 15 and go to
    // C [M][nsplit] (and M0), the others activation act >> 4 and go to C1 [M][N - nsplit] (and M1).
    const int ntx = (N + 63) >> 6;
    const bool second = mode == 1 && (int)blockIdx.x >= ntx;
    if (second) { Bm = B1; bias = bias1; C = C1; }
    constexpr int LDB = TRANSB ? 65 : 64;
    __shared__ __attribute__((aligned(16))) float As[32 * GT_LDA];
    __shared__ __attribute__((aligned(16))) float Bs[32 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = (second ? (int)blockIdx.x - ntx : (int)blockIdx.x) * 64;
    const bool a_vec = ((lda & 3) == 0) && ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_vec = ((ldb & 3) == 0) && (TRANSB ? ((K & 3) == 0) : ((N & 3) == 0)) &&
                       ((reinterpret_cast<uintptr_t>(Bm) & 15) == 0);
    f32x16 acc = zero16(), acc1 = zero16();   // two independent MFMA chains
    float va[2][4], vb[2][4];                  // prefetched chunk (indices are compile-time after unrolling)
    // A element (row = idx>>3, k4 = (idx&7)*4); B element: TRANSB ? (n = idx>>3, k4) : (kk = idx>>4, n4 = (idx&15)*4)
#define GEMM_LOAD(k0_)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                      \
        const int idx = tid + 256 * u;                                                                   \
        {                                                                                                \
            const int row = idx >> 3, k4 = (idx & 7) * 4;                                                \
            const int gm = m0 + row, gk = (k0_) + k4;                                                    \
            va[u][0] = va[u][1] = va[u][2] = va[u][3] = 0.f;                                             \
            if (gm < M) {                                                                                \
                const float* p = Ap + (size_t)gm * lda + gk;                                              \
                if (a_vec && gk + 3 < K) {                                                               \
                    const float4 t = *reinterpret_cast<const float4*>(p);                                \
                    va[u][0] = t.x; va[u][1] = t.y; va[u][2] = t.z; va[u][3] = t.w;                      \
                } else {                                                                                 \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) if (gk + j < K) va[u][j] = p[j];       \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        vb[u][0] = vb[u][1] = vb[u][2] = vb[u][3] = 0.f;                                                 \
        if (TRANSB == 0) {                                                                               \
            const int kk = idx >> 4, n4 = (idx & 15) * 4;                                                \
            const int gk = (k0_) + kk, gn = n0 + n4;                                                     \
            if (gk < K) {                                                                                \
                const float* p = Bp + (size_t)gk * ldb + gn;                                             \
                if (b_vec && gn + 3 < N) {                                                               \
                    const float4 t = *reinterpret_cast<const float4*>(p);                                \
                    vb[u][0] = t.x; vb[u][1] = t.y; vb[u][2] = t.z; vb[u][3] = t.w;                      \
                } else {                                                                                 \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) if (gn + j < N) vb[u][j] = p[j];       \
                }                                                                                        \
            }                                                                                            \
        } else {                                                                                         \
            const int n = idx >> 3, k4 = (idx & 7) * 4;                                                  \
            const int gn = n0 + n, gk = (k0_) + k4;                                                      \
            if (gn < N) {                                                                                \
                const float* p = Bp + (size_t)gn * ldb + gk;                                             \
                if (b_vec && gk + 3 < K) {                                                               \
                    const float4 t = *reinterpret_cast<const float4*>(p);                                \
                    vb[u][0] = t.x; vb[u][1] = t.y; vb[u][2] = t.z; vb[u][3] = t.w;                      \
                } else {                                                                                 \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) if (gk + j < K) vb[u][j] = p[j];       \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
    }
#define GEMM_COMMIT()                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                      \
        const int idx = tid + 256 * u;                                                                   \
        {                                                                                                \
            const int row = idx >> 3, k4 = (idx & 7) * 4;                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) As[(k4 + j) * GT_LDA + row] = va[u][j];        \
        }                                                                                                \
        if (TRANSB == 0) {                                                                               \
            const int kk = idx >> 4, n4 = (idx & 15) * 4;                                                \
            *reinterpret_cast<float4*>(&Bs[kk * LDB + n4]) = make_float4(vb[u][0], vb[u][1], vb[u][2], vb[u][3]); \
        } else {                                                                                         \
            const int n = idx >> 3, k4 = (idx & 7) * 4;                                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) Bs[(k4 + j) * LDB + n] = vb[u][j];             \
        }                                                                                                \
    }
    // chunk g of the (possibly concatenated) K axis: operand pair g / nck, k offset (g % nck) * 32
    const int nck = (K + 31) >> 5, ng = mode == 2 ? 2 * nck : nck;
#define GEMM_LOADG(g_)                                                                                   \
    {                                                                                                    \
        const int h_ = (g_) >= nck ? 1 : 0;                                                              \
        const float* Ap = h_ ? A1 : A;                                                                   \
        const float* Bp = h_ ? B1 : Bm;                                                                  \
        GEMM_LOAD(((g_) - h_ * nck) * 32)                                                                \
    }
    GEMM_LOADG(0)
    for (int g = 0; g < ng; ++g) {
        GEMM_COMMIT()
        lds_barrier();
        if (g + 1 < ng) GEMM_LOADG(g + 1)
        {
            // 4-deep LDS operand ring pinned with sched_barrier (see conv.hip)
            float ra[4], rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ra[u] = As[(2 * u + hi) * GT_LDA + wr * 32 + li];
                rb[u] = Bs[(2 * u + hi) * LDB + wc * 32 + li];
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int u = s & 3;
                __builtin_amdgcn_sched_barrier(0);
                if (s & 1) acc1 = MFMA_F32_32x32x2(ra[u], rb[u], acc1);
                else acc = MFMA_F32_32x32x2(ra[u], rb[u], acc);
                if (s + 4 < 16) {
                    ra[u] = As[(2 * (s + 4) + hi) * GT_LDA + wr * 32 + li];
                    rb[u] = Bs[(2 * (s + 4) + hi) * LDB + wc * 32 + li];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
    }
#undef GEMM_LOADG
#undef GEMM_LOAD
#undef GEMM_COMMIT
    const int col = n0 + wc * 32 + li;
    if (epi.stat_part) {
        // BatchNorm statistics of this 64 x 64 tile (common.h GemmEpi; the launcher admits mode 0 without bias / activation / accumulate only):
        // a lane holds 16 rows of one column, the two row halves meet through a lane swap, the two row waves through LDS (As is free behind the
        // last chunk's barrier), fixed order; one [sum | sum of squares] partial per (row block, column)
        float s_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = (m0 + wr * 32 + mfma_row(r, hi) < M) ? acc[r] + acc1[r] : 0.f;
            s_ += v; q_ = fmaf(v, v, q_);
        }
        s_ += __shfl_xor(s_, 32); q_ += __shfl_xor(q_, 32);
        if (hi == 0) { As[(wr * 64 + wc * 32 + li) * 2] = s_; As[(wr * 64 + wc * 32 + li) * 2 + 1] = q_; }
        lds_barrier();
        if (tid < 64) {      // (columns past N: zeros, so that a 32-channel product leaves a clean 64-channel partial)
            const bool in_ = n0 + tid < N;
            float* pp = epi.stat_part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 128 + tid;
            pp[0] = in_ ? As[tid * 2] + As[(64 + tid) * 2] : 0.f;
            pp[64] = in_ ? As[tid * 2 + 1] + As[(64 + tid) * 2 + 1] : 0.f;
        }
    }
    if ((mode == 0 || mode == 3) && m0 + wr * 32 + 32 <= M && n0 + wc * 32 + 32 <= N && (ldc & 3) == 0 &&
        (reinterpret_cast<uintptr_t>(C) & 15) == 0 && (mode == 0 || (reinterpret_cast<uintptr_t>(C1) & 15) == 0)) {
        // whole 32 x 32 tile: 4 dwordx4 stores per wave instead of 16 dword stores (common.h: quad_transpose4)
        const float bv = bias ? bias[col] : 0.f;
        const size_t o0 = (size_t)(m0 + wr * 32 + 4 * hi + (li & 3)) * ldc + n0 + wc * 32 + (li & ~3);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v_[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v_[j] = (acc[4 * q + j] + acc1[4 * q + j]) + bv;
                if (act == 1) v_[j] = 1.f / (1.f + expf(-v_[j]));
                else if (act == 2) v_[j] = tanhf(v_[j]);
                else if (act == 3) v_[j] = fmaxf(v_[j], 0.f);
            }
            float4 o = quad_transpose4(v_[0], v_[1], v_[2], v_[3], li);
            float4* p = reinterpret_cast<float4*>(C + o0 + (size_t)(8 * q) * ldc);
            if (accumulate) { const float4 c0 = *p; o = make_float4(o.x + c0.x, o.y + c0.y, o.z + c0.z, o.w + c0.w); }
            *p = o;
            if (mode == 3) *reinterpret_cast<float4*>(C1 + o0 + (size_t)(8 * q) * ldc) = o;
        }
        return;
    }
    if (col < N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wr * 32 + mfma_row(r, hi);
            if (row < M) {
                float v = (acc[r] + acc1[r]) + bv;
                if (mode == 4) {
                    const bool first = col < nsplit;
                    const int a_ = first ? (act & 15) : (act >> 4);
                    if (a_ == 1) v = 1.f / (1.f + expf(-v));
                    else if (a_ == 2) v = tanhf(v);
                    const size_t o_ = first ? (size_t)row * nsplit + col : (size_t)row * (N - nsplit) + (col - nsplit);
                    (first ? C : C1)[o_] = v;
                    float* m_ = first ? M0 : M1;
                    if (m_) m_[o_] = v;
                    continue;
                }
                if (act == 1) v = 1.f / (1.f + expf(-v));
                else if (act == 2) v = tanhf(v);
                else if (act == 3) v = fmaxf(v, 0.f);
                float* p = C + (size_t)row * ldc + col;
                if (accumulate) v += *p;
                *p = v;
                if (mode == 3) C1[(size_t)row * ldc + col] = v;
            }
        }
    }
}

static int gemm_go(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, const float* bias, float* C, int ldc, int M,
                   int N, int K, int transb, int act, int accumulate, const float* A1, const float* B1, const float* bias1, float* C1,
                   int mode, float* M0 = nullptr, float* M1 = nullptr, int nsplit = 0) {
    if (M <= 0 || N <= 0 || K <= 0) return -1;
    if (mode == 2 && (K & 31)) return -2;
    const GemmEpi epi = g_gemm_epi;
    if (epi.stat_part && (bias || act || accumulate || mode)) return -3;
    if (epi.addg) return -3;      // the gated add lives in the split-bf16 kernels' epilogues only
    dim3 grid((N + 63) / 64 * (mode == 1 ? 2 : 1), (M + 63) / 64);
    if (transb)
        hipLaunchKernelGGL(gemm_f32_kernel<1>, grid, dim3(256), 0, st, A, lda, Bm, ldb, bias, C, ldc, M, N, K, act, accumulate, A1, B1,
                           bias1, C1, mode, M0, M1, nsplit, epi);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<0>, grid, dim3(256), 0, st, A, lda, Bm, ldb, bias, C, ldc, M, N, K, act, accumulate, A1, B1,
                           bias1, C1, mode, M0, M1, nsplit, epi);
    return 0;
}

int launch_gemm(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, const float* bias, float* C,
                int ldc, int M, int N, int K, int transb, int act, int accumulate) {
    return gemm_go(st, A, lda, Bm, ldb, bias, C, ldc, M, N, K, transb, act, accumulate, nullptr, nullptr, nullptr, nullptr, 0);
}

// C = act(A B + bias), also stored to `mirror` (same ldc) when it is not null
int launch_gemm_mirror(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, const float* bias, float* C, float* mirror,
                       int ldc, int M, int N, int K, int transb, int act) {
    return gemm_go(st, A, lda, Bm, ldb, bias, C, ldc, M, N, K, transb, act, 0, nullptr, nullptr, nullptr, mirror, mirror ? 3 : 0);
}

// both heads' outputs in one product: [C0 | C1] = act0 / act1 (A [W0 | W1] + [b0 | b1]) with the weights already concatenated
// (Bm [K][n0 + n1], bias [n0 + n1]); C0 [M][n0], C1 [M][n1] and optional copies M0 / M1
int launch_gemm_heads(hipStream_t st, const float* A, int lda, const float* Bm, const float* bias, float* C0, float* C1, float* M0,
                      float* M1, int M, int n0, int n1, int K, int act0, int act1) {
    return gemm_go(st, A, lda, Bm, n0 + n1, bias, C0, 0, M, n0 + n1, K, 0, act0 | (act1 << 4), 0, nullptr, nullptr, nullptr, C1, 4, M0, M1, n0);
}

// ---- two linear layers in a row (the heads: Conv1D(128, 1) then Dense(n), no activation between): y2 = act(x Weff + beff) with
// Weff = W1 W2, beff = b1 W2 + b2, and every gradient follows from F = x^T dy2 and cs = colsum(dy2) (api.hip, heads_fused):
//   dW2 = W1^T F + b1 (x) cs,  db2 = cs,  dW1 = F W2^T,  db1 = cs W2^T,  dx = dy2 Weff^T.
// weff_all [K + 1][NT]: rows 0..K-1 = [W1s W2s | W1d W2d], row K = [b1s W2s + b2s | b1d W2d + b2d]   (NT = n0 + n1)
__global__ __launch_bounds__(64) void heads_weff_kernel(HeadsLin h, float* __restrict__ weff) { heads_weff_body(h, weff, blockIdx.x, threadIdx.x); }
int launch_heads_weff(hipStream_t st, const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2,
                      const int* n, int K, int Hd, float* weff) {
    if (n[0] + n[1] > 64) return -1;
    HeadsLin h;
    for (int i = 0; i < 2; ++i) { h.w1[i] = w1[i]; h.b1[i] = b1[i]; h.w2[i] = w2[i]; h.b2[i] = b2[i]; h.n[i] = n[i]; }
    h.K = K; h.Hd = Hd;
    hipLaunchKernelGGL(heads_weff_kernel, dim3(K + 1), dim3(64), 0, st, h, weff);
    return 0;
}

// gradients of both heads' four tensors from F [K][NT] and cs [NT] (double accumulation, one thread per output element)
struct HeadsGrad { const float *w1[2], *b1[2], *w2[2]; float *dw1[2], *db1[2], *dw2[2], *db2[2]; int n[2]; int K, Hd; };
__global__ __launch_bounds__(256) void heads_grad_kernel(HeadsGrad h, const float* __restrict__ F, const float* __restrict__ cs) {
    const int NT = h.n[0] + h.n[1];
    const int hd = blockIdx.y, N = h.n[hd], c0 = hd ? h.n[0] : 0;
    const int nW1 = h.K * h.Hd, nW2 = h.Hd * N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nW1) {                                   // dW1[k][j] = sum_n F[k][c0 + n] W2[j][n]
        const int k = i / h.Hd, j = i - k * h.Hd;
        double s0 = 0.0, s1 = 0.0;
        int n = 0;
        for (; n + 1 < N; n += 2) {
            s0 += (double)F[(size_t)k * NT + c0 + n] * (double)h.w2[hd][(size_t)j * N + n];
            s1 += (double)F[(size_t)k * NT + c0 + n + 1] * (double)h.w2[hd][(size_t)j * N + n + 1];
        }
        if (n < N) s0 += (double)F[(size_t)k * NT + c0 + n] * (double)h.w2[hd][(size_t)j * N + n];
        h.dw1[hd][i] = (float)(s0 + s1);
    } else if (i < nW1 + nW2) {                      // dW2[j][n] = sum_k W1[k][j] F[k][c0 + n] + b1[j] cs[c0 + n]
        const int e = i - nW1, j = e / N, n = e - j * N;
        double s0 = (double)h.b1[hd][j] * (double)cs[c0 + n], s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = 0;
#pragma unroll 2
        for (; k + 3 < h.K; k += 4) {
            s0 += (double)h.w1[hd][(size_t)k * h.Hd + j] * (double)F[(size_t)k * NT + c0 + n];
            s1 += (double)h.w1[hd][(size_t)(k + 1) * h.Hd + j] * (double)F[(size_t)(k + 1) * NT + c0 + n];
            s2 += (double)h.w1[hd][(size_t)(k + 2) * h.Hd + j] * (double)F[(size_t)(k + 2) * NT + c0 + n];
            s3 += (double)h.w1[hd][(size_t)(k + 3) * h.Hd + j] * (double)F[(size_t)(k + 3) * NT + c0 + n];
        }
        for (; k < h.K; ++k) s0 += (double)h.w1[hd][(size_t)k * h.Hd + j] * (double)F[(size_t)k * NT + c0 + n];
        h.dw2[hd][e] = (float)((s0 + s1) + (s2 + s3));
    } else if (i < nW1 + nW2 + h.Hd) {               // db1[j] = sum_n cs[c0 + n] W2[j][n]
        const int j = i - nW1 - nW2;
        double s = 0.0;
        for (int n = 0; n < N; ++n) s += (double)cs[c0 + n] * (double)h.w2[hd][(size_t)j * N + n];
        h.db1[hd][j] = (float)s;
    } else if (i < nW1 + nW2 + h.Hd + N) {           // db2 = cs
        const int n = i - nW1 - nW2 - h.Hd;
        h.db2[hd][n] = cs[c0 + n];
    }
}
int launch_heads_grad(hipStream_t st, const float* const* w1, const float* const* b1, const float* const* w2, float* const* dw1,
                      float* const* db1, float* const* dw2, float* const* db2, const int* n, int K, int Hd, const float* F,
                      const float* cs) {
    HeadsGrad h;
    for (int i = 0; i < 2; ++i) {
        h.w1[i] = w1[i]; h.b1[i] = b1[i]; h.w2[i] = w2[i]; h.dw1[i] = dw1[i]; h.db1[i] = db1[i]; h.dw2[i] = dw2[i]; h.db2[i] = db2[i];
        h.n[i] = n[i];
    }
    h.K = K; h.Hd = Hd;
    const int nmax = K * Hd + Hd * (n[0] > n[1] ? n[0] : n[1]) + Hd + 64;
    hipLaunchKernelGGL(heads_grad_kernel, dim3((nmax + 255) / 256, 2), dim3(256), 0, st, h, F, cs);
    return 0;
}

// two products that share A in one launch: C0 = act(A B0 + bias0), C1 = act(A B1 + bias1) (same shapes and leading dimensions)
int launch_gemm_dual_n(hipStream_t st, const float* A, int lda, const float* B0, const float* B1, int ldb, const float* bias0,
                       const float* bias1, float* C0, float* C1, int ldc, int M, int N, int K, int transb, int act) {
    return gemm_go(st, A, lda, B0, ldb, bias0, C0, ldc, M, N, K, transb, act, 0, nullptr, B1, bias1, C1, 1);
}

// one product over a concatenated K axis: C = act(A0 B0 + A1 B1 + bias) (+C); K % 32 == 0
int launch_gemm_dual_k(hipStream_t st, const float* A0, const float* A1, int lda, const float* B0, const float* B1, int ldb,
                       const float* bias, float* C, int ldc, int M, int N, int K, int transb, int act, int accumulate) {
    return gemm_go(st, A0, lda, B0, ldb, bias, C, ldc, M, N, K, transb, act, accumulate, A1, B1, nullptr, nullptr, 2);
}

// ------------------------------------------------------------------------------------------------
// TN: out[k1][n] = sum_m A[m][k1] * B[m][n].  A-operand[i=k1][kk=m], B-operand[kk=m][j=n]: both tiles
// are read with the lane index on the contiguous axis, so the LDS images are plain row-major copies.
#define TN_MAX_SPLITS 128
int gemm_tn_max_splits() { return TN_MAX_SPLITS; }

// CR = rows per chunk: 32, or 64 for the long skinny products of resnet50_block's first stages (M = 10^5 rows, a 64 x 64 output): the
// loop prefetches one chunk ahead, so a chunk costs a load round trip, and twice the rows per trip halve the trips
template <int CR>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int lda,
                                                      const float* __restrict__ Bm, int ldb,
                                                      float* __restrict__ slab, int M, int K1, int N,
                                                      int rows_per_split, int S, int shift, int want_bias) {
    __shared__ __attribute__((aligned(16))) float As[CR * 64];
    __shared__ __attribute__((aligned(16))) float Bs[CR * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, li = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * 64, k10 = blockIdx.y * 64;
    const int mbeg = blockIdx.z * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    const bool a_vec = ((lda & 3) == 0) && ((K1 & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_vec = ((ldb & 3) == 0) && ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(Bm) & 15) == 0);
    f32x16 acc = zero16(), acc1 = zero16();
    float bsum = 0.f;  // tid < 64 of the k1-tile-0 blocks: column sum of B (bias gradient)
    const bool do_bias = want_bias && blockIdx.y == 0 && tid < 64;
    constexpr int NU = CR / 16;       // float4 per thread and operand per chunk
    float va[NU][4], vb[NU][4];
#define TN_LOAD(mm0_)                                                                                    \
    _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                     \
        const int idx = tid + 256 * u;                                                                   \
        const int r = idx >> 4, c4 = (idx & 15) * 4;                                                     \
        const int gm = (mm0_) + r;                                                                       \
        va[u][0] = va[u][1] = va[u][2] = va[u][3] = 0.f;                                                 \
        vb[u][0] = vb[u][1] = vb[u][2] = vb[u][3] = 0.f;                                                 \
        if (gm < mend) {                                                                                 \
            int am = gm;                                                                                 \
            bool ok = true;                                                                              \
            if (shift != 0) {   /* A row with the time shift (H_prev for the recurrent-kernel gradient) */ \
                const int t = gm % S;                                                                    \
                ok = (t + shift >= 0) && (t + shift < S);                                                \
                am = gm + shift;                                                                         \
            }                                                                                            \
            if (ok) {                                                                                    \
                const float* p = A + (size_t)am * lda + k10 + c4;                                        \
                if (a_vec && k10 + c4 + 3 < K1) {                                                        \
                    const float4 t4 = *reinterpret_cast<const float4*>(p);                               \
                    va[u][0] = t4.x; va[u][1] = t4.y; va[u][2] = t4.z; va[u][3] = t4.w;                  \
                } else {                                                                                 \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) if (k10 + c4 + j < K1) va[u][j] = p[j]; \
                }                                                                                        \
            }                                                                                            \
            const float* q = Bm + (size_t)gm * ldb + n0 + c4;                                            \
            if (b_vec && n0 + c4 + 3 < N) {                                                              \
                const float4 t4 = *reinterpret_cast<const float4*>(q);                                   \
                vb[u][0] = t4.x; vb[u][1] = t4.y; vb[u][2] = t4.z; vb[u][3] = t4.w;                      \
            } else {                                                                                     \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) if (n0 + c4 + j < N) vb[u][j] = q[j];      \
            }                                                                                            \
        }                                                                                                \
    }
#define TN_COMMIT()                                                                                      \
    _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                                      \
        const int idx = tid + 256 * u;                                                                   \
        const int r = idx >> 4, c4 = (idx & 15) * 4;                                                     \
        *reinterpret_cast<float4*>(&As[r * 64 + c4]) = make_float4(va[u][0], va[u][1], va[u][2], va[u][3]); \
        *reinterpret_cast<float4*>(&Bs[r * 64 + c4]) = make_float4(vb[u][0], vb[u][1], vb[u][2], vb[u][3]); \
    }
    TN_LOAD(mbeg)
    for (int mm0 = mbeg; mm0 < mend; mm0 += CR) {
        TN_COMMIT()
        lds_barrier();
        if (mm0 + CR < mend) TN_LOAD(mm0 + CR)
        if (do_bias) {
#pragma unroll
            for (int r = 0; r < CR; ++r) bsum += Bs[r * 64 + tid];
        }
        {
            float ra[4], rb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ra[u] = As[(2 * u + hi) * 64 + wr * 32 + li];
                rb[u] = Bs[(2 * u + hi) * 64 + wc * 32 + li];
            }
#pragma unroll
            for (int s = 0; s < CR / 2; ++s) {
                const int u = s & 3;
                __builtin_amdgcn_sched_barrier(0);
                if (s & 1) acc1 = MFMA_F32_32x32x2(ra[u], rb[u], acc1);
                else acc = MFMA_F32_32x32x2(ra[u], rb[u], acc);
                if (s + 4 < CR / 2) {
                    ra[u] = As[(2 * (s + 4) + hi) * 64 + wr * 32 + li];
                    rb[u] = Bs[(2 * (s + 4) + hi) * 64 + wc * 32 + li];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();
    }
#undef TN_LOAD
#undef TN_COMMIT
    float* out = slab + (size_t)blockIdx.z * ((size_t)K1 * N + N);
    if (do_bias && n0 + tid < N) out[(size_t)K1 * N + n0 + tid] = bsum;
    const int col = n0 + wc * 32 + li;
    if (k10 + wr * 32 + 32 <= K1 && n0 + wc * 32 + 32 <= N && (N & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        // whole tile: 4 dwordx4 stores (common.h: quad_transpose4)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(out + (size_t)(k10 + wr * 32 + 8 * q + 4 * hi + (li & 3)) * N + n0 + wc * 32 + (li & ~3)) =
                quad_transpose4(acc[4 * q] + acc1[4 * q], acc[4 * q + 1] + acc1[4 * q + 1], acc[4 * q + 2] + acc1[4 * q + 2],
                                acc[4 * q + 3] + acc1[4 * q + 3], li);
        return;
    }
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = k10 + wr * 32 + mfma_row(r, hi);
            if (row < K1) out[(size_t)row * N + col] = acc[r] + acc1[r];
        }
    }
}

int launch_gemm_tn(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, float* slab, int* nslab,
                   int M, int K1, int N, int S, int shift, int want_bias, int max_splits, int chunk_rows) {
    if (M <= 0 || K1 <= 0 || N <= 0) return -1;
    if (max_splits <= 0) max_splits = TN_MAX_SPLITS;       // callers with small K1 x N slabs may ask for more (the slab buffer permitting)
    // Rows per split: the loop prefetches one 32-row chunk ahead, i.e. every chunk costs a load round trip (~2 us) unless other
    // blocks of the CU cover it: short splits = many co-resident blocks (16 KB of LDS each) and few dependent chunks per block
    int splits = (M + 159) / 160;
    if (splits > max_splits) splits = max_splits;
    int rps = (M + splits - 1) / splits;
    const int cr = chunk_rows == 64 ? 64 : 32;
    rps = (rps + cr - 1) / cr * cr;
    splits = (M + rps - 1) / rps;
    dim3 grid((N + 63) / 64, (K1 + 63) / 64, splits);
    if (cr == 64) hipLaunchKernelGGL(gemm_tn_kernel<64>, grid, dim3(256), 0, st, A, lda, Bm, ldb, slab, M, K1, N, rps, S > 0 ? S : M, shift, want_bias);
    else hipLaunchKernelGGL(gemm_tn_kernel<32>, grid, dim3(256), 0, st, A, lda, Bm, ldb, slab, M, K1, N, rps, S > 0 ? S : M, shift, want_bias);
    *nslab = splits;
    return 0;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int ld, float* __restrict__ slab,
                                                     int M, int N, int rows_per_split) {
    const int mbeg = blockIdx.x * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.f;
        for (int m = mbeg; m < mend; ++m) s += X[(size_t)m * ld + n];
        slab[(size_t)blockIdx.x * N + n] = s;
    }
}

int launch_colsum(hipStream_t st, const float* X, int ld, float* slab, int* nslab, int M, int N) {
    int splits = (M + 255) / 256;
    if (splits > 256) splits = 256;
    const int rps = (M + splits - 1) / splits;
    splits = (M + rps - 1) / rps;
    hipLaunchKernelGGL(colsum_kernel, dim3(splits), dim3(256), 0, st, X, ld, slab, M, N, rps);
    *nslab = splits;
    return 0;
}

// this thread's share of out[i] = sum_z slab[z][i]: slabs g, g + 8, ... with 4 independent loads in flight
__device__ __forceinline__ double slab_group_sum(const float* __restrict__ slab, int nslab, int64_t stride, int64_t i, int g) {
    double s = 0.0;
    int z = g;
    for (; z + 24 < nslab; z += 32) {
        const float v0 = slab[(size_t)z * stride + i], v1 = slab[(size_t)(z + 8) * stride + i];
        const float v2 = slab[(size_t)(z + 16) * stride + i], v3 = slab[(size_t)(z + 24) * stride + i];
        s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; z < nslab; z += 8) s += (double)slab[(size_t)z * stride + i];
    return s;
}
// out[i] (+)= sum_z slab[z][i].  Block = 32 outputs x 8 slab groups: thread (o = tid & 31, g = tid >> 5) adds
// slabs g, g+8, ... (independent, coalesced 128-B rows), the 8 group sums are combined through LDS in a
// fixed order -> bit-reproducible and 8x the memory parallelism of one thread per output.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, int nslab,
                                                           int64_t stride, float* __restrict__ out, int64_t n,
                                                           int accumulate) {
    __shared__ double red[256];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + o;
    const double s = i < n ? slab_group_sum(slab, nslab, stride, i, g) : 0.0;
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + o];
        float v = (float)t;
        if (accumulate) v += out[i];
        out[i] = v;
    }
}

// slab[z] = [n_w main | n_b tail]: out_w[i] = sum_z main, out_b[i] = sum_z tail (same block scheme)
__global__ __launch_bounds__(256) void reduce_slabs2_kernel(const float* __restrict__ slab, int nslab, int64_t stride,
                                                            float* __restrict__ out_w, int64_t n_w,
                                                            float* __restrict__ out_b, int64_t n_b) {
    __shared__ double red[256];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + o;
    const int64_t n = n_w + n_b;
    const double s = i < n ? slab_group_sum(slab, nslab, stride, i, g) : 0.0;
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + o];
        if (i < n_w) out_w[i] = (float)t;
        else out_b[i - n_w] = (float)t;
    }
}

// the same for the jobs of launch_gemm_tn_sb_batch: blockIdx.y = job
__global__ __launch_bounds__(256) void reduce_slabs2_batch_kernel(const float* __restrict__ slab, int nslab, int64_t stride, TnJobs jobs,
                                                                  int64_t n_w, int64_t n_b) {
    __shared__ double red[256];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + o;
    const int64_t n = n_w + n_b;
    const float* src = slab + (size_t)blockIdx.y * nslab * stride;
    const double s = i < n ? slab_group_sum(src, nslab, stride, i, g) : 0.0;
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + o];
        if (i < n_w) jobs.out_w[blockIdx.y][i] = (float)t;
        else jobs.out_b[blockIdx.y][i - n_w] = (float)t;
    }
}
int launch_reduce_slabs2_batch(hipStream_t st, const float* slab, int nslab, int64_t stride, const TnJobs& jobs, int njobs, int64_t n_w, int64_t n_b) {
    const int64_t n = n_w + n_b;
    hipLaunchKernelGGL(reduce_slabs2_batch_kernel, dim3((unsigned)((n + 31) / 32), njobs), dim3(256), 0, st, slab, nslab, stride, jobs, n_w, n_b);
    return 0;
}

int launch_reduce_slabs2(hipStream_t st, const float* slab, int nslab, int64_t stride, float* out_w, int64_t n_w,
                         float* out_b, int64_t n_b) {
    const int64_t n = n_w + n_b;
    hipLaunchKernelGGL(reduce_slabs2_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, slab, nslab, stride, out_w,
                       n_w, out_b, n_b);
    return 0;
}

int launch_reduce_slabs(hipStream_t st, const float* slab, int nslab, int64_t slab_stride, float* out,
                        int64_t n, int accumulate) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, slab, nslab,
                       slab_stride, out, n, accumulate);
    return 0;
}

// Many slabs (thousands: one per workgroup of a streaming kernel): a single launch of reduce_slabs_kernel is n / 32 workgroups each walking ALL slabs
// (576 outputs x 4 800 slabs: 18 workgroups, ~100 us).  Two stages instead: blockIdx.y = a group of `per` slabs summed into tmp[y][n] (double inside,
// float out), then the groups by reduce_slabs_kernel.  Fixed order in both stages.
__global__ __launch_bounds__(256) void reduce_slabs_stage_kernel(const float* __restrict__ slab, int nslab, int per, int64_t stride,
                                                                 float* __restrict__ tmp, int64_t n) {
    __shared__ double red[256];
    const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + o;
    const int z0 = blockIdx.y * per, cnt = nslab - z0 < per ? nslab - z0 : per;
    const double s = i < n ? slab_group_sum(slab + (size_t)z0 * stride, cnt, stride, i, g) : 0.0;
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * 32 + o];
        tmp[(size_t)blockIdx.y * n + i] = (float)t;
    }
}
int reduce_slabs_groups(int nslab) { return (nslab + 63) / 64; }      // rows of `tmp` launch_reduce_slabs_2stage needs ([groups][n] floats)
int launch_reduce_slabs_2stage(hipStream_t st, const float* slab, int nslab, int64_t slab_stride, float* out, int64_t n, float* tmp) {
    if (n <= 0) return 0;
    const int groups = reduce_slabs_groups(nslab);
    hipLaunchKernelGGL(reduce_slabs_stage_kernel, dim3((unsigned)((n + 31) / 32), (unsigned)groups), dim3(256), 0, st, slab, nslab, 64, slab_stride, tmp, n);
    return launch_reduce_slabs(st, tmp, groups, n, out, n, 0);
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 x = reinterpret_cast<const float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(out)[i] = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
}

int launch_mul(hipStream_t st, const float* a, const float* b, float* out, int64_t n) {
    if (n & 3) return -1;
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(mul_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a, b, out, n4);
    return 0;
}
