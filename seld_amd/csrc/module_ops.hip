// module_ops.hip — the operators seld_amd/modules.py composes the reference's configurable 2-D blocks from (modules.mother_block /
// mother_stage, modules.py:15-43, 184-298): Conv2D(k, 'same', strides) as im2col + the fp32-MFMA GEMM of gemm.hip, training-mode
// BatchNormalization, activations, skip sums, channel concatenation and the squeeze-and-excitation tail.  ANY channel count / kernel /
// stride (the reference's own test shapes use 3, 6, 8, 11 channels: modules_test.py:8-28, 154-200): correctness-first kernels around the
// one dense product, which runs on the matrix cores; tuned forms exist for the shapes the BASELINE configurations use (conv_sb.hip,
// resnet.hip), not for these.  C ABI "seld_m_*": asynchronous on the caller's stream, no allocation, caller-provided scratch.
#include "common.h"
#include <cmath>
#include "../../include/seld_hip.h"
#include <math.h>

namespace {

// TensorFlow 'SAME': out = ceil(in / stride), pad_total = max((out - 1) * stride + k - in, 0), pad_before = pad_total / 2
__host__ __device__ inline int same_out(int in, int stride) { return (in + stride - 1) / stride; }
__host__ __device__ inline int same_pad_before(int in, int k, int stride) {
    const int out = (in + stride - 1) / stride;
    const int tot = (out - 1) * stride + k - in;
    return tot > 0 ? tot / 2 : 0;
}

// col[(b, ho, wo)][(ki, kj, c)] = x[b][ho sh + ki - ph][wo sw + kj - pw][c]  (0 outside)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int64_t total, int H, int W, int C,
                                                     int kh, int kw, int sh, int sw, int Ho, int Wo, int ph, int pw) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int K = kh * kw * C;
    const int64_t row = e / K;
    const int k = (int)(e - row * K);
    const int c = k % C, kj = (k / C) % kw, ki = k / (C * kw);
    const int wo = (int)(row % Wo), ho = (int)((row / Wo) % Ho);
    const int64_t b = row / ((int64_t)Wo * Ho);
    const int h = ho * sh + ki - ph, w = wo * sw + kj - pw;
    col[e] = (h >= 0 && h < H && w >= 0 && w < W) ? x[((b * H + h) * W + w) * C + c] : 0.f;
}

// dx[b][h][w][c] (+)= sum over (ki, kj) with (h + ph - ki) % sh == 0 ... of dcol[(b, ho, wo)][(ki, kj, c)]: gather form, fixed order
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int64_t total, int H, int W, int C,
                                                     int kh, int kw, int sh, int sw, int Ho, int Wo, int ph, int pw, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C), w = (int)((e / C) % W), h = (int)((e / ((int64_t)C * W)) % H);
    const int64_t b = e / ((int64_t)C * W * H);
    const int K = kh * kw * C;
    float s = 0.f;
    for (int ki = 0; ki < kh; ++ki) {
        const int hn = h + ph - ki;
        if (hn < 0 || hn % sh) continue;
        const int ho = hn / sh;
        if (ho >= Ho) continue;
        for (int kj = 0; kj < kw; ++kj) {
            const int wn = w + pw - kj;
            if (wn < 0 || wn % sw) continue;
            const int wo = wn / sw;
            if (wo >= Wo) continue;
            s += dcol[((b * Ho + ho) * Wo + wo) * K + (ki * kw + kj) * C + c];
        }
    }
    dx[e] = accumulate ? dx[e] + s : s;
}

// one workgroup per channel: sums over the pixels in double, fixed order (thread-strided partial sums, then a tree)
template <int NS>
__device__ __forceinline__ void block_reduce(double (&v)[NS], double* red) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NS; ++k) red[k * 256 + tid] = v[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int k = 0; k < NS; ++k) red[k * 256 + tid] += red[k * 256 + tid + s];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = red[k * 256];
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, int64_t npix, int C, float* __restrict__ mean,
                                                       float* __restrict__ var) {
    __shared__ double red[2 * 256];
    const int c = blockIdx.x;
    double v[2] = {0.0, 0.0};
    for (int64_t p = threadIdx.x; p < npix; p += 256) {
        const double t = z[p * C + c];
        v[0] += t;
        v[1] += t * t;
    }
    block_reduce<2>(v, red);
    if (threadIdx.x == 0) {
        const double m = v[0] / (double)npix;
        mean[c] = (float)m;
        var[c] = (float)fmax(v[1] / (double)npix - m * m, 0.0);      // biased batch variance (Keras normalises with it)
    }
}

// ---- the same sums as bn_stats_kernel / bn_bwd_sums_kernel in TWO stages that fill the card (round 5: one workgroup per channel walked a strided
// column of the tensor — 96 workgroups for a 96-channel, 162 MB tensor took 0.77 / 1.17 ms, 30 % of a mother_stage train step).  Stage 1: workgroup b
// takes a contiguous run of pixels; for C <= 256 its threads cover R = 256 / C pixels x C channels per trip (R C consecutive floats: coalesced), thread t
// always channel t % C; wider tensors give a thread the channels t, t + 256, ... (at most 8: C <= 2048).  Sums in double, the R rows of a channel
// combined through LDS in a fixed order, one [2][C] double partial per workgroup.  Stage 2: one thread per channel adds the partials in workgroup order.
#define BNP_MAX_BLOCKS 1024
#define BNP_MAX_SLOTS 8
template <bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ mean,
                                                         const float* __restrict__ var, float eps, int64_t npix, int C, double* __restrict__ part) {
    __shared__ double red[2 * 256];
    const int t = threadIdx.x;
    const int64_t per = (npix + gridDim.x - 1) / gridDim.x, p0 = (int64_t)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
    double* out = part + (size_t)blockIdx.x * 2 * C;
    if (C <= 256) {
        const int R = 256 / C, r = t / C, c = t - r * C;
        const bool on = r < R;
        double mu = 0.0, is = 0.0;
        if (BWD && on) { mu = mean[c]; is = 1.0 / sqrt((double)var[c] + (double)eps); }
        double a = 0.0, b = 0.0;
        if (on)
            for (int64_t p = p0 + r; p < p1; p += R) {
                const double v = z[p * C + c];
                if (BWD) { const double d = dy[p * C + c]; a += d; b += d * (v - mu) * is; }
                else { a += v; b += v * v; }
            }
        red[t] = a; red[256 + t] = b;
        __syncthreads();
        if (t < C) {
            double sa = 0.0, sb = 0.0;
            for (int k = 0; k < R; ++k) { sa += red[k * C + t]; sb += red[256 + k * C + t]; }
            out[t] = sa; out[C + t] = sb;
        }
        return;
    }
    double a[BNP_MAX_SLOTS], b[BNP_MAX_SLOTS], mu[BNP_MAX_SLOTS], is[BNP_MAX_SLOTS];
#pragma unroll
    for (int k = 0; k < BNP_MAX_SLOTS; ++k) {
        a[k] = b[k] = mu[k] = is[k] = 0.0;
        const int c = t + 256 * k;
        if (BWD && c < C) { mu[k] = mean[c]; is[k] = 1.0 / sqrt((double)var[c] + (double)eps); }
    }
    for (int64_t p = p0; p < p1; ++p)
#pragma unroll
        for (int k = 0; k < BNP_MAX_SLOTS; ++k) {
            const int c = t + 256 * k;
            if (c < C) {
                const double v = z[p * C + c];
                if (BWD) { const double d = dy[p * C + c]; a[k] += d; b[k] += d * (v - mu[k]) * is[k]; }
                else { a[k] += v; b[k] += v * v; }
            }
        }
#pragma unroll
    for (int k = 0; k < BNP_MAX_SLOTS; ++k) {
        const int c = t + 256 * k;
        if (c < C) { out[c] = a[k]; out[C + c] = b[k]; }
    }
}
// stage 2.  BWD: o0 = dbeta, o1 = dgamma; else o0 = mean, o1 = biased variance.  Workgroup = 32 channels x 32 slices of the partial rows (a thread walks
// rows s, s + 32, ... with four loads of each sum in flight; the slices meet through LDS in a fixed order): one thread per channel walking all 1 024 rows
// took 266 us per call, 10 % of the mother_stage step.
template <bool BWD>
__global__ __launch_bounds__(1024) void bn_partial_fold_kernel(const double* __restrict__ part, int nb, int64_t npix, int C, float* __restrict__ o0,
                                                               float* __restrict__ o1) {
    __shared__ double ra[1024], rb[1024];
    const int lc = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + lc;
    double sa = 0.0, sb = 0.0;
    if (c < C) {
        int k = sl;
        for (; k + 96 < nb; k += 128) {
            const double a0 = part[(size_t)k * 2 * C + c], a1 = part[(size_t)(k + 32) * 2 * C + c], a2 = part[(size_t)(k + 64) * 2 * C + c],
                         a3 = part[(size_t)(k + 96) * 2 * C + c];
            const double b0 = part[(size_t)k * 2 * C + C + c], b1 = part[(size_t)(k + 32) * 2 * C + C + c], b2 = part[(size_t)(k + 64) * 2 * C + C + c],
                         b3 = part[(size_t)(k + 96) * 2 * C + C + c];
            sa += (a0 + a1) + (a2 + a3);
            sb += (b0 + b1) + (b2 + b3);
        }
        for (; k < nb; k += 32) { sa += part[(size_t)k * 2 * C + c]; sb += part[(size_t)k * 2 * C + C + c]; }
    }
    ra[threadIdx.x] = sa; rb[threadIdx.x] = sb;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    sa = 0.0; sb = 0.0;
    for (int j = 0; j < 32; ++j) { sa += ra[j * 32 + lc]; sb += rb[j * 32 + lc]; }
    if (BWD) { o0[c] = (float)sa; o1[c] = (float)sb; }
    else {
        const double m = sa / (double)npix;
        o0[c] = (float)m;
        o1[c] = (float)fmax(sb / (double)npix - m * m, 0.0);      // biased batch variance (Keras normalises with it)
    }
}
inline int bnp_blocks(int64_t npix) { const int64_t b = (npix + 63) / 64; return (int)(b < BNP_MAX_BLOCKS ? (b < 1 ? 1 : b) : BNP_MAX_BLOCKS); }

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ var,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       float* __restrict__ out, int64_t n, int C, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e % C);
    const float y = (z[e] - mean[c]) * rsqrtf(var[c] + eps) * gamma[c] + beta[c];
    out[e] = accumulate ? out[e] + y : y;
}

// moving statistics: Keras momentum form; the moving variance is fed the Bessel-corrected batch variance (fused batch norm)
__global__ void bn_moving_kernel(const float* mean, const float* var, float* mov_mean, float* mov_var, int C, float momentum, double count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float unb = (float)((double)var[c] * (count > 1.0 ? count / (count - 1.0) : 1.0));
    mov_mean[c] = mov_mean[c] * momentum + mean[c] * (1.f - momentum);
    mov_var[c] = mov_var[c] * momentum + unb * (1.f - momentum);
}

// dgamma = sum dy xhat, dbeta = sum dy (per channel, double, fixed order)
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ mean,
                                                          const float* __restrict__ var, float eps, int64_t npix, int C,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double red[2 * 256];
    const int c = blockIdx.x;
    const double mu = mean[c], is = 1.0 / sqrt((double)var[c] + (double)eps);
    double v[2] = {0.0, 0.0};
    for (int64_t p = threadIdx.x; p < npix; p += 256) {
        const double d = dy[p * C + c];
        v[0] += d;
        v[1] += d * ((double)z[p * C + c] - mu) * is;
    }
    block_reduce<2>(v, red);
    if (threadIdx.x == 0) { dbeta[c] = (float)v[0]; dgamma[c] = (float)v[1]; }
}

// dz = gamma invstd (dy - dbeta / N - xhat dgamma / N)
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(const float* __restrict__ z, const float* __restrict__ dy, const float* __restrict__ mean,
                                                        const float* __restrict__ var, const float* __restrict__ gamma, float eps,
                                                        const float* __restrict__ dgamma, const float* __restrict__ dbeta, float* __restrict__ dz,
                                                        int64_t n, int C, double inv_count) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e % C);
    const float is = rsqrtf(var[c] + eps);
    const float xh = (z[e] - mean[c]) * is;
    dz[e] = gamma[c] * is * (dy[e] - (float)(dbeta[c] * inv_count) - xh * (float)(dgamma[c] * inv_count));
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
// kinds: 0 linear, 1 sigmoid, 2 tanh, 3 relu (SELD_ACT_*), 4 swish
__global__ __launch_bounds__(256) void act_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int kind) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float v = x[e];
    y[e] = kind == 1 ? sigm(v) : (kind == 2 ? tanhf(v) : (kind == 3 ? fmaxf(v, 0.f) : (kind == 4 ? v * sigm(v) : v)));
}
// dx (+)= dy act'(x), from the PRE-activation x
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n,
                                                      int kind, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const float v = x[e];
    float d;
    if (kind == 1) { const float s = sigm(v); d = s * (1.f - s); }
    else if (kind == 2) { const float t = tanhf(v); d = 1.f - t * t; }
    else if (kind == 3) d = v > 0.f ? 1.f : 0.f;
    else if (kind == 4) { const float s = sigm(v); d = s + v * s * (1.f - s); }
    else d = 1.f;
    const float g = dy[e] * d;
    dx[e] = accumulate ? dx[e] + g : g;
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n, float alpha) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) dst[e] += alpha * src[e];
}

// mode 0: dst[r][off + c] = src[r][c];  mode 1 (backward of the concatenation): src[r][c] += dst[r][off + c]
__global__ __launch_bounds__(256) void copy_channels_kernel(float* __restrict__ src, float* __restrict__ dst, int64_t rows, int Cs, int Cd, int off,
                                                            int mode) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= rows * Cs) return;
    const int64_t r = e / Cs;
    const int c = (int)(e - r * Cs);
    if (mode == 0) dst[r * Cd + off + c] = src[e];
    else src[e] += dst[r * Cd + off + c];
}

// squeeze: out[b][c] = mean over the HW pixels of x[b][.][c]; one workgroup per (b, c)
__global__ __launch_bounds__(256) void mean_hw_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int C) {
    __shared__ double red[256];
    const int c = blockIdx.x % C;
    const int64_t b = blockIdx.x / C;
    double v[1] = {0.0};
    for (int p = threadIdx.x; p < HW; p += 256) v[0] += x[(b * HW + p) * C + c];
    block_reduce<1>(v, red);
    if (threadIdx.x == 0) out[b * C + c] = (float)(v[0] / (double)HW);
}
// excite: y = x * s[b][c]
__global__ __launch_bounds__(256) void scale_hw_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int64_t n,
                                                       int HW, int C) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e % C);
    const int64_t b = e / ((int64_t)HW * C);
    y[e] = x[e] * s[b * C + c];
}
// ds[b][c] = sum over pixels of dy x (one workgroup per (b, c)); dx = dy s (+ dmean[b][c] / HW when dmean is given: the squeeze's gradient)
__global__ __launch_bounds__(256) void scale_hw_bwd_ds_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ ds, int HW,
                                                              int C) {
    __shared__ double red[256];
    const int c = blockIdx.x % C;
    const int64_t b = blockIdx.x / C;
    double v[1] = {0.0};
    for (int p = threadIdx.x; p < HW; p += 256) {
        const int64_t a = (b * HW + p) * C + c;
        v[0] += (double)dy[a] * (double)x[a];
    }
    block_reduce<1>(v, red);
    if (threadIdx.x == 0) ds[b * C + c] = (float)v[0];
}
__global__ __launch_bounds__(256) void scale_hw_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ s, const float* __restrict__ dmean,
                                                              float* __restrict__ dx, int64_t n, int HW, int C, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int c = (int)(e % C);
    const int64_t b = e / ((int64_t)HW * C);
    float g = dy[e] * s[b * C + c];
    if (dmean) g += dmean[b * C + c] * (1.f / (float)HW);
    dx[e] = accumulate ? dx[e] + g : g;
}

inline unsigned nblk(int64_t n) { return (unsigned)((n + 255) / 256); }
inline int ok() { return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP; }

}  // namespace

extern "C" {

int seld_m_conv_out(int in, int stride) { return same_out(in, stride); }

int seld_m_im2col(const float* x, float* col, int B, int H, int W, int C, int kh, int kw, int sh, int sw, void* stream) {
    if (!x || !col || B < 1 || H < 1 || W < 1 || C < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1) return SELD_ERR_INVALID;
    const int Ho = same_out(H, sh), Wo = same_out(W, sw);
    const int64_t total = (int64_t)B * Ho * Wo * kh * kw * C;
    if (nblk(total) == 0 || total > (int64_t)0x7fffffff * 256) return SELD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(im2col_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, col, total, H, W, C, kh, kw, sh, sw, Ho, Wo,
                       same_pad_before(H, kh, sh), same_pad_before(W, kw, sw));
    return ok();
}

int seld_m_col2im(const float* dcol, float* dx, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int accumulate, void* stream) {
    if (!dcol || !dx || B < 1 || H < 1 || W < 1 || C < 1 || kh < 1 || kw < 1 || sh < 1 || sw < 1) return SELD_ERR_INVALID;
    const int64_t total = (int64_t)B * H * W * C;
    hipLaunchKernelGGL(col2im_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, dcol, dx, total, H, W, C, kh, kw, sh, sw,
                       same_out(H, sh), same_out(W, sw), same_pad_before(H, kh, sh), same_pad_before(W, kw, sw), accumulate);
    return ok();
}

/* C[M,N] (+)= A[M,K] op(B) + bias on the fp32 MFMA GEMM (gemm.hip); transb = 1: B is [N,K] */
int seld_m_gemm(const float* A, const float* Bm, const float* bias, float* Cm, int M, int N, int K, int transb, int accumulate, void* stream) {
    if (!A || !Bm || !Cm) return SELD_ERR_INVALID;
    if (launch_gemm((hipStream_t)stream, A, K, Bm, transb ? K : N, bias, Cm, N, M, N, K, transb, 0, accumulate)) return SELD_ERR_INVALID;
    return ok();
}

int64_t seld_m_gemm_tn_scratch(int K1, int N) { return (int64_t)gemm_tn_max_splits() * ((int64_t)K1 * N + N); }

/* C[K1,N] = A[M,K1]^T B[M,N], colsum[N] = sum_m B[m,:] (may be NULL); slab: caller scratch of seld_m_gemm_tn_scratch(K1, N) floats.
 * seq > 0, shift = -1 | +1: row m of A is replaced by row m + shift of the same length-`seq` sequence, zero outside it (the recurrent
 * kernel's gradient h_prev^T dgh of a GRU direction); seq = 0: no shift */
int seld_m_gemm_tn(const float* A, const float* Bm, float* Cm, float* colsum, float* slab, int M, int K1, int N, int seq, int shift, void* stream) {
    if (!A || !Bm || !Cm || !slab || (seq > 0 && M % seq)) return SELD_ERR_INVALID;
    int ns = 0;
    if (launch_gemm_tn((hipStream_t)stream, A, K1, Bm, N, slab, &ns, M, K1, N, seq, seq > 0 ? shift : 0, colsum ? 1 : 0, 0, 0)) return SELD_ERR_INVALID;
    if (colsum) launch_reduce_slabs2((hipStream_t)stream, slab, ns, (int64_t)K1 * N + N, Cm, (int64_t)K1 * N, colsum, N);
    else launch_reduce_slabs((hipStream_t)stream, slab, ns, (int64_t)K1 * N + N, Cm, (int64_t)K1 * N, 0);
    return ok();
}

/* floats of caller scratch seld_m_bn_stats / seld_m_bn_bwd take for C channels (the per-workgroup partial sums of their first stage) */
int64_t seld_m_bn_scratch(int C) { return C > 0 ? (int64_t)BNP_MAX_BLOCKS * 4 * C : -1; }
int seld_m_bn_stats(const float* z, int64_t npix, int C, float* mean, float* var, float* scratch, void* stream) {
    if (!z || !mean || !var || npix < 1 || C < 1) return SELD_ERR_INVALID;
    if (!scratch || C > 256 * BNP_MAX_SLOTS || (reinterpret_cast<uintptr_t>(scratch) & 7)) {      // no scratch (or a very wide tensor): one workgroup per channel
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, z, npix, C, mean, var);
        return ok();
    }
    const int nb = bnp_blocks(npix);
    double* part = reinterpret_cast<double*>(scratch);
    hipLaunchKernelGGL((bn_partial_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, z, nullptr, nullptr, nullptr, 0.f, npix, C, part);
    hipLaunchKernelGGL((bn_partial_fold_kernel<false>), dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, nb, npix, C, mean, var);
    return ok();
}
int seld_m_bn_apply(const float* z, const float* mean, const float* var, const float* gamma, const float* beta, float eps, float* out,
                    int64_t npix, int C, int accumulate, void* stream) {
    if (!z || !mean || !var || !gamma || !beta || !out) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, z, mean, var, gamma, beta, eps, out, npix * C, C, accumulate);
    return ok();
}
int seld_m_bn_moving(const float* mean, const float* var, float* mov_mean, float* mov_var, int C, float momentum, int64_t count, void* stream) {
    if (!mean || !var || !mov_mean || !mov_var) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(bn_moving_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, mean, var, mov_mean, mov_var, C, momentum, (double)count);
    return ok();
}
int seld_m_bn_bwd(const float* z, const float* dy, const float* mean, const float* var, const float* gamma, float eps, float* dz, float* dgamma,
                  float* dbeta, int64_t npix, int C, float* scratch, void* stream) {
    if (!z || !dy || !mean || !var || !gamma || !dz || !dgamma || !dbeta) return SELD_ERR_INVALID;
    if (!scratch || C > 256 * BNP_MAX_SLOTS || (reinterpret_cast<uintptr_t>(scratch) & 7))
        hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, z, dy, mean, var, eps, npix, C, dgamma, dbeta);
    else {
        const int nb = bnp_blocks(npix);
        double* part = reinterpret_cast<double*>(scratch);
        hipLaunchKernelGGL((bn_partial_kernel<true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, z, dy, mean, var, eps, npix, C, part);
        hipLaunchKernelGGL((bn_partial_fold_kernel<true>), dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, nb, npix, C, dbeta, dgamma);
    }
    hipLaunchKernelGGL(bn_bwd_dz_kernel, dim3(nblk(npix * C)), dim3(256), 0, (hipStream_t)stream, z, dy, mean, var, gamma, eps, dgamma, dbeta, dz,
                       npix * C, C, 1.0 / (double)npix);
    return ok();
}

int seld_m_act(const float* x, float* y, int64_t n, int kind, void* stream) {
    if (!x || !y || kind < 0 || kind > 4) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(act_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, kind);
    return ok();
}
int seld_m_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int kind, int accumulate, void* stream) {
    if (!x || !dy || !dx || kind < 0 || kind > 4) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, kind, accumulate);
    return ok();
}
int seld_m_axpy(float* dst, const float* src, int64_t n, float alpha, void* stream) {
    if (!dst || !src) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(axpy_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, dst, src, n, alpha);
    return ok();
}
int seld_m_copy_channels(float* src, float* dst, int64_t rows, int Cs, int Cd, int off, int mode, void* stream) {
    if (!src || !dst || Cs < 1 || off < 0 || off + Cs > Cd) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(copy_channels_kernel, dim3(nblk(rows * Cs)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, Cs, Cd, off, mode);
    return ok();
}
int seld_m_mean_hw(const float* x, float* out, int B, int HW, int C, void* stream) {
    if (!x || !out) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(mean_hw_kernel, dim3((unsigned)B * C), dim3(256), 0, (hipStream_t)stream, x, out, HW, C);
    return ok();
}
int seld_m_scale_hw(const float* x, const float* s, float* y, int B, int HW, int C, void* stream) {
    if (!x || !s || !y) return SELD_ERR_INVALID;
    const int64_t n = (int64_t)B * HW * C;
    hipLaunchKernelGGL(scale_hw_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, x, s, y, n, HW, C);
    return ok();
}
/* y = x s: ds[b][c] = sum_px dy x */
int seld_m_scale_hw_bwd_ds(const float* x, const float* dy, float* ds, int B, int HW, int C, void* stream) {
    if (!x || !dy || !ds) return SELD_ERR_INVALID;
    hipLaunchKernelGGL(scale_hw_bwd_ds_kernel, dim3((unsigned)B * C), dim3(256), 0, (hipStream_t)stream, x, dy, ds, HW, C);
    return ok();
}
/* dx (+)= dy s + dmean / HW (dmean may be NULL) */
int seld_m_scale_hw_bwd_dx(const float* dy, const float* s, const float* dmean, float* dx, int B, int HW, int C, int accumulate, void* stream) {
    if (!dy || !s || !dx) return SELD_ERR_INVALID;
    const int64_t n = (int64_t)B * HW * C;
    hipLaunchKernelGGL(scale_hw_bwd_dx_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, dy, s, dmean, dx, n, HW, C, accumulate);
    return ok();
}

/* ---- the recurrent block, the losses and Adam on the CALLER'S stream, asynchronous like every other module operator (round 5: the composed
 * step makes no host synchronisation; the seld_k_* forms of these run on the null stream and synchronise the device) ---------------------- */
int seld_m_gru_fwd(const float* gx_f, const float* gx_b, const float* U_f, const float* U_b, const float* brec_f, const float* brec_b, float* h_f,
                   float* h_b, float* saved_f, float* saved_b, float* out, int B, int S, int units, void* stream) {
    if (units != 128) return SELD_ERR_UNSUPPORTED;
    if (!gx_f || !gx_b || !U_f || !U_b || !brec_f || !brec_b || !h_f || !h_b || B < 1 || S < 1) return SELD_ERR_INVALID;
    launch_gru_fwd((hipStream_t)stream, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, saved_f, saved_b, B, S);
    if (out) launch_mul((hipStream_t)stream, h_f, h_b, out, (int64_t)B * S * 128);
    return ok();
}
int seld_m_gru_bwd(const float* dout, const float* h_f, const float* h_b, const float* saved_f, const float* saved_b, const float* U_f,
                   const float* U_b, float* dgx_f, float* dgx_b, float* dgh_f, float* dgh_b, int B, int S, int units, void* stream) {
    if (units != 128) return SELD_ERR_UNSUPPORTED;
    if (!dout || !h_f || !h_b || !saved_f || !saved_b || !U_f || !U_b || !dgx_f || !dgx_b || !dgh_f || !dgh_b || B < 1 || S < 1) return SELD_ERR_INVALID;
    launch_gru_bwd((hipStream_t)stream, dout, h_f, h_b, saved_f, saved_b, U_f, U_b, dgx_f, dgx_b, dgh_f, dgh_b, B, S);
    return ok();
}
/* floats of caller scratch seld_m_losses needs for `rows` = B * S label frames */
int64_t seld_m_losses_scratch(int rows) { return rows > 0 ? (int64_t)loss_scratch_floats(rows) + 4 : -1; }
int seld_m_losses(const float* sed, const float* doa, const float* y_sed, const float* y_doa, const seld_loss_cfg* cfg, float* sloss, float* dloss,
                  float* dsed_pre, float* ddoa_pre, float* scratch, int B, int S, int nc, void* stream) {
    if (!sed || !doa || !y_sed || !y_doa || !cfg || !sloss || !dloss || !scratch || B < 1 || S < 1) return SELD_ERR_INVALID;
    const int rows = B * S;
    float* den = scratch + loss_scratch_floats(rows);
    if (cfg->doa_loss == SELD_DOA_MMSE) {
        if (cfg->mmse_den > 0.f) launch_fill((hipStream_t)stream, den, 1, cfg->mmse_den);
        else launch_mmse_den((hipStream_t)stream, y_doa, den, scratch, rows, nc);
    }
    launch_losses((hipStream_t)stream, sed, doa, y_sed, y_doa, cfg->doa_loss, cfg->w_sed, cfg->w_doa, cfg->sed_grad_scale, den, sloss, dloss, dsed_pre,
                  ddoa_pre, scratch, B, S, nc);
    return ok();
}
int seld_m_adam(float* theta, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int64_t step, void* stream) {
    if (!theta || !g || !m || !v || n <= 0 || step < 1) return SELD_ERR_INVALID;
    const double t = (double)step;
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    launch_adam((hipStream_t)stream, theta, g, m, v, n, lr_t, beta1, beta2, eps);
    return ok();
}

}  // extern "C"
