// xception.hip — the kernels of `xception_block`'s middle flow (spec/XCEPTION_BLOCK.md; model_config/xception_gru.json:2-11 names the
// block, the reference snapshot does not define it): residual modules of three  ReLU -> SeparableConv2D(64, 3, use_bias=False) ->
// BatchNormalization  on [B, S, 16, 64] NHWC fp32.  A separable convolution is a depthwise 3x3 (9 MACs per element: HBM-bound,
// written here as streaming kernels with float4 over the channels) followed by a 1x1 convolution = a [pixels, 64] x [64, 64]
// product (the fp32 MFMA GEMM of gemm.hip; its kernel gradient is the TN product of gemm.hip).  BatchNorm here has neither ReLU nor
// pooling behind it (the NEXT unit's ReLU is applied by its depthwise kernel on load), so it gets plain statistics / apply /
// backward kernels; the per-channel finalisation is bn_pool.hip's.
//
//   dw3x3_fwd        y[p][c]  = sum_taps k[tap][c] relu(x[p + tap][c])                       ('same' padding)
//   dw3x3_bwd_data   dx[p][c] = [x[p][c] > 0] sum_taps k[tap][c] dy[p - tap][c]  (+ add[p][c]: the residual branch's gradient)
//   dw3x3_bwd_w      dk[tap][c] = sum_p relu(x[p + tap][c]) dy[p][c]                         (per-workgroup slabs, fixed-order reduce)
//                    (fusing the two backward kernels into one pass over x / dy was measured: 5.1 ms per step against 3.0 — the
//                    row-per-workgroup shape the kernel-gradient sums need has 1/40 of the data kernel's parallelism)
//   bn_stats_plain   per-workgroup [sum z | sum z^2]                                        -> bn_finalize (bn_pool.hip)
//   bn_apply         out = z scale + shift (+ res)
//   bn_bwd_reduce_plain   per-workgroup [sum dy | sum dy xhat]                              -> bn_bwd_finalize
//   bn_bwd_dz_plain  dz = scale (dy - c1 - xhat c2)
#include "common.h"

#define XC_MAX_PARTIAL 512      // depthwise kernel-gradient slabs (one per workgroup of image rows): 2048 cost 0.4 ms per step more (slab traffic + combine)
#define XC_BN_PARTIAL 512       // BatchNorm partial sums: the single-workgroup finalisation reads all of them
#define XU_MAX_BLOCKS 512       // persistent workgroups of xc_unit_fwd: each leaves one BatchNorm partial
// ONE buffer (ctx xc_part) receives the BatchNorm partials of xc_unit_fwd (<= XU_MAX_BLOCKS), of the statistics / backward-sum passes
// (<= XC_BN_PARTIAL) and nothing else: its capacity is the largest of the producers' caps, not whichever constant happens to be equal today
constexpr int xc_cmax(int a, int b) { return a > b ? a : b; }
int xc_partial_capacity() { return xc_cmax(xc_cmax(XC_MAX_PARTIAL, XC_BN_PARTIAL), XU_MAX_BLOCKS); }

__device__ __forceinline__ float4 relu4(float4 v) { return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)); }
__device__ __forceinline__ float4 fma4v(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// one thread per (pixel, group of 4 channels); C = 64 -> 16 groups; k [9][64] (Keras depthwise_kernel [3,3,64,1]).  (A sliding-window
// form — a thread walks image rows, keeps its 3 x 3 neighbourhood in registers and loads one new row per output — was measured: 1.8
// against 1.4 ms per step forward: the nine L1-served loads of 4.9 M independent threads beat a third of the loads from 0.3 M.)
// `aff` (optional, [scale 64 | shift 64]): the unit's input is relu(x scale + shift) of a stored pre-BN tensor x instead of a stored
// activation (the previous unit's BatchNormalization folded into this unit's loads): forward source / backward gate source
template <bool BWD, bool AFF>
__global__ __launch_bounds__(256) void dw3x3_kernel(const float* __restrict__ src, const float* __restrict__ k, const float* __restrict__ xin,
                                                    const float* __restrict__ add, float* __restrict__ dst, int64_t npix, int H, int W,
                                                    const float* __restrict__ aff, int xcd_map) {
    // Workgroups are handed to the 8 XCDs round-robin (blockIdx % 8), and a workgroup of 16 pixels is one image row when W = 16: rows r - 1, r, r + 1
    // would sit on three different XCDs and every row of src would be fetched into three L2s (measured: 2.3 TB/s of algorithmic bytes with the
    // identity map).  XCD x takes the x-th CONTIGUOUS eighth of the rows instead: the halo rows hit in its own L2.
    unsigned lb = blockIdx.x;
    if (xcd_map) {
        const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lb & 7, j = lb >> 3;
        lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int64_t gid = (int64_t)lb * 256 + threadIdx.x;
    if (gid >= npix * 16) return;
    const int g = (int)(gid & 15);
    const int64_t p = gid >> 4;
    const int f = (int)(p % W), t = (int)((p / W) % H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dt = tap / 3 - 1, df = tap % 3 - 1;
        // forward: y[p] += k[tap] relu(x[p + tap]);  backward: dx[p] += k[tap] dy[p - tap]
        const int tt = BWD ? t - dt : t + dt, ff = BWD ? f - df : f + df;
        if (tt >= 0 && tt < H && ff >= 0 && ff < W) {
            float4 v = *reinterpret_cast<const float4*>(src + (p + (int64_t)(BWD ? -1 : 1) * (dt * W + df)) * 64 + 4 * g);
            if (!BWD && AFF) v = fma4v(v, reinterpret_cast<const float4*>(aff)[g], reinterpret_cast<const float4*>(aff + 64)[g]);
            if (!BWD) v = relu4(v);
            acc = fma4v(reinterpret_cast<const float4*>(k + tap * 64)[g], v, acc);
        }
    }
    if (BWD) {
        float4 xv = *reinterpret_cast<const float4*>(xin + p * 64 + 4 * g);
        if (AFF) xv = fma4v(xv, reinterpret_cast<const float4*>(aff)[g], reinterpret_cast<const float4*>(aff + 64)[g]);
        acc = make_float4(xv.x > 0.f ? acc.x : 0.f, xv.y > 0.f ? acc.y : 0.f, xv.z > 0.f ? acc.z : 0.f, xv.w > 0.f ? acc.w : 0.f);
        if (add) {
            const float4 a = *reinterpret_cast<const float4*>(add + p * 64 + 4 * g);
            acc = make_float4(acc.x + a.x, acc.y + a.y, acc.z + a.z, acc.w + a.w);
        }
    }
    *reinterpret_cast<float4*>(dst + p * 64 + 4 * g) = acc;
}

// W = 16 (the block's geometry): one workgroup = one image row (16 pixels x 16 channel groups), so the row index, the time index and the row's
// validity are workgroup-uniform (scalar unit), a lane's offset inside a row is 4 tid floats, and the nine taps are one scalar row base + a
// 32-bit lane offset each.  The generic kernel above spends ~300 vector + ~300 scalar instructions per float4 of output on 64-bit index
// arithmetic (p % W, p / W % H), per-tap bounds tests and branches — it ran at 2.3 TB/s of algorithmic bytes, instruction-bound; this form is
// ~60 vector instructions.  Same taps in the same order, the same fmaf per tap: the same bits.
template <bool BWD, bool AFF, bool EDGE>
__device__ __forceinline__ void dw3x3_w16_row(const float* __restrict__ srow, const float4 (&kk)[9], int tid, int t, int H, float4& acc) {
    const int f = tid >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dt = tap / 3 - 1, df = tap % 3 - 1;
        const int tt = BWD ? t - dt : t + dt, ff = BWD ? f - df : f + df;      // tt: uniform
        if (EDGE && (tt < 0 || tt >= H)) continue;
        const bool okf = df == 0 || (ff >= 0 && ff < 16);
        const float* q = srow + (BWD ? -1 : 1) * (dt * 1024 + df * 64) + 4 * tid;
        const float4 v = okf ? *reinterpret_cast<const float4*>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc = fma4v(kk[tap], v, acc);
    }
}
template <bool BWD, bool AFF>
__global__ __launch_bounds__(256) void dw3x3_w16_kernel(const float* __restrict__ src, const float* __restrict__ k, const float* __restrict__ xin,
                                                        const float* __restrict__ add, float* __restrict__ dst, int nrows, int H,
                                                        const float* __restrict__ aff, int xcd_map) {
    unsigned lb = blockIdx.x;
    if (xcd_map) {      // XCD x takes the x-th contiguous eighth of the rows (see dw3x3_kernel)
        const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lb & 7, j = lb >> 3;
        lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tid = threadIdx.x, g = tid & 15;
    const int t = (int)(lb % (unsigned)H);
    const float* srow = src + (size_t)lb * 1024;
    float4 kk[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) kk[tap] = reinterpret_cast<const float4*>(k + tap * 64)[g];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xv = make_float4(1.f, 1.f, 1.f, 1.f), av = make_float4(0.f, 0.f, 0.f, 0.f);
    if (BWD) {
        xv = *reinterpret_cast<const float4*>(xin + (size_t)lb * 1024 + 4 * tid);
        if (add) av = *reinterpret_cast<const float4*>(add + (size_t)lb * 1024 + 4 * tid);
    }
    if (!BWD && AFF) {
        // forward with the previous unit's BatchNormalization folded into the loads: relu(x scale + shift) per tap (fallback path: xc_fused_fwd = 0)
        const float4 sc = reinterpret_cast<const float4*>(aff)[g], sh = reinterpret_cast<const float4*>(aff + 64)[g];
        const int f = tid >> 4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dt = tap / 3 - 1, df = tap % 3 - 1;
            const int tt = t + dt, ff = f + df;
            if (tt < 0 || tt >= H) continue;
            if (ff >= 0 && ff < 16) {
                float4 v = *reinterpret_cast<const float4*>(srow + dt * 1024 + df * 64 + 4 * tid);
                v = relu4(fma4v(v, sc, sh));
                acc = fma4v(kk[tap], v, acc);
            }
        }
    } else if (t > 0 && t < H - 1) {
        if (BWD) dw3x3_w16_row<true, AFF, false>(srow, kk, tid, t, H, acc);
        else {
            // forward on a stored activation: ReLU on load
            const int f = tid >> 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dt = tap / 3 - 1, df = tap % 3 - 1, ff = f + df;
                const bool okf = df == 0 || (ff >= 0 && ff < 16);
                const float4 v = okf ? relu4(*reinterpret_cast<const float4*>(srow + dt * 1024 + df * 64 + 4 * tid)) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = fma4v(kk[tap], v, acc);
            }
        }
    } else {
        if (BWD) dw3x3_w16_row<true, AFF, true>(srow, kk, tid, t, H, acc);
        else {
            const int f = tid >> 4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dt = tap / 3 - 1, df = tap % 3 - 1, tt = t + dt, ff = f + df;
                if (tt < 0 || tt >= H) continue;
                const bool okf = df == 0 || (ff >= 0 && ff < 16);
                const float4 v = okf ? relu4(*reinterpret_cast<const float4*>(srow + dt * 1024 + df * 64 + 4 * tid)) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = fma4v(kk[tap], v, acc);
            }
        }
    }
    if (BWD) {
        if (AFF) xv = fma4v(xv, reinterpret_cast<const float4*>(aff)[g], reinterpret_cast<const float4*>(aff + 64)[g]);
        acc = make_float4(xv.x > 0.f ? acc.x : 0.f, xv.y > 0.f ? acc.y : 0.f, xv.z > 0.f ? acc.z : 0.f, xv.w > 0.f ? acc.w : 0.f);
        if (add) acc = make_float4(acc.x + av.x, acc.y + av.y, acc.z + av.z, acc.w + av.w);
    }
    *reinterpret_cast<float4*>(dst + (size_t)lb * 1024 + 4 * tid) = acc;
}

// Input gradient AND kernel-gradient slabs of the depthwise convolution in one pass (W = 16; round 5).  dk[tap][c] = sum_p a[p + tap][c] dy[p][c]
// is, summed over q = p + tap instead,  sum_q a[q][c] dy[q - tap][c]:  the thread that forms dx[q] = [a[q] > 0] sum_tap k[tap] dy[q - tap] already
// holds all nine dy[q - tap] and its own a[q] (the ReLU gate's source) in registers, so the kernel gradient costs nine more float4 FMAs per
// output and NO loads: the separate pass over (unit input, dy) — 157 MB per unit, dw3x3_bwd_w at 1.6 TB/s on the side stream — is gone.
// A workgroup takes XD_R consecutive image rows (256 threads = 16 pixels x 16 channel groups, one row at a time, each row's nine loads
// independent of the previous row's), keeps dk in 36 registers across them and leaves ONE slab [9][64]: columns combined through LDS in a
// fixed order, slabs by reduce_slabs in a fixed order.  The tap weights sit in LDS (re-read per row: registers are what bounds the occupancy
// this load-latency-bound kernel lives on).
// STATS (AFF units: xin is the PREVIOUS unit's pre-BN tensor z and dx is that BatchNormalization's output gradient): the backward sums of that
// BatchNormalization — [sum dx | sum dx xhat], xhat = (z - mean) invstd, xc_reduce_kernel<true>'s arithmetic per element — leave with the slab, one
// [128] partial per workgroup (bn_partial; folded 64 to 1 by xc_fold_partials before the finalisation): its separate pass over (z, dx) is gone too.
#define XD_R 4
#define XD_ROW (9 * 64 + 128 + 4)
template <bool AFF, bool STATS>
__global__ __launch_bounds__(256) void dw3x3_w16_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ k, const float* __restrict__ xin,
                                                                  const float* __restrict__ add, float* __restrict__ dx, float* __restrict__ slab,
                                                                  int nrows, int H, const float* __restrict__ aff, int xcd_map,
                                                                  const float* __restrict__ bn_mean, const float* __restrict__ bn_invstd,
                                                                  float* __restrict__ bn_partial) {
    static_assert(!STATS || AFF, "the sums belong to the BatchNormalization folded into this unit's loads");
    __shared__ __attribute__((aligned(16))) float ks[9 * 64];
    __shared__ __attribute__((aligned(16))) float red[16][XD_ROW];
    unsigned lb = blockIdx.x;
    if (xcd_map) {
        const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, x = lb & 7, j = lb >> 3;
        lb = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    }
    const int tid = threadIdx.x, g = tid & 15, f = tid >> 4;
    for (int i = tid; i < 9 * 16; i += 256) reinterpret_cast<float4*>(ks)[i] = reinterpret_cast<const float4*>(k)[i];
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AFF) { sc = reinterpret_cast<const float4*>(aff)[g]; sh = reinterpret_cast<const float4*>(aff + 64)[g]; }
    float4 dk[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) dk[tap] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, mu = s1, is = s1;
    if (STATS) { mu = reinterpret_cast<const float4*>(bn_mean)[g]; is = reinterpret_cast<const float4*>(bn_invstd)[g]; }
    __syncthreads();
    const int r0 = (int)lb * XD_R;
#pragma unroll 1
    for (int i = 0; i < XD_R; ++i) {
        const int r = r0 + i;
        if (r >= nrows) break;
        const int t = r % H;                                        // uniform
        const float* srow = dy + (size_t)r * 1024 + 4 * tid;
        float4 xv = *reinterpret_cast<const float4*>(xin + (size_t)r * 1024 + 4 * tid);
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
        if (add) av = *reinterpret_cast<const float4*>(add + (size_t)r * 1024 + 4 * tid);
        float4 v[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {                         // dy[q - tap]: all nine loads in flight before the first FMA
            const int dt = tap / 3 - 1, df = tap % 3 - 1;
            const int tt = t - dt, ff = f - df;
            const bool ok = tt >= 0 && tt < H && (df == 0 || (ff >= 0 && ff < 16));
            v[tap] = ok ? *reinterpret_cast<const float4*>(srow - (dt * 1024 + df * 64)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float4 zv = xv;
        if (AFF) xv = fma4v(xv, sc, sh);
        const float4 a = relu4(xv);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            acc = fma4v(reinterpret_cast<const float4*>(ks)[tap * 16 + g], v[tap], acc);
            dk[tap] = fma4v(a, v[tap], dk[tap]);
        }
        acc = make_float4(xv.x > 0.f ? acc.x : 0.f, xv.y > 0.f ? acc.y : 0.f, xv.z > 0.f ? acc.z : 0.f, xv.w > 0.f ? acc.w : 0.f);
        if (add) acc = make_float4(acc.x + av.x, acc.y + av.y, acc.z + av.z, acc.w + av.w);
        *reinterpret_cast<float4*>(dx + (size_t)r * 1024 + 4 * tid) = acc;
        if (STATS) {
            s1 = make_float4(s1.x + acc.x, s1.y + acc.y, s1.z + acc.z, s1.w + acc.w);
            s2 = make_float4(s2.x + acc.x * (zv.x - mu.x) * is.x, s2.y + acc.y * (zv.y - mu.y) * is.y, s2.z + acc.z * (zv.z - mu.z) * is.z,
                             s2.w + acc.w * (zv.w - mu.w) * is.w);
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<float4*>(&red[f][tap * 64 + 4 * g]) = dk[tap];
    if (STATS) {
        *reinterpret_cast<float4*>(&red[f][576 + 4 * g]) = s1;
        *reinterpret_cast<float4*>(&red[f][640 + 4 * g]) = s2;
    }
    __syncthreads();
    for (int i = tid; i < (STATS ? 9 * 64 + 128 : 9 * 64); i += 256) {
        float s_ = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) s_ += red[sl][i];
        if (i < 576) slab[(size_t)lb * 576 + i] = s_;
        else bn_partial[(size_t)lb * 128 + (i - 576)] = s_;
    }
}
// partial [n][128] -> out [(n + per - 1) / per][128]: `per` workgroup partials per output row (a multiple of 8: eight loads in flight), double
// accumulation, fixed order
__global__ __launch_bounds__(128) void xc_fold_partials_kernel(const float* __restrict__ partial, int n, int per, float* __restrict__ out) {
    const int j0 = blockIdx.x * per, j1 = j0 + per < n ? j0 + per : n;
    double s_ = 0.0;
    for (int j = j0; j < j1; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = j + u < j1 ? partial[(size_t)(j + u) * 128 + threadIdx.x] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s_ += (double)v[u];
    }
    out[(size_t)blockIdx.x * 128 + threadIdx.x] = (float)s_;
}
int launch_xc_fold_partials(hipStream_t st, const float* partial, int n, float* out, int* nout) {
    int per = (n + xc_partial_capacity() - 1) / xc_partial_capacity();      // the finalisation's buffer holds xc_partial_capacity() rows
    per = per < 16 ? 16 : (per + 7) & ~7;
    const int nb = (n + per - 1) / per;
    hipLaunchKernelGGL(xc_fold_partials_kernel, dim3((unsigned)nb), dim3(128), 0, st, partial, n, per, out);
    *nout = nb;
    return 0;
}

int xc_dw_fused_slabs(int B, int H) { return (B * H + XD_R - 1) / XD_R; }
int launch_dw3x3_bwd_fused(hipStream_t st, const float* dy, const float* k, const float* xin, const float* add, float* dx, float* slab, int* nslab,
                           int B, int H, int W, const float* aff, const float* bn_mean, const float* bn_invstd, float* bn_partial) {
    if (W != 16 || (int64_t)B * H >= (1 << 30)) return -3;
    if (bn_partial && (!aff || !bn_mean || !bn_invstd || add)) return -3;
    const int nb = xc_dw_fused_slabs(B, H);
    if (bn_partial) hipLaunchKernelGGL((dw3x3_w16_bwd_fused_kernel<true, true>), dim3((unsigned)nb), dim3(256), 0, st, dy, k, xin, add, dx, slab, B * H, H, aff, g_xc_xcd_map, bn_mean, bn_invstd, bn_partial);
    else if (aff) hipLaunchKernelGGL((dw3x3_w16_bwd_fused_kernel<true, false>), dim3((unsigned)nb), dim3(256), 0, st, dy, k, xin, add, dx, slab, B * H, H, aff, g_xc_xcd_map, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL((dw3x3_w16_bwd_fused_kernel<false, false>), dim3((unsigned)nb), dim3(256), 0, st, dy, k, xin, add, dx, slab, B * H, H, aff, g_xc_xcd_map, nullptr, nullptr, nullptr);
    *nslab = nb;
    return 0;
}

int g_xc_w16 = 1;          // option "xc_w16": the row-per-workgroup depthwise kernels for W = 16 (0: the generic kernel, for A/B)
int g_xc_xcd_map = 1;      // option "xc_xcd_map": XCD-contiguous row ranges in the depthwise kernels (0: identity map, for A/B)
int launch_dw3x3_fwd(hipStream_t st, const float* x, const float* k, float* y, int B, int H, int W, const float* aff) {
    const int64_t npix = (int64_t)B * H * W;
    if (W == 16 && g_xc_w16 && (int64_t)B * H < (1 << 30)) {
        if (aff) hipLaunchKernelGGL((dw3x3_w16_kernel<false, true>), dim3((unsigned)(B * H)), dim3(256), 0, st, x, k, nullptr, nullptr, y, B * H, H, aff, g_xc_xcd_map);
        else hipLaunchKernelGGL((dw3x3_w16_kernel<false, false>), dim3((unsigned)(B * H)), dim3(256), 0, st, x, k, nullptr, nullptr, y, B * H, H, aff, g_xc_xcd_map);
        return 0;
    }
    if (aff) hipLaunchKernelGGL((dw3x3_kernel<false, true>), dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, x, k, nullptr, nullptr, y, npix, H, W, aff, g_xc_xcd_map);
    else hipLaunchKernelGGL((dw3x3_kernel<false, false>), dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, x, k, nullptr, nullptr, y, npix, H, W, aff, g_xc_xcd_map);
    return 0;
}
int launch_dw3x3_bwd_data(hipStream_t st, const float* dy, const float* k, const float* xin, const float* add, float* dx, int B, int H,
                          int W, const float* aff) {
    const int64_t npix = (int64_t)B * H * W;
    if (W == 16 && g_xc_w16 && (int64_t)B * H < (1 << 30)) {
        if (aff) hipLaunchKernelGGL((dw3x3_w16_kernel<true, true>), dim3((unsigned)(B * H)), dim3(256), 0, st, dy, k, xin, add, dx, B * H, H, aff, g_xc_xcd_map);
        else hipLaunchKernelGGL((dw3x3_w16_kernel<true, false>), dim3((unsigned)(B * H)), dim3(256), 0, st, dy, k, xin, add, dx, B * H, H, aff, g_xc_xcd_map);
        return 0;
    }
    if (aff) hipLaunchKernelGGL((dw3x3_kernel<true, true>), dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, dy, k, xin, add, dx, npix, H, W, aff, g_xc_xcd_map);
    else hipLaunchKernelGGL((dw3x3_kernel<true, false>), dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, dy, k, xin, add, dx, npix, H, W, aff, g_xc_xcd_map);
    return 0;
}

// dk[tap][c] = sum_p relu(x[p + tap][c]) dy[p][c].  A workgroup takes whole image rows (W pixels x 64 channels = one contiguous
// 4 W floats x 16 line): thread (slot = tid >> 4, g = tid & 15) owns pixel column `slot` (W = 16) of the rows blockIdx, + gridDim, ...
// and keeps the 3 x 3 x 4 sums in registers; the three input rows it needs are read once each per output row through L1 (the
// neighbouring columns belong to the neighbouring threads of the same workgroup).  Slots combined through LDS in a fixed order
// -> slab[blockIdx][9][64]
template <bool AFF>
__global__ __launch_bounds__(256) void dw3x3_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab,
                                                          int64_t nrows, int H, int W, const float* __restrict__ aff) {
    __shared__ float red[16][9 * 64 + 4];
    const int tid = threadIdx.x, g = tid & 15, slot = tid >> 4;
    float4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 asc = make_float4(1.f, 1.f, 1.f, 1.f), ash = zero4;          // identity unless the input is a pre-BN tensor (see dw3x3_kernel)
    if (AFF) { asc = reinterpret_cast<const float4*>(aff)[g]; ash = reinterpret_cast<const float4*>(aff + 64)[g]; }
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const int t = (int)(r % H);
        for (int f = slot; f < W; f += 16) {
            const int64_t p = r * W + f;
            const float4 d = *reinterpret_cast<const float4*>(dy + p * 64 + 4 * g);
            float4 v[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {          // all nine loads in flight before the first FMA
                const int dt = tap / 3 - 1, df = tap % 3 - 1;
                const bool ok = t + dt >= 0 && t + dt < H && f + df >= 0 && f + df < W;
                v[tap] = ok ? *reinterpret_cast<const float4*>(x + (p + dt * W + df) * 64 + 4 * g) : zero4;
            }
            if (AFF) {                                   // after all nine loads are in flight; padding stays 0 (the ACTIVATION is zero-padded)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dt = tap / 3 - 1, df = tap % 3 - 1;
                    const bool ok = t + dt >= 0 && t + dt < H && f + df >= 0 && f + df < W;
                    if (ok) v[tap] = fma4v(v[tap], asc, ash);
                }
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[tap] = fma4v(relu4(v[tap]), d, acc[tap]);
        }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) *reinterpret_cast<float4*>(&red[slot][tap * 64 + 4 * g]) = acc[tap];
    __syncthreads();
    for (int i = tid; i < 9 * 64; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) s += red[sl][i];
        slab[(size_t)blockIdx.x * 576 + i] = s;
    }
}

int launch_dw3x3_bwd_w(hipStream_t st, const float* x, const float* dy, float* slab, int* nslab, int B, int H, int W, const float* aff) {
    const int64_t nrows = (int64_t)B * H;
    int64_t blocks = nrows < XC_MAX_PARTIAL ? nrows : XC_MAX_PARTIAL;
    if (aff) hipLaunchKernelGGL(dw3x3_bwd_w_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, x, dy, slab, nrows, H, W, aff);
    else hipLaunchKernelGGL(dw3x3_bwd_w_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, x, dy, slab, nrows, H, W, aff);
    *nslab = (int)blocks;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// One separable-convolution unit's forward in ONE pass (W = 16):  x -> ReLU -> depthwise 3x3 -> dwo (stored: the backward's pointwise
// kernel gradient needs it) -> pointwise 64 x 64 on the fp32 MFMA -> z, with BatchNorm's [sum z | sum z^2] partials from the
// accumulators.  Run as three kernels (depthwise, GEMM, statistics) the unit moves 550 MB through HBM; this way 236 MB.
// Tile = 8 image rows x 16 columns = 128 pixels, 4 waves; LDS: the ReLU'd input rows with a one-row halo [10][16][64] (40 KB) and
// the depthwise output TRANSPOSED [64 ch][128 px + pad] (33 KB) = the MFMA's A operand (lane = pixel, k = channel: one ds_read_b32
// per MFMA, conflict-free); the pointwise weights live in 64 registers per lane (B operand).  Persistent workgroups, two per CU:
// one's loads and depthwise phase run beside the other's MFMAs.
#define XU_ROWS 8
#define XU_LDA 129      // odd row stride: the transposed depthwise writes (row 4 g + j, 16 groups per wave) spread over the banks
__global__ __launch_bounds__(256, 2) void xc_unit_fwd_kernel(const float* __restrict__ x, const float* __restrict__ kdw, const float* __restrict__ wpw,
                                                             float* __restrict__ dwo, float* __restrict__ z, float* __restrict__ partial, int B, int H,
                                                             int want_stats, const float* __restrict__ aff) {
    constexpr int W = 16, TP = XU_ROWS * W;
    extern __shared__ __attribute__((aligned(16))) float xu_smem[];
    float* R = xu_smem;                          // [(XU_ROWS + 2)][16][64]
    float* At = R + (XU_ROWS + 2) * W * 64;      // [64][XU_LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int g = tid & 15, pslot = tid >> 4;
    float4 kw[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) kw[i] = reinterpret_cast<const float4*>(kdw + i * 64)[g];
    float wreg[2][32];                           // B operand: W[2 s + kk][32 c + li]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s_ = 0; s_ < 32; ++s_) wreg[c][s_] = wpw[(2 * s_ + kk) * 64 + 32 * c + li];
    const int tiles_per_img = (H + XU_ROWS - 1) / XU_ROWS, ntiles = B * tiles_per_img;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // The input rows of a tile are requested one tile ahead (10 float4 per thread: row u of the region, this thread's column slot), so
    // that a tile's first phase does not start with a round trip to memory; BatchNorm-on-load / ReLU are applied at the commit.
    constexpr int NPRE = XU_ROWS + 2;
    float4 pre[NPRE];
    float4 asc = make_float4(1.f, 1.f, 1.f, 1.f), ash = zero4;
    if (aff) { asc = reinterpret_cast<const float4*>(aff)[tid & 15]; ash = reinterpret_cast<const float4*>(aff + 64)[tid & 15]; }
#define XU_ISSUE(tile_)                                                                                            \
    {                                                                                                              \
        const int b_ = (tile_) / tiles_per_img, t0_ = ((tile_) - b_ * tiles_per_img) * XU_ROWS;                    \
        _Pragma("unroll") for (int u = 0; u < NPRE; ++u) {                                                         \
            const int t_ = t0_ - 1 + u;                                                                            \
            const bool ok_ = t_ >= 0 && t_ < H;                                                                    \
            pre[u] = reinterpret_cast<const float4*>(x + ((size_t)(b_ * H + (ok_ ? t_ : 0)) * W) * 64)[tid];       \
        }                                                                                                          \
    }
    if ((int)blockIdx.x < ntiles) XU_ISSUE((int)blockIdx.x)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img, t0 = (tile - b * tiles_per_img) * XU_ROWS;
        // ---- commit input rows t0 - 1 .. t0 + XU_ROWS (zeros outside the image), BatchNorm of the previous unit and ReLU on the way in
#pragma unroll
        for (int u = 0; u < NPRE; ++u) {
            const int t = t0 - 1 + u;
            float4 v = zero4;
            if (t >= 0 && t < H) v = relu4(aff ? fma4v(pre[u], asc, ash) : pre[u]);
            reinterpret_cast<float4*>(R)[tid + 256 * u] = v;
        }
        __syncthreads();
        {   // next tile's rows: in flight under this tile's depthwise and pointwise phases (the last tile re-reads itself)
            const int nxt = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
            XU_ISSUE(nxt)
        }
        // ---- depthwise: pixel p = 16 it + pslot (row p >> 4, column p & 15), channels 4 g ..
#pragma unroll
        for (int it = 0; it < TP / 16; ++it) {
            const int p = 16 * it + pslot, r = p >> 4, f = p & 15;
            float4 acc = zero4;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dt = tap / 3, df = tap % 3 - 1;           // region row r + dt holds image row t0 + r + dt - 1
                if (f + df >= 0 && f + df < W) acc = fma4v(kw[tap], reinterpret_cast<const float4*>(R + ((r + dt) * W + f + df) * 64)[g], acc);
            }
            if (t0 + r < H) *reinterpret_cast<float4*>(dwo + ((size_t)(b * H + t0 + r) * W + f) * 64 + 4 * g) = acc;
            At[(4 * g + 0) * XU_LDA + p] = acc.x; At[(4 * g + 1) * XU_LDA + p] = acc.y;
            At[(4 * g + 2) * XU_LDA + p] = acc.z; At[(4 * g + 3) * XU_LDA + p] = acc.w;
        }
        __syncthreads();
        // ---- pointwise: wave w = pixels 32 w .. 32 w + 31 of the tile, two 32-channel output tiles
        f32x16 acc[2] = {zero16(), zero16()};
        const float* arow = At + kk * XU_LDA + 32 * wave + li;
#pragma unroll
        for (int s_ = 0; s_ < 32; ++s_) {
            const float a = arow[2 * s_ * XU_LDA];
            acc[0] = MFMA_F32_32x32x2(a, wreg[0][s_], acc[0]);
            acc[1] = MFMA_F32_32x32x2(a, wreg[1][s_], acc[1]);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int pq = 32 * wave + 8 * q + 4 * kk;          // pixels pq .. pq + 3 of the tile in registers 4 q .. 4 q + 3
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (t0 + ((pq + j) >> 4) < H) { s1[c] += acc[c][4 * q + j]; s2[c] = fmaf(acc[c][4 * q + j], acc[c][4 * q + j], s2[c]); }
                const float4 o = quad_transpose4(acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3], li);
                const int p = pq + (li & 3);
                if (t0 + (p >> 4) < H)
                    *reinterpret_cast<float4*>(z + ((size_t)(b * H + t0 + (p >> 4)) * W + (p & 15)) * 64 + 32 * c + (li & ~3)) = o;
            }
        __syncthreads();        // At and R are free for the next tile
    }
#undef XU_ISSUE
    if (want_stats) {
        float* red = R;          // [4][128]
#pragma unroll
        for (int c = 0; c < 2; ++c) { s1[c] += __shfl_xor(s1[c], 32); s2[c] += __shfl_xor(s2[c], 32); }
        if (kk == 0) {
            red[wave * 128 + li] = s1[0]; red[wave * 128 + 32 + li] = s1[1];
            red[wave * 128 + 64 + li] = s2[0]; red[wave * 128 + 96 + li] = s2[1];
        }
        __syncthreads();
        if (tid < 128) partial[(size_t)blockIdx.x * 128 + tid] = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
    }
}
int launch_xc_unit_fwd(hipStream_t st, const float* x, const float* kdw, const float* wpw, float* dwo, float* z, float* partial, int* npartial,
                       int B, int H, int W, const float* aff) {
    if (W != 16) return -2;
    const int ntiles = B * ((H + XU_ROWS - 1) / XU_ROWS);
    const int grid = ntiles < XU_MAX_BLOCKS ? ntiles : XU_MAX_BLOCKS;
    const size_t smem = (size_t)((XU_ROWS + 2) * 16 * 64 + 64 * XU_LDA) * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(xc_unit_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(xc_unit_fwd_kernel, dim3(grid), dim3(256), smem, st, x, kdw, wpw, dwo, z, partial, B, H, partial ? 1 : 0, aff);
    if (npartial) *npartial = grid;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The pointwise half of a unit's backward in ONE pass:  dz = BatchNorm'(gY) formed on load (never stored), F1 = dz W^T (gradient
// w.r.t. the depthwise output) and the kernel gradient dW = dwo^T dz, both on the fp32 MFMA.  As three kernels (xc_bn_bwd_dz, gemm_tn +
// combine, gemm_f32) the step moved 7 tensor passes per unit; this way 4 (z, gY, dwo in, F1 out).
// Tile = 128 pixels, 4 waves.  LDS: dz [128][65] (row stride 65: a lane per PIXEL reading channel k is conflict-free, and so is a lane
// per CHANNEL reading pixel k) and dwo [128][64].  F1: wave w = pixels 32 w .., two 32-channel tiles, W^T in 64 registers per lane.
// dW: wave (a, c) owns the 32 x 32 tile (input channels 32 a .., output channels 32 c ..) over ALL pixels the workgroup visits
// (persistent, two per CU); one slab of 4 096 floats per workgroup, combined in a fixed order by reduce_slabs.
#define XPB_MAX_BLOCKS 512
__global__ __launch_bounds__(256, 2) void xc_pw_bwd_kernel(const float* __restrict__ z, const float* __restrict__ gy, const float* __restrict__ dwo,
                                                           const float* __restrict__ wpw, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ c1c2, float* __restrict__ f1, float* __restrict__ slab,
                                                           int64_t npix) {
    extern __shared__ __attribute__((aligned(16))) float xpb_smem[];
    float* Dw = xpb_smem;                 // [128][64]
    float* Dz = xpb_smem + 128 * 64;      // [128][65]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int g = tid & 15;
    const int wa = wave >> 1, wc = wave & 1;
    float wreg[2][32];                           // B operand of F1 = dz W^T: B[k = o][n = i] = W[i][o], k = 2 s + kk, n = 32 c + li
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int s_ = 0; s_ < 32; ++s_) wreg[c][s_] = wpw[(32 * c + li) * 64 + 2 * s_ + kk];
    const float4 mu = reinterpret_cast<const float4*>(mean)[g], is = reinterpret_cast<const float4*>(invstd)[g];
    const float4 sc = reinterpret_cast<const float4*>(scale)[g], c1 = reinterpret_cast<const float4*>(c1c2)[g];
    const float4 c2 = reinterpret_cast<const float4*>(c1c2 + 64)[g];
    f32x16 dw = zero16();
    const int64_t ntiles = (npix + 127) / 128;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p0 = tile * 128;
        // ---- stage: thread = (pixel slot tid >> 4, channels 4 g ..), 8 pixels each
        float4 zv[8], dv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t p = p0 + 16 * u + (tid >> 4);
            const bool ok = p < npix;
            const size_t o_ = (size_t)(ok ? p : 0) * 16 + g;
            zv[u] = reinterpret_cast<const float4*>(z)[o_];
            dv[u] = reinterpret_cast<const float4*>(gy)[o_];
            wv[u] = reinterpret_cast<const float4*>(dwo)[o_];
            if (!ok) { dv[u] = make_float4(0.f, 0.f, 0.f, 0.f); wv[u] = dv[u]; zv[u] = mu; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int pl = 16 * u + (tid >> 4);
            const bool ok = p0 + pl < npix;
            // the expression of xc_bn_bwd_dz_kernel, bit for bit
            float4 o;
            o.x = sc.x * (dv[u].x - c1.x - (zv[u].x - mu.x) * is.x * c2.x);
            o.y = sc.y * (dv[u].y - c1.y - (zv[u].y - mu.y) * is.y * c2.y);
            o.z = sc.z * (dv[u].z - c1.z - (zv[u].z - mu.z) * is.z * c2.z);
            o.w = sc.w * (dv[u].w - c1.w - (zv[u].w - mu.w) * is.w * c2.w);
            if (!ok) o = make_float4(0.f, 0.f, 0.f, 0.f);       // pixels past the end add nothing to dW
            float* d = Dz + pl * 65 + 4 * g;
            d[0] = o.x; d[1] = o.y; d[2] = o.z; d[3] = o.w;
            *reinterpret_cast<float4*>(Dw + pl * 64 + 4 * g) = wv[u];
        }
        __syncthreads();
        // ---- F1 tile of this wave's 32 pixels
        f32x16 acc[2] = {zero16(), zero16()};
        const float* arow = Dz + (32 * wave + li) * 65 + kk;
#pragma unroll
        for (int s_ = 0; s_ < 32; ++s_) {
            const float a = arow[2 * s_];
            acc[0] = MFMA_F32_32x32x2(a, wreg[0][s_], acc[0]);
            acc[1] = MFMA_F32_32x32x2(a, wreg[1][s_], acc[1]);
        }
        // ---- dW tile (wa, wc) over the 128 pixels: A[i][k = pixel] = dwo[pixel][32 wa + i], B[k = pixel][o] = dz[pixel][32 wc + o]
        const float* ap = Dw + kk * 64 + 32 * wa + li;
        const float* bp = Dz + kk * 65 + 32 * wc + li;
#pragma unroll 16
        for (int s_ = 0; s_ < 64; ++s_) dw = MFMA_F32_32x32x2(ap[2 * s_ * 64], bp[2 * s_ * 65], dw);
        // ---- store F1: accumulator row = pixel, column = input channel
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 o = quad_transpose4(acc[c][4 * q], acc[c][4 * q + 1], acc[c][4 * q + 2], acc[c][4 * q + 3], li);
                const int64_t p = p0 + 32 * wave + 8 * q + 4 * kk + (li & 3);
                if (p < npix) *reinterpret_cast<float4*>(f1 + (size_t)p * 64 + 32 * c + (li & ~3)) = o;
            }
        __syncthreads();        // the tile's LDS images are free
    }
    // this workgroup's share of dW [64 in][64 out]: rows 32 wa + (accumulator row), columns 32 wc + li
    float* out = slab + (size_t)blockIdx.x * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(out + (size_t)(32 * wa + 8 * q + 4 * kk + (li & 3)) * 64 + 32 * wc + (li & ~3)) =
            quad_transpose4(dw[4 * q], dw[4 * q + 1], dw[4 * q + 2], dw[4 * q + 3], li);
}
int xc_pw_bwd_slabs() { return XPB_MAX_BLOCKS; }
int launch_xc_pw_bwd(hipStream_t st, const float* z, const float* gy, const float* dwo, const float* wpw, const float* mean, const float* invstd,
                     const float* scale, const float* c1c2, float* f1, float* slab, int* nslab, int64_t npix) {
    const int64_t ntiles = (npix + 127) / 128;
    const int grid = (int)(ntiles < XPB_MAX_BLOCKS ? ntiles : XPB_MAX_BLOCKS);
    const size_t smem = (size_t)(128 * 64 + 128 * 65) * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void*>(xc_pw_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(xc_pw_bwd_kernel, dim3(grid), dim3(256), smem, st, z, gy, dwo, wpw, mean, invstd, scale, c1c2, f1, slab, npix);
    *nslab = grid;
    return 0;
}

// per-workgroup [sum a | sum a b] over the pixels, 64 channels: STATS: a = b = z (sum z, sum z^2);
// BWD: a = dy, b = xhat = (z - mean) invstd (sum dy, sum dy xhat).  partial[blockIdx][128]
template <bool BWD>
__global__ __launch_bounds__(256) void xc_reduce_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                                        float* __restrict__ partial, int64_t npix) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, g = tid & 15, slot = tid >> 4;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, mu = s1, is = s1;
    if (BWD) { mu = reinterpret_cast<const float4*>(mean)[g]; is = reinterpret_cast<const float4*>(invstd)[g]; }
    for (int64_t p = (int64_t)blockIdx.x * 16 + slot; p < npix; p += (int64_t)gridDim.x * 16) {
        const float4 zv = *reinterpret_cast<const float4*>(z + p * 64 + 4 * g);
        if (BWD) {
            const float4 d = *reinterpret_cast<const float4*>(dy + p * 64 + 4 * g);
            s1 = make_float4(s1.x + d.x, s1.y + d.y, s1.z + d.z, s1.w + d.w);
            s2 = make_float4(s2.x + d.x * (zv.x - mu.x) * is.x, s2.y + d.y * (zv.y - mu.y) * is.y, s2.z + d.z * (zv.z - mu.z) * is.z,
                             s2.w + d.w * (zv.w - mu.w) * is.w);
        } else {
            s1 = make_float4(s1.x + zv.x, s1.y + zv.y, s1.z + zv.z, s1.w + zv.w);
            s2 = fma4v(zv, zv, s2);
        }
    }
    *reinterpret_cast<float4*>(&red[tid * 8]) = s1;
    *reinterpret_cast<float4*>(&red[tid * 8 + 4]) = s2;
    __syncthreads();
    if (tid < 128) {
        const int kind = tid >> 6, ch = tid & 63, gg = ch >> 2, cc = ch & 3;
        float s = 0.f;
        for (int sl = 0; sl < 16; ++sl) s += red[(sl * 16 + gg) * 8 + kind * 4 + cc];
        partial[(size_t)blockIdx.x * 128 + tid] = s;
    }
}

int launch_xc_bn_stats(hipStream_t st, const float* z, float* partial, int* npartial, int64_t npix) {
    int64_t blocks = (npix + 15) / 16;
    if (blocks > XC_BN_PARTIAL) blocks = XC_BN_PARTIAL;
    hipLaunchKernelGGL(xc_reduce_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, z, nullptr, nullptr, nullptr, partial, npix);
    *npartial = (int)blocks;
    return 0;
}
int launch_xc_bn_bwd_reduce(hipStream_t st, const float* z, const float* dy, const float* mean, const float* invstd, float* partial,
                            int* npartial, int64_t npix) {
    int64_t blocks = (npix + 15) / 16;
    if (blocks > XC_BN_PARTIAL) blocks = XC_BN_PARTIAL;
    hipLaunchKernelGGL(xc_reduce_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, z, dy, mean, invstd, partial, npix);
    *npartial = (int)blocks;
    return 0;
}

// out = z scale + shift (+ res)
__global__ __launch_bounds__(256) void xc_bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ res,
                                                          float* __restrict__ out, int64_t n4) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int g = (int)(gid & 15);
    float4 o = fma4v(reinterpret_cast<const float4*>(z)[gid], reinterpret_cast<const float4*>(scale)[g], reinterpret_cast<const float4*>(shift)[g]);
    if (res) {
        const float4 r = reinterpret_cast<const float4*>(res)[gid];
        o = make_float4(o.x + r.x, o.y + r.y, o.z + r.z, o.w + r.w);
    }
    reinterpret_cast<float4*>(out)[gid] = o;
}
int launch_xc_bn_apply(hipStream_t st, const float* z, const float* scale, const float* shift, const float* res, float* out, int64_t npix) {
    hipLaunchKernelGGL(xc_bn_apply_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, z, scale, shift, res, out, npix * 16);
    return 0;
}

// dz = scale (dy - c1 - xhat c2), xhat = (z - mean) invstd;  c1c2 = [c1 | c2]
__global__ __launch_bounds__(256) void xc_bn_bwd_dz_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ scale, const float* __restrict__ c1c2,
                                                           float* __restrict__ dz, int64_t n4) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n4) return;
    const int g = (int)(gid & 15);
    const float4 zv = reinterpret_cast<const float4*>(z)[gid], d = reinterpret_cast<const float4*>(dy)[gid];
    const float4 mu = reinterpret_cast<const float4*>(mean)[g], is = reinterpret_cast<const float4*>(invstd)[g];
    const float4 sc = reinterpret_cast<const float4*>(scale)[g], c1 = reinterpret_cast<const float4*>(c1c2)[g];
    const float4 c2 = reinterpret_cast<const float4*>(c1c2 + 64)[g];
    float4 o;
    o.x = sc.x * (d.x - c1.x - (zv.x - mu.x) * is.x * c2.x);
    o.y = sc.y * (d.y - c1.y - (zv.y - mu.y) * is.y * c2.y);
    o.z = sc.z * (d.z - c1.z - (zv.z - mu.z) * is.z * c2.z);
    o.w = sc.w * (d.w - c1.w - (zv.w - mu.w) * is.w * c2.w);
    reinterpret_cast<float4*>(dz)[gid] = o;
}
int launch_xc_bn_bwd_dz(hipStream_t st, const float* z, const float* dy, const float* mean, const float* invstd, const float* scale,
                        const float* c1c2, float* dz, int64_t npix) {
    hipLaunchKernelGGL(xc_bn_bwd_dz_kernel, dim3((unsigned)((npix * 16 + 255) / 256)), dim3(256), 0, st, z, dy, mean, invstd, scale, c1c2, dz,
                       npix * 16);
    return 0;
}

// identity BatchNorm coefficients for the exit pool (ReLU -> MaxPooling2D((1, 8)) reuses bn_relu_pool_fwd / bn_pool_bwd_dz):
// ident = [mean 0 | invstd 1 | scale 1 | shift 0 | c1 0 | c2 0] x 64
__global__ void xc_ident_kernel(float* ident) {
    const int i = threadIdx.x;      // 384
    ident[i] = (i >= 64 && i < 192) ? 1.f : 0.f;
}
int launch_xc_ident(hipStream_t st, float* ident) {
    hipLaunchKernelGGL(xc_ident_kernel, dim3(1), dim3(384), 0, st, ident);
    return 0;
}
