// kernels_api.hip — per-kernel C ABI entry points (include/seld_hip.h, "seld_k_*"): each runs ONE
// kernel family of the hot path on caller-provided device buffers so the parity tests can compare it
// with the oracle in isolation.  Null stream; scratch is allocated and freed per call (test use only).
#include "common.h"
#include <algorithm>
#include "../../include/seld_hip.h"
#include <vector>
#include <math.h>
#include <string.h>

namespace {
int g_conv64_split_bf16 = 1;
int g_conv1_split_bf16 = 1;
int g_rn_split_bf16 = 1;
int g_gemm_tn_sb = 1;
struct Scratch {
    std::vector<void*> p;
    float* get(size_t n) { void* q = nullptr; if (hipMalloc(&q, n * sizeof(float) + 256) != hipSuccess) return nullptr; p.push_back(q); return (float*)q; }
    ~Scratch() { hipDeviceSynchronize(); for (void* q : p) hipFree(q); }
};
int done() { return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP; }

__global__ void coeffs_kernel(const float* mean, const float* invstd, const float* gamma, const float* beta, float* scale,
                              float* shift, int C) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * invstd[c];
    scale[c] = sc;
    shift[c] = beta[c] - mean[c] * sc;
}
}  // namespace

extern "C" {

int seld_k_set_option(const char* key, int value) {
    if (!key) return SELD_ERR_INVALID;
    if (!strcmp(key, "conv64_split_bf16")) { g_conv64_split_bf16 = value != 0; return SELD_OK; }
    if (!strcmp(key, "conv1_split_bf16")) { g_conv1_split_bf16 = value != 0; return SELD_OK; }
    if (!strcmp(key, "gru_var")) { g_gru_var = value; return SELD_OK; }
    if (!strcmp(key, "conv64_dbuf")) { g_conv64_dbuf = value != 0; return SELD_OK; }
    if (!strcmp(key, "gemm_tn_split_bf16")) { g_gemm_tn_sb = value != 0; return SELD_OK; }
    if (!strcmp(key, "bf16_single")) { g_mfma_one = value != 0; return SELD_OK; }
    if (!strcmp(key, "gsb_dbg")) { g_gsb_dbg = value; return SELD_OK; }
    if (!strcmp(key, "xc_w16")) { g_xc_w16 = value != 0; return SELD_OK; }
    if (!strcmp(key, "xc_xcd_map")) { g_xc_xcd_map = value != 0; return SELD_OK; }
    if (!strcmp(key, "bwd_four_products")) { g_bwd_four = value != 0; return SELD_OK; }
    if (!strcmp(key, "rn_split_bf16")) { g_rn_split_bf16 = value != 0; return SELD_OK; }
    return SELD_ERR_INVALID;
}

int seld_k_conv3x3_fwd(const float* x, const float* w, const float* bias, float* z, float* stats, int B, int H, int W,
                       int Cin, int Cout) {
    if (!x || !w || !z || Cout != 64) return SELD_ERR_UNSUPPORTED;
    Scratch s;
    float* part = stats ? s.get((size_t)conv_stat_partial_capacity() * 128) : nullptr;
    if (stats && !part) return SELD_ERR_NOMEM;
    int np = 0, rc;
    if (Cin == 64 && g_conv64_split_bf16) {
        unsigned short* wsp = reinterpret_cast<unsigned short*>(s.get(9 * 3 * 4096 / 2 + 16));
        if (!wsp) return SELD_ERR_NOMEM;
        launch_split_weights(0, w, wsp);
        rc = launch_conv64_fwd_sb(0, x, wsp, bias, z, part, &np, B, H, W);
    } else if (Cin == 64) rc = launch_conv64_fwd(0, x, w, bias, z, part, &np, B, H, W);
    else if (W == 64) rc = launch_conv_first_fwd(0, x, w, bias, z, part, &np, B, H, Cin);
    else return SELD_ERR_UNSUPPORTED;
    if (rc) return SELD_ERR_UNSUPPORTED;
    if (stats) launch_reduce_slabs(0, part, np, 128, stats, 128, 0);
    return done();
}

int seld_k_conv_first_fwd_pool(const float* x, const float* w, const float* bias, const float* gamma, float* z, float* zext,
                               unsigned char* amax, float* stats, int B, int H, int Cin) {
    if (!x || !w || !gamma || !zext || (z != nullptr && amax == nullptr)) return SELD_ERR_INVALID;
    Scratch s;
    float* part = stats ? s.get((size_t)conv_pool_stat_capacity() * 128) : nullptr;
    if (stats && !part) return SELD_ERR_NOMEM;
    int np = 0;
    if (launch_conv_first_fwd_pool(0, x, w, bias, gamma, z, zext, amax, part, &np, B, H, Cin, g_conv1_split_bf16)) return SELD_ERR_UNSUPPORTED;
    if (stats) launch_reduce_slabs(0, part, np, 128, stats, 128, 0);
    return done();
}

int seld_k_bn_relu_ext(const float* zext, const float* scale, const float* shift, float* p, int64_t n) {
    if (!zext || !scale || !shift || !p) return SELD_ERR_INVALID;
    if (launch_bn_relu_ext(0, zext, scale, shift, p, n)) return SELD_ERR_UNSUPPORTED;
    return done();
}

int seld_k_conv3x3_dgrad(const float* dz, const float* w, float* dx, int B, int H, int W, int Cin, int Cout) {
    if (!dz || !w || !dx || Cin != 64 || Cout != 64) return SELD_ERR_UNSUPPORTED;
    Scratch s;
    float* wt = s.get(9 * 4096);
    if (!wt) return SELD_ERR_NOMEM;
    launch_flip_weights(0, w, wt);
    if (g_conv64_split_bf16) {
        unsigned short* wsp = reinterpret_cast<unsigned short*>(s.get(9 * 3 * 4096 / 2 + 16));
        if (!wsp) return SELD_ERR_NOMEM;
        // the model's own route: the flipped planes straight from w (prep.h split_weights_body, flip = 1) and the input-gradient launcher, which takes
        // the four-product form under option "bwd_four_products"
        const float* ws[1] = {w}; unsigned short* ds[1] = {wsp}; const int fl[1] = {1};
        launch_split_weights_batch(0, 1, ws, ds, fl);
        launch_conv64_dgrad_sb(0, dz, wsp, dx, B, H, W);
    } else
        launch_conv64_fwd(0, dz, wt, nullptr, dx, nullptr, nullptr, B, H, W);
    return done();
}

int seld_k_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* db, int B, int H, int W, int Cin, int Cout) {
    if (!x || !dz || !dw || !db || Cout != 64) return SELD_ERR_UNSUPPORTED;
    Scratch s;
    float* slab = s.get((size_t)conv_wgrad_slab_capacity() * (9 * 4096 + 64));
    float* tmp = s.get(9 * 4096 + 64);
    if (!slab || !tmp) return SELD_ERR_NOMEM;
    int ns = 0;
    if (Cin == 64) {
        if (g_conv64_split_bf16 && conv64_wgrad_sb_usable(W)) {
            if (launch_conv64_wgrad_sb(0, x, dz, slab, &ns, B, H, W)) return SELD_ERR_UNSUPPORTED;
        } else if (launch_conv64_wgrad(0, x, dz, slab, &ns, B, H, W)) return SELD_ERR_UNSUPPORTED;
        launch_reduce_slabs(0, slab, ns, 9 * 4096 + 64, tmp, 9 * 4096 + 64, 0);
        hipMemcpyAsync(dw, tmp, 9 * 4096 * 4, hipMemcpyDeviceToDevice, 0);
        hipMemcpyAsync(db, tmp + 9 * 4096, 64 * 4, hipMemcpyDeviceToDevice, 0);
    } else if (W == 64) {
        if (launch_conv_first_wgrad(0, x, dz, slab, &ns, B, H, Cin)) return SELD_ERR_UNSUPPORTED;
        launch_reduce_slabs(0, slab, ns, conv_first_wgrad_slab_stride(Cin), tmp, (int64_t)(9 * Cin + 1) * 64, 0);
        hipMemcpyAsync(dw, tmp, (size_t)9 * Cin * 64 * 4, hipMemcpyDeviceToDevice, 0);
        hipMemcpyAsync(db, tmp + 9 * Cin * 64, 64 * 4, hipMemcpyDeviceToDevice, 0);
    } else return SELD_ERR_UNSUPPORTED;
    return done();
}

int seld_k_conv1_bwd_fused(const float* x, const float* z, const float* dp, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, float* dw, float* db, float* dgamma, float* dbeta,
                           int B, int H, int Cin, int pt, int pf) {
    if (!x || !z || !dp || !mean || !invstd || !gamma || !beta || !dw || !db || !dgamma || !dbeta) return SELD_ERR_INVALID;
    const int W = 64, C = 64;
    if (H % pt || W % pf) return SELD_ERR_UNSUPPORTED;
    Scratch s;
    float* coef = s.get(64 * 6);   // mean | invstd | scale | shift | c1 | c2
    float* part = s.get((size_t)bn_partial_capacity() * 128);
    float* pbuf = s.get((size_t)B * (H / pt) * (W / pf) * 64);
    float* slab = s.get((size_t)conv_wgrad_slab_capacity() * conv_first_wgrad_slab_stride(Cin));
    float* tmp = s.get(conv_first_wgrad_slab_stride(Cin));
    unsigned char* amax = reinterpret_cast<unsigned char*>(s.get(((size_t)B * (H / pt) * (W / pf) * 64 + 3) / 4));
    if (!coef || !part || !pbuf || !slab || !tmp || !amax) return SELD_ERR_NOMEM;
    hipMemcpyAsync(coef, mean, 256, hipMemcpyDeviceToDevice, 0);
    hipMemcpyAsync(coef + 64, invstd, 256, hipMemcpyDeviceToDevice, 0);
    hipLaunchKernelGGL(coeffs_kernel, dim3(1), dim3(64), 0, 0, mean, invstd, gamma, beta, coef + 128, coef + 192, C);
    if (launch_bn_relu_pool_fwd(0, z, coef + 128, coef + 192, pbuf, B, H, W, C, pt, pf)) return SELD_ERR_UNSUPPORTED;
    int np = 0, ns = 0;
    if (launch_bn_pool_bwd_reduce(0, z, pbuf, dp, coef, coef + 64, coef + 128, coef + 192, part, &np, B, H, W, C, pt, pf)) return SELD_ERR_UNSUPPORTED;
    launch_bn_bwd_finalize(0, part, np, (double)B * H * W, dgamma, dbeta, coef + 256, C);
    if (launch_pool_argext(0, z, gamma, amax, B, H, W, pt, pf)) return SELD_ERR_UNSUPPORTED;
    if (launch_conv_first_wgrad_fused(0, x, z, pbuf, dp, amax, coef, slab, &ns, B, H, Cin, pt, pf)) return SELD_ERR_UNSUPPORTED;
    launch_reduce_slabs(0, slab, ns, conv_first_wgrad_slab_stride(Cin), tmp, (int64_t)(9 * Cin + 1) * 64, 0);
    hipMemcpyAsync(dw, tmp, (size_t)9 * Cin * 64 * 4, hipMemcpyDeviceToDevice, 0);
    hipMemcpyAsync(db, tmp + 9 * Cin * 64, 64 * 4, hipMemcpyDeviceToDevice, 0);
    return done();
}

int seld_k_conv1_gram(const float* x, float* G, int B, int H, int Cin) {
    if (!x || !G) return SELD_ERR_INVALID;
    if (Cin != 7 && Cin != 10) return SELD_ERR_UNSUPPORTED;
    const size_t kp = (size_t)conv_gram_dim(Cin);
    Scratch s;
    float* slab = s.get((size_t)conv_gram_slab_capacity() * kp * kp);
    if (!slab) return SELD_ERR_NOMEM;
    int ns = 0;
    if (launch_conv_first_gram(0, x, slab, &ns, B, H, Cin)) return SELD_ERR_UNSUPPORTED;
    launch_reduce_slabs(0, slab, ns, (int64_t)(kp * kp), G, (int64_t)(kp * kp), 0);
    return done();
}

int seld_k_conv1_train_gram(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                            const float* dp, float* p, float* dw, float* db, float* dgamma, float* dbeta, int B, int H, int Cin) {
    if (!x || !w || !bias || !gamma || !beta || !dp || !p || !dw || !db || !dgamma || !dbeta) return SELD_ERR_INVALID;
    if ((Cin != 7 && Cin != 10) || H % 5) return SELD_ERR_UNSUPPORTED;
    const int W = 64, C = 64;
    const size_t np = (size_t)B * (H / 5) * 16 * 64, kp = (size_t)conv_gram_dim(Cin);
    Scratch s;
    float* coef = s.get(64 * 6);          // mean | invstd | scale | shift | c1 | c2
    float* mov = s.get(128);
    float* part = s.get((size_t)conv_pool_stat_capacity() * 128);
    float* bpart = s.get((size_t)bn_partial_capacity() * 128);
    float* zext = s.get(np);
    unsigned char* amax = reinterpret_cast<unsigned char*>(s.get((np + 3) / 4));
    float* gslab = s.get((size_t)conv_gram_slab_capacity() * kp * kp);
    float* gm = s.get(kp * kp);
    float* mslab = s.get((size_t)conv_msparse_slab_capacity() * kp * 64);
    float* mm = s.get(kp * 64);
    float* out = s.get((size_t)(9 * Cin + 1) * 64);
    if (!coef || !mov || !part || !bpart || !zext || !amax || !gslab || !gm || !mslab || !mm || !out) return SELD_ERR_NOMEM;
    hipMemsetAsync(mov, 0, 128 * sizeof(float), 0);
    int npart = 0, nb = 0, ns = 0;
    // forward without z: window extremes + positions + statistics; then BN coefficients and the pooled activation
    if (launch_conv_first_fwd_pool(0, x, w, bias, gamma, nullptr, zext, amax, part, &npart, B, H, Cin, g_conv1_split_bf16)) return SELD_ERR_UNSUPPORTED;
    launch_bn_finalize(0, part, npart, (double)B * H * W, gamma, beta, mov, mov + 64, coef, coef + 64, coef + 128, coef + 192, C, 1);
    launch_bn_relu_ext(0, zext, coef + 128, coef + 192, p, (int64_t)np);
    // backward: BN sums from the pooled tensors, then dW = ka (G W + g b) + g kb + M
    if (launch_bn_pool_bwd_reduce(0, zext, p, dp, coef, coef + 64, coef + 128, coef + 192, bpart, &nb, B, H, W, C, 5, 4, 1))
        return SELD_ERR_UNSUPPORTED;
    launch_bn_bwd_finalize(0, bpart, nb, (double)B * H * W, dgamma, dbeta, coef + 256, C);
    if (launch_conv_first_gram(0, x, gslab, &ns, B, H, Cin)) return SELD_ERR_UNSUPPORTED;
    launch_reduce_slabs(0, gslab, ns, (int64_t)(kp * kp), gm, (int64_t)(kp * kp), 0);
    if (launch_conv_first_msparse(0, x, p, dp, amax, coef + 128, mslab, &ns, B, H, Cin)) return SELD_ERR_UNSUPPORTED;
    launch_reduce_slabs(0, mslab, ns, (int64_t)(kp * 64), mm, (int64_t)(kp * 64), 0);
    launch_conv_first_assemble(0, gm, mm, w, bias, coef, out, out + 9 * Cin * 64, Cin);
    hipMemcpyAsync(dw, out, (size_t)9 * Cin * 64 * 4, hipMemcpyDeviceToDevice, 0);
    hipMemcpyAsync(db, out + 9 * Cin * 64, 64 * 4, hipMemcpyDeviceToDevice, 0);
    return done();
}

int seld_k_bn_relu_pool_fwd(const float* z, const float* scale, const float* shift, float* p, int B, int H, int W, int C,
                            int pt, int pf) {
    if (!z || !scale || !shift || !p) return SELD_ERR_INVALID;
    if (launch_bn_relu_pool_fwd(0, z, scale, shift, p, B, H, W, C, pt, pf)) return SELD_ERR_UNSUPPORTED;
    return done();
}

int seld_k_bn_relu_pool_bwd(const float* z, const float* dp, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, float* dz, float* dgamma, float* dbeta, int B, int H, int W, int C, int pt,
                            int pf) {
    if (!z || !dp || !mean || !invstd || !gamma || !beta || !dz || !dgamma || !dbeta) return SELD_ERR_INVALID;
    if (C != 64) return SELD_ERR_UNSUPPORTED;
    Scratch s;
    float* coef = s.get(64 * 4);
    float* part = s.get((size_t)bn_partial_capacity() * 128);
    float* pbuf = s.get((size_t)B * (H / pt) * (W / pf) * 64);
    if (!coef || !part || !pbuf) return SELD_ERR_NOMEM;
    float *scale = coef, *shift = coef + 64, *c1c2 = coef + 128;
    hipLaunchKernelGGL(coeffs_kernel, dim3(1), dim3(64), 0, 0, mean, invstd, gamma, beta, scale, shift, C);
    int np = 0;
    if (launch_bn_relu_pool_fwd(0, z, scale, shift, pbuf, B, H, W, C, pt, pf)) return SELD_ERR_UNSUPPORTED;
    if (launch_bn_pool_bwd_reduce(0, z, pbuf, dp, mean, invstd, scale, shift, part, &np, B, H, W, C, pt, pf)) return SELD_ERR_UNSUPPORTED;
    launch_bn_bwd_finalize(0, part, np, (double)B * H * W, dgamma, dbeta, c1c2, C);
    launch_bn_pool_bwd_dz(0, z, dp, mean, invstd, scale, shift, c1c2, dz, B, H, W, C, pt, pf);
    return done();
}

int seld_k_gemm(const float* A, const float* Bm, const float* bias, float* C, int M, int N, int K, int transb, int act,
                int accumulate) {
    if (!A || !Bm || !C) return SELD_ERR_INVALID;
    if (launch_gemm(0, A, K, Bm, transb ? K : N, bias, C, N, M, N, K, transb, act, accumulate)) return SELD_ERR_INVALID;
    return done();
}

int seld_k_gemm_pair_n(const float* A, const float* B0, const float* B1, const float* bias0, const float* bias1, float* C0,
                       float* C1, int M, int N, int K, int transb, int act) {
    if (!A || !B0 || !B1 || !C0 || !C1) return SELD_ERR_INVALID;
    if (launch_gemm_dual_n(0, A, K, B0, B1, transb ? K : N, bias0, bias1, C0, C1, N, M, N, K, transb, act)) return SELD_ERR_INVALID;
    return done();
}

int seld_k_gemm_pair_k(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias, float* C, int M,
                       int N, int K, int transb, int act, int accumulate) {
    if (!A0 || !A1 || !B0 || !B1 || !C) return SELD_ERR_INVALID;
    if (launch_gemm_dual_k(0, A0, A1, K, B0, B1, transb ? K : N, bias, C, N, M, N, K, transb, act, accumulate))
        return SELD_ERR_INVALID;
    return done();
}

int seld_k_gemm_sb(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias0, const float* bias1,
                   float* C0, float* C1, int M, int N, int K, int transb, int act, int mode) {
    if (!A0 || !B0 || !C0 || mode < 0 || mode > 2 || (mode && !B1) || (mode == 1 && !C1) || (mode == 2 && !A1)) return SELD_ERR_INVALID;
    if (M <= 0 || N <= 0 || K <= 0 || !gemm_sb_usable(A0, K, N, K) || (mode == 2 && !gemm_sb_usable(A1, K, N, K))) return SELD_ERR_INVALID;
    Scratch s;
    const size_t ne = gemm_sb_split_elems(K, N);
    unsigned short* sp = reinterpret_cast<unsigned short*>(s.get(ne));   // 2 operands x ne bf16 = ne floats
    if (!sp) return SELD_ERR_NOMEM;
    const float* src[2] = {B0, B1};
    unsigned short* dst[2] = {sp, sp + ne};
    const int ldb[2] = {transb ? K : N, transb ? K : N}, tb[2] = {transb, transb}, Ks[2] = {K, K}, Ns[2] = {N, N};
    if (launch_gemm_split_b(0, mode ? 2 : 1, src, dst, ldb, tb, Ks, Ns)) return SELD_ERR_INVALID;
    if (launch_gemm_sb(0, A0, A1, K, dst[0], dst[1], bias0, bias1, C0, C1, N, M, N, K, act, mode)) return SELD_ERR_INVALID;
    return done();
}

int seld_k_xc_dw_bwd(const float* dy, const float* k, const float* xin, const float* add, const float* aff, const float* bn_mean,
                     const float* bn_invstd, float* dx, float* dk, float* sums, int B, int H, int fused) {
    if (!dy || !k || !xin || !dx || !dk || B < 1 || H < 1) return SELD_ERR_INVALID;
    const bool want_sums = bn_mean != nullptr;
    if (want_sums && (!bn_invstd || !sums || !aff || add)) return SELD_ERR_INVALID;      // the sums belong to the BatchNormalization folded into the loads
    const int64_t npix = (int64_t)B * H * 16;
    Scratch s;
    int ns = 0, np = 0;
    if (fused) {
        const int nb = xc_dw_fused_slabs(B, H);
        float* slab = s.get((size_t)nb * 576);
        float* tmp = s.get((size_t)reduce_slabs_groups(nb) * 576);
        float* part = s.get((size_t)nb * 128);
        float* folded = s.get((size_t)xc_partial_capacity() * 128);
        if (!slab || !tmp || !part || !folded) return SELD_ERR_NOMEM;
        if (launch_dw3x3_bwd_fused(0, dy, k, xin, add, dx, slab, &ns, B, H, 16, aff, bn_mean, bn_invstd, want_sums ? part : nullptr)) return SELD_ERR_UNSUPPORTED;
        launch_reduce_slabs_2stage(0, slab, ns, 576, dk, 576, tmp);
        if (want_sums) {
            launch_xc_fold_partials(0, part, ns, folded, &np);
            launch_reduce_slabs(0, folded, np, 128, sums, 128, 0);
        }
    } else {
        float* slab = s.get((size_t)xc_partial_capacity() * 576);
        float* part = s.get((size_t)xc_partial_capacity() * 128);
        if (!slab || !part) return SELD_ERR_NOMEM;
        launch_dw3x3_bwd_data(0, dy, k, xin, add, dx, B, H, 16, aff);
        launch_dw3x3_bwd_w(0, xin, dy, slab, &ns, B, H, 16, aff);
        launch_reduce_slabs(0, slab, ns, 576, dk, 576, 0);
        if (want_sums) {
            launch_xc_bn_bwd_reduce(0, xin, dx, bn_mean, bn_invstd, part, &np, npix);
            launch_reduce_slabs(0, part, np, 128, sums, 128, 0);
        }
    }
    return done();
}

int seld_k_gemm_tn(const float* A, const float* Bm, float* C, float* colsum, int M, int K1, int N) {
    if (!A || !Bm || !C) return SELD_ERR_INVALID;
    Scratch s;
    float* slab = s.get((size_t)gemm_tn_max_splits() * ((size_t)K1 * N + N));
    if (!slab) return SELD_ERR_NOMEM;
    int ns = 0;
    if (g_gemm_tn_sb && gemm_tn_sb_usable(A, K1, Bm, N, K1, N)) {
        if (launch_gemm_tn_sb(0, A, K1, Bm, N, slab, &ns, M, N, 0, 0, colsum ? 1 : 0)) return SELD_ERR_INVALID;
    } else if (launch_gemm_tn(0, A, K1, Bm, N, slab, &ns, M, K1, N, 0, 0, colsum ? 1 : 0)) return SELD_ERR_INVALID;
    if (colsum) launch_reduce_slabs2(0, slab, ns, (int64_t)K1 * N + N, C, (int64_t)K1 * N, colsum, N);
    else launch_reduce_slabs(0, slab, ns, (int64_t)K1 * N + N, C, (int64_t)K1 * N, 0);
    return done();
}

int seld_k_gru_fwd(const float* gx_f, const float* gx_b, const float* U_f, const float* U_b, const float* brec_f,
                   const float* brec_b, float* h_f, float* h_b, float* saved_f, float* saved_b, float* out, int B, int S,
                   int units) {
    if (units != 128) return SELD_ERR_UNSUPPORTED;
    if (!gx_f || !gx_b || !U_f || !U_b || !brec_f || !brec_b || !h_f || !h_b) return SELD_ERR_INVALID;
    launch_gru_fwd(0, gx_f, gx_b, U_f, U_b, brec_f, brec_b, h_f, h_b, saved_f, saved_b, B, S);
    if (out) launch_mul(0, h_f, h_b, out, (int64_t)B * S * 128);
    return done();
}

int seld_k_gru_bwd(const float* dout, const float* h_f, const float* h_b, const float* saved_f, const float* saved_b,
                   const float* U_f, const float* U_b, float* dgx_f, float* dgx_b, float* dgh_f, float* dgh_b, int B, int S,
                   int units) {
    if (units != 128) return SELD_ERR_UNSUPPORTED;
    if (!dout || !h_f || !h_b || !saved_f || !saved_b || !U_f || !U_b || !dgx_f || !dgx_b || !dgh_f || !dgh_b) return SELD_ERR_INVALID;
    launch_gru_bwd(0, dout, h_f, h_b, saved_f, saved_b, U_f, U_b, dgx_f, dgx_b, dgh_f, dgh_b, B, S);
    return done();
}

int seld_k_losses(const float* sed, const float* doa, const float* y_sed, const float* y_doa, const seld_loss_cfg* cfg,
                  float* sloss, float* dloss, float* dsed_pre, float* ddoa_pre, int B, int S, int nc) {
    if (!sed || !doa || !y_sed || !y_doa || !cfg || !sloss || !dloss) return SELD_ERR_INVALID;
    Scratch s;
    const int rows = B * S;
    float* scr = s.get((size_t)loss_scratch_floats(rows));
    float* den = s.get(4);
    if (!scr || !den) return SELD_ERR_NOMEM;
    if (cfg->doa_loss == SELD_DOA_MMSE) {
        if (cfg->mmse_den > 0.f) hipMemcpy(den, &cfg->mmse_den, 4, hipMemcpyHostToDevice);
        else launch_mmse_den(0, y_doa, den, scr, rows, nc);
    }
    launch_losses(0, sed, doa, y_sed, y_doa, cfg->doa_loss, cfg->w_sed, cfg->w_doa, cfg->sed_grad_scale, den, sloss, dloss,
                  dsed_pre, ddoa_pre, scr, B, S, nc);
    return done();
}

int seld_k_adam(float* theta, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                int64_t step) {
    if (!theta || !g || !m || !v || n <= 0 || step < 1) return SELD_ERR_INVALID;
    const double t = (double)step;
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    launch_adam(0, theta, g, m, v, n, lr_t, beta1, beta2, eps);
    return done();
}

// VALU-only load on `blocks` CUs (the recurrence's shape); lane 0 of every block reports shader cycles and 100 MHz ticks
__global__ __launch_bounds__(512) void valu_clock_kernel(unsigned long long* out, int iters, float seed) {
    float a = seed + threadIdx.x * 1e-9f, b = seed * 0.5f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a = fmaf(a, 1.0000001f, 1e-9f);
        b = fmaf(b, 0.9999999f, 1e-9f);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x] = c1 - c0;
        out[3 * blockIdx.x + 1] = r1 - r0;
        out[3 * blockIdx.x + 2] = (unsigned long long)__float_as_uint(a + b);   // keeps the loop alive
    }
}

int seld_k_valu_clock_mhz(int blocks, double* mhz) {
    if (!mhz || blocks < 1 || blocks > 1024) return SELD_ERR_INVALID;
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, (size_t)blocks * 3 * sizeof(unsigned long long)) != hipSuccess) return SELD_ERR_NOMEM;
    std::vector<unsigned long long> h((size_t)blocks * 3);
    for (int rep = 0; rep < 2; ++rep)      // the second launch is measured (clocks settled)
        hipLaunchKernelGGL(valu_clock_kernel, dim3(blocks), dim3(512), 0, 0, d, 200000, 1.0f);
    const bool ok = hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess;
    hipFree(d);
    if (!ok) return SELD_ERR_HIP;
    std::vector<double> r;
    for (int i = 0; i < blocks; ++i)
        if (h[3 * i + 1]) r.push_back(100.0 * (double)h[3 * i] / (double)h[3 * i + 1]);
    if (r.empty()) return SELD_ERR_HIP;
    std::sort(r.begin(), r.end());
    *mhz = r[r.size() / 2];
    return SELD_OK;
}

// ---- resnet50_block pieces (resnet.hip + the fp32 GEMMs), as api.hip composes them
// split `w` ([K,N]; transposed: the planes of w^T) into scratch planes when the split-bf16 product takes the shape, as the model does per step
// transposed == 2: the matrix of a 3x3 kernel's input-gradient convolution (K = 9 Cin: flipped taps, channels swapped)
static unsigned short* rn_k_split(Scratch& s, const float* w, int K, int N, int transposed) {
    if (!g_rn_split_bf16 || !(transposed ? rn_sb_dgrad_ok(K, N) : rn_sb_fwd_ok(K, N))) return nullptr;
    unsigned short* d = reinterpret_cast<unsigned short*>(s.get((gemm_sb_split_elems(K, N) + 1) / 2));
    if (!d) return nullptr;
    const float* src[1] = {w}; unsigned short* dst[1] = {d};
    const int ldb[1] = {N}, tb[1] = {transposed}, Ks[1] = {transposed == 2 ? 9 * N : transposed ? N : K}, Ns[1] = {transposed == 2 ? K / 9 : transposed ? K : N};
    return launch_gemm_split_b(0, 1, src, dst, ldb, tb, Ks, Ns) ? nullptr : d;
}

int seld_k_rn_conv(const float* x, const float* w, float* z, int B, int H, int W, int Cin, int Cout, int ksize, int stride_f) {
    if (!x || !w || !z) return SELD_ERR_INVALID;
    if ((ksize != 1 && ksize != 3) || (ksize == 3 && stride_f != 1) || stride_f < 1 || W % stride_f || Cin % 4) return SELD_ERR_UNSUPPORTED;
    const int Wo = W / stride_f;
    const int M = B * H * Wo, K = ksize * ksize * Cin;
    Scratch s;
    const unsigned short* wsp = rn_k_split(s, w, K, Cout, 0);
    if (ksize == 1) {
        if (launch_rn_product_fwd(0, x, Cin * stride_f, w, wsp, z, M, K, Cout)) return SELD_ERR_INVALID;
        return done();
    }
    if (wsp && rn_conv3_sb_ok(Cin, Cout)) {      // im2col rows formed on load
        if (launch_rn_conv3_fwd(0, x, wsp, z, B, H, W, Cin, Cout)) return SELD_ERR_INVALID;
        return done();
    }
    float* col = s.get((size_t)M * 9 * Cin);
    if (!col) return SELD_ERR_NOMEM;
    launch_im2col3x3(0, x, col, B, H, W, Cin);
    if (launch_rn_product_fwd(0, col, K, w, wsp, z, M, K, Cout)) return SELD_ERR_INVALID;
    return done();
}

int seld_k_rn_conv_bwd(const float* x, const float* w, const float* dz, float* dw, float* dx, int B, int H, int W, int Cin, int Cout,
                       int ksize, int stride_f) {
    if (!x || !w || !dz || !dw || !dx) return SELD_ERR_INVALID;
    if ((ksize != 1 && ksize != 3) || (ksize == 3 && stride_f != 1) || stride_f < 1 || W % stride_f || Cin % 4) return SELD_ERR_UNSUPPORTED;
    const int Wo = W / stride_f, M = B * H * Wo, K1 = ksize * ksize * Cin;
    Scratch s;
    const int64_t cap = tn_slab_capacity();     // the model's slab buffer
    float* slab = s.get((size_t)cap);
    if (!slab) return SELD_ERR_NOMEM;
    if (ksize == 3 && g_rn_split_bf16 && rn_conv3_sb_ok(Cin, Cout)) {      // both gradients with the im2col rows formed on load
        const unsigned short* wsp_f = rn_k_split(s, w, K1, Cout, 2);
        if (!wsp_f) return SELD_ERR_NOMEM;
        if (launch_rn_conv3_wgrad(0, x, dz, slab, cap, dw, B, H, W, Cin, Cout)) return SELD_ERR_INVALID;
        if (launch_rn_conv3_dgrad(0, dz, wsp_f, dx, B, H, W, Cin, Cout)) return SELD_ERR_INVALID;
        return done();
    }
    float* col = ksize == 3 ? s.get((size_t)M * 9 * Cin) : nullptr;
    float* dcol = ksize == 3 ? s.get((size_t)M * 9 * Cin) : nullptr;
    if (ksize == 3 && (!col || !dcol)) return SELD_ERR_NOMEM;
    const unsigned short* wsp_t = rn_k_split(s, w, K1, Cout, 1);
    if (ksize == 3) {
        launch_im2col3x3(0, x, col, B, H, W, Cin);
        if (launch_rn_product_wgrad(0, col, K1, dz, slab, cap, dw, M, K1, Cout, g_rn_split_bf16)) return SELD_ERR_INVALID;
        if (launch_rn_product_dgrad(0, dz, w, wsp_t, dcol, K1, M, K1, Cout, 0)) return SELD_ERR_INVALID;
        launch_col2im3x3(0, dcol, dx, B, H, W, Cin);
    } else {
        const int ldx = Cin * stride_f;
        if (launch_rn_product_wgrad(0, x, ldx, dz, slab, cap, dw, M, K1, Cout, g_rn_split_bf16)) return SELD_ERR_INVALID;
        if (stride_f > 1 && hipMemsetAsync(dx, 0, (size_t)B * H * W * Cin * sizeof(float), 0) != hipSuccess) return SELD_ERR_HIP;
        if (launch_rn_product_dgrad(0, dz, w, wsp_t, dx, ldx, M, K1, Cout, 0)) return SELD_ERR_INVALID;
    }
    return done();
}

int seld_k_rn_bn(const float* z, const float* gamma, const float* beta, const float* res, float* out, float* mean, float* invstd,
                 int64_t npix, int C, int relu) {
    if (!z || !gamma || !beta || !out) return SELD_ERR_INVALID;
    if (C % 32 || npix <= 0) return SELD_ERR_UNSUPPORTED;
    Scratch s;
    float* part = s.get((size_t)rn_partial_capacity() * ((C + 63) / 64) * 128);
    float* coef = s.get((size_t)6 * C);
    float* mov = s.get((size_t)2 * C);
    if (!part || !coef || !mov) return SELD_ERR_NOMEM;
    if (hipMemsetAsync(mov, 0, (size_t)2 * C * sizeof(float), 0) != hipSuccess) return SELD_ERR_HIP;
    int nbx = 0;
    if (launch_rn_bn_stats(0, z, part, &nbx, npix, C)) return SELD_ERR_UNSUPPORTED;
    launch_rn_bn_finalize(0, part, nbx, (double)npix, gamma, beta, mov, mov + C, coef, C, 1);
    launch_rn_bn_apply(0, z, coef, res, out, npix, C, relu);
    if (mean && hipMemcpyAsync(mean, coef, C * sizeof(float), hipMemcpyDeviceToDevice, 0) != hipSuccess) return SELD_ERR_HIP;
    if (invstd && hipMemcpyAsync(invstd, coef + C, C * sizeof(float), hipMemcpyDeviceToDevice, 0) != hipSuccess) return SELD_ERR_HIP;
    return done();
}

int seld_k_rn_bn_bwd(const float* z, const float* dy, const float* mask, const float* gamma, float* dz, float* dgamma, float* dbeta,
                     int64_t npix, int C) {
    if (!z || !dy || !gamma || !dz || !dgamma || !dbeta) return SELD_ERR_INVALID;
    if (C % 32 || npix <= 0) return SELD_ERR_UNSUPPORTED;
    Scratch s;
    float* part = s.get((size_t)rn_partial_capacity() * ((C + 63) / 64) * 128);
    float* coef = s.get((size_t)6 * C);
    float* mov = s.get((size_t)3 * C);
    if (!part || !coef || !mov) return SELD_ERR_NOMEM;
    if (hipMemsetAsync(mov, 0, (size_t)3 * C * sizeof(float), 0) != hipSuccess) return SELD_ERR_HIP;
    int nbx = 0;
    launch_rn_bn_stats(0, z, part, &nbx, npix, C);
    launch_rn_bn_finalize(0, part, nbx, (double)npix, gamma, mov + 2 * C, mov, mov + C, coef, C, 1);     // beta plays no part in the backward
    launch_rn_bn_bwd_reduce(0, z, dy, mask, coef, part, &nbx, npix, C);
    launch_rn_bn_bwd_finalize(0, part, nbx, (double)npix, dgamma, dbeta, coef, C);
    launch_rn_bn_bwd_dz(0, z, dy, mask, coef, dz, npix, C);
    return done();
}

int seld_k_gru_timing(int which, unsigned long long* cycles, int blocks) {
    if (!cycles) return SELD_ERR_INVALID;
    const int rc = gru_timing_read(which, cycles, blocks);
    return rc == 0 ? SELD_OK : (rc == -2 ? SELD_ERR_UNSUPPORTED : SELD_ERR_INVALID);
}

int seld_device_clocks(int device, int* compute_units, int* clock_khz, int* mem_clock_khz, int* mem_bus_bits) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return SELD_ERR_HIP;
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (mem_clock_khz) *mem_clock_khz = p.memoryClockRate;
    if (mem_bus_bits) *mem_bus_bits = p.memoryBusWidth;
    return SELD_OK;
}

}  // extern "C"
