// common.h — shared device helpers for the gfx950 SELDnet kernels (wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: exact fp32 (k-ordered fmaf chain), 64 FLOP/clk/SIMD.
// A: lane l holds A[i = l&31][k = l>>5];  B: lane l holds B[k = l>>5][j = l&31].
// C/D: reg r of lane l is C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
#define MFMA_F32_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// ---- MFMA 32x32 accumulator tile -> float4 stores.  A lane of the accumulator layout holds ONE output channel (column) of 16
// pixels; stored as it stands that is 16 dword stores per tile, and a conv epilogue of 64 - 128 dword stores per wave was measured at
// a quarter of the kernel (tools/tune_conv64.py: the same bytes as dwordx4 stores cost an eighth of that).  A 4 x 4 transpose inside
// each quad of lanes (two DPP butterfly stages) turns registers 4q..4q+3 (pixels 8q + 4hi + 0..3, channel li) into pixel 8q + 4hi +
// (li & 3), channels 4 (li >> 2) .. + 3: one float4.
__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
}
__device__ __forceinline__ float4 quad_transpose4(float x0, float x1, float x2, float x3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float r;
    r = dpp_quad_xor1(b0 ? x0 : x1); if (b0) x0 = r; else x1 = r;
    r = dpp_quad_xor1(b0 ? x2 : x3); if (b0) x2 = r; else x3 = r;
    r = dpp_quad_xor2(b1 ? x0 : x2); if (b1) x0 = r; else x2 = r;
    r = dpp_quad_xor2(b1 ? x1 : x3); if (b1) x1 = r; else x3 = r;
    return make_float4(x0, x1, x2, x3);
}


// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt (it is a
// workgroup-scope fence for global memory), which would serialise the prefetched global loads and the
// epilogue stores of the pipelined conv kernels behind every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define SELD_BN_EPS 1e-3f
#define SELD_BN_MOMENTUM 0.99f

// ---- per-step weight pre-passes: job descriptors (kernel arguments by value) and the merged launch (prep.hip) ----------------
#define GSB_MAX_JOBS 16
// bf16 single-product mode (SELD_DTYPE_BF16 / option "bf16_single"): the split-bf16 kernels that implement it take ONE bf16 MFMA product per
// fp32 product, operands rounded to nearest-even bf16 (fp32 accumulation), instead of the six products of the exact 3-way split.  Set by
// api.hip from the ctx before it enqueues a pass (process-wide: contexts of different modes must not enqueue concurrently from different
// host threads); kernels without a single-product form keep the exact six products: the weight pre-split writes ALL three planes in both
// modes (plane 0 = the rounded value in this mode, planes 1-2 the exact split of the rest, prep.h), so they compute with exact weights.
extern int g_mfma_one;
// round-to-nearest-even bf16 of an fp32 value, as its 16 high bits (NaN stays a quiet NaN)
__host__ __device__ __forceinline__ unsigned bf16_rne_bits(float x) {
    unsigned u;
    __builtin_memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // NaN stays NaN (quiet): the carry below would turn 0x7fff.... into -0
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
struct GemmSplitJobs {
    int njobs;
    int one;       // 1: plane 0 = round-to-nearest bf16 (planes 1, 2 = the exact split of x - plane 0, for consumers without a ONE form)
    const float* src[GSB_MAX_JOBS];
    unsigned short* dst[GSB_MAX_JOBS];
    int ldb[GSB_MAX_JOBS], transb[GSB_MAX_JOBS], K[GSB_MAX_JOBS], N[GSB_MAX_JOBS];
};
struct SplitWeightJobs { const float* w[8]; unsigned short* dst[8]; int flip[8]; int one; };
struct HeadsLin { const float *w1[2], *b1[2], *w2[2], *b2[2]; int n[2]; int K, Hd; };
// one launch for all three kinds (any of them may be empty: na / nb = 0, weff = nullptr)
int launch_weight_prep(hipStream_t st, const GemmSplitJobs& a, int na, const SplitWeightJobs& b, int nb, const HeadsLin& h, float* weff);

// ---- launcher prototypes (implemented in the .hip files; used by api.hip) -------------------
struct ConvFirstArgs { const float* x; const float* w; const float* bias; float* z; float* stat_partial; int B, H; };

int launch_conv_first_fwd(hipStream_t st, const float* x, const float* w, const float* bias, float* z,
                          float* stat_partial, int* n_partial, int B, int H, int Cin);
int launch_conv_first_fwd_pool(hipStream_t st, const float* x, const float* w, const float* bias, const float* gamma,
                               float* z, float* zext, unsigned char* amax, float* stat_partial, int* n_partial, int B, int H,
                               int Cin, int split_bf16 = 1);
int launch_conv_first_fwd_pool_sb(hipStream_t st, const float* x, const float* w, const float* bias, const float* gamma,
                                  float* zext, unsigned char* amax, float* stat_partial, int* n_partial, int B, int H, int Cin);
int conv_pool_sb_stat_capacity();
int launch_pool_argext(hipStream_t st, const float* z, const float* gamma, unsigned char* amax, int B, int H, int W, int pt,
                       int pf);
int conv_pool_stat_capacity();
int launch_bn_relu_ext(hipStream_t st, const float* zext, const float* scale, const float* shift, float* p, int64_t n);
int launch_conv64_fwd(hipStream_t st, const float* x, const float* w9, const float* bias, float* z,
                      float* stat_partial, int* n_partial, int B, int H, int W);
int conv_stat_partial_capacity();  // max blocks writing stat partials
int launch_conv_first_wgrad(hipStream_t st, const float* x, const float* dz, float* slab, int* n_slab,
                            int B, int H, int Cin);
// fused BN/ReLU/pool backward + first-layer wgrad; coef = [mean|invstd|scale|shift|c1|c2] x 64 (contiguous)
int launch_conv_first_wgrad_fused(hipStream_t st, const float* x, const float* z, const float* p, const float* dp,
                                  const unsigned char* amax, const float* coef, float* slab, int* n_slab, int B, int H,
                                  int Cin, int pt, int pf);
int conv_gram_dim(int Cin);
extern int g_gram_bg_blocks;
extern int g_sbd_dgrad_r8;    // conv_sb.hip (experiment)
extern int g_xc_w16;          // xception.hip: row-per-workgroup depthwise kernels for W = 16
extern int g_xc_xcd_map;      // xception.hip: XCD-contiguous row ranges in the depthwise kernels
int conv_gram_slab_capacity();
int conv_msparse_slab_capacity();
int launch_conv_first_gram(hipStream_t st, const float* x, float* slab, int* n_slab, int B, int H, int Cin, int background = 0, int part = 0, int nparts = 1);
int launch_conv_first_msparse(hipStream_t st, const float* x, const float* p, const float* dp, const unsigned char* amax,
                              const float* scale, float* slab, int* n_slab, int B, int H, int Cin);
int launch_conv_first_assemble(hipStream_t st, const float* G, const float* M, const float* W, const float* bias, const float* coef,
                               float* dW, float* db, int Cin);
// split-bf16 form with transposed LDS reads (conv_wgrad_sb.hip): W = 16, 8 or 4; same slab layout as launch_conv64_wgrad
int conv64_wgrad_sb_usable(int W);
int launch_conv64_wgrad_sb(hipStream_t st, const float* x, const float* dz, float* slab, int* n_slab, int B, int H, int W);
int launch_conv64_wgrad(hipStream_t st, const float* x, const float* dz, float* slab, int* n_slab,
                        int B, int H, int W);
int conv_wgrad_slab_capacity();
int conv_first_wgrad_slab_stride(int Cin);   // floats per first-layer wgrad slab: rows 0..9*Cin-1 kernel, row 9*Cin bias
int launch_flip_weights(hipStream_t st, const float* w, float* wt);
// split-bf16 conv (conv_sb.hip): w [9][in][out] fp32 -> planes [9][3][out][in] bf16; conv with 6 bf16 MFMAs per product
int launch_split_weights(hipStream_t st, const float* w, unsigned short* wsp);
// up to 8 tensors in one launch; flip[i] != 0: the planes of the flipped (input-gradient) weights, from the unflipped tensor
int launch_split_weights_batch(hipStream_t st, int n, const float* const* w, unsigned short* const* dst, const int* flip);
extern int g_tn_lds_floor_kb;   // gemm_tn_sb.hip: extra dynamic LDS (KB) of the GRU weight-gradient batch launches (0 = none)
extern int g_gru_var;       // gru.hip: step-body variants (bit 0 forward, bit 1 backward)
extern int g_conv64_dbuf;   // conv_sb.hip: 1 = double-buffered-weights kernel (default)
// two-plane split of two floats for the four-product form: hi = the truncated upper 16 bits, mid = the residual ROUNDED to bf16 (half up on the
// magnitude: one integer add) — with a truncated mid the dropped remainder has the sign of x for every element, and the products it would have
// carried bias every sum by the same ~2^-16 relative; rounded, it is zero-mean and the dropped terms average out over a sum
__device__ __forceinline__ void split2r_pair(float x0, float x1, unsigned& h, unsigned& m) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    m = __builtin_amdgcn_perm(__float_as_uint(r1) + 0x8000u, __float_as_uint(r0) + 0x8000u, 0x07060302);
}
extern thread_local int g_gsb_four_now;    // gemm_sb.hip
extern int g_tn_tile_blocks;   // gemm_tn_sb.hip
extern int g_bwd_four;        // conv_sb.hip: backward-only products on four of the six split-bf16 terms (option "bwd_four_products")
// a backward-only launch_gemm_sb (input gradients: their B planes come from a transposed / flipped gemm_split_b job) inside this scope takes the four-product form
struct BwdFourScope {
    BwdFourScope() { g_gsb_four_now = g_bwd_four && !g_mfma_one; }
    ~BwdFourScope() { g_gsb_four_now = 0; }
};
int launch_conv64_dgrad_sb(hipStream_t st, const float* dz, const unsigned short* wsp_flip, float* dx, int B, int H, int W);
// pre_scale / pre_shift / pre_out (conv64_fwd_sb_takes_pre(W)): x holds the previous block's window extremes; its BatchNorm + ReLU are applied on load
// and the activated tensor is written to pre_out (!= x)
int launch_conv64_fwd_sb(hipStream_t st, const float* x, const unsigned short* wsp, const float* bias, float* z,
                         float* stat_partial, int* n_partial, int B, int H, int W, const float* pre_scale = nullptr, const float* pre_shift = nullptr,
                         float* pre_out = nullptr, const float* ext_gamma = nullptr, float* ext_out = nullptr);      // ext_*: W = 16, with pre_*: the (1,4) windows' extremes of z
int conv64_fwd_sb_takes_pre(int W);
int conv_sb_partial_capacity();  // [3,3,64,64] -> dgrad weights
int launch_reduce_slabs(hipStream_t st, const float* slab, int nslab, int64_t slab_stride, float* out,
                        int64_t n, int accumulate);
// thousands of slabs: groups of 64 into tmp [reduce_slabs_groups(nslab)][n], then the groups (gemm.hip)
int reduce_slabs_groups(int nslab);
int launch_reduce_slabs_2stage(hipStream_t st, const float* slab, int nslab, int64_t slab_stride, float* out, int64_t n, float* tmp);

int launch_bn_finalize(hipStream_t st, const float* partial, int npartial, double count, const float* gamma,
                       const float* beta, float* mov_mean, float* mov_var, float* mean, float* invstd,
                       float* scale, float* shift, int C, int update_moving);
int launch_bn_eval_coeffs(hipStream_t st, const float* gamma, const float* beta, const float* mov_mean,
                          const float* mov_var, float* scale, float* shift, int C);
int launch_bn_relu_pool_fwd(hipStream_t st, const float* z, const float* scale, const float* shift, float* p,
                            int B, int H, int W, int C, int pt, int pf);
int launch_bn_pool_bwd_reduce(hipStream_t st, const float* z, const float* p, const float* dp, const float* mean,
                              const float* invstd, const float* scale, const float* shift, float* partial,
                              int* npartial, int B, int H, int W, int C, int pt, int pf, int z_is_pooled_extreme = 0);
int launch_bn_bwd_finalize(hipStream_t st, const float* partial, int npartial, double count, float* dgamma,
                           float* dbeta, float* c1c2, int C);
// resnet.hip: channel-count-generic layers of resnet50_block (spec/RESNET50_BLOCK.md)
int rn_partial_capacity();
int launch_im2col3x3(hipStream_t st, const float* y, float* col, int B, int H, int W, int C);
int launch_col2im3x3(hipStream_t st, const float* dcol, float* dy, int B, int H, int W, int C);
int launch_rn_bn_stats(hipStream_t st, const float* z, float* partial, int* nbx, int64_t npix, int C);
int launch_rn_bn_bwd_reduce(hipStream_t st, const float* z, const float* dy, const float* mask, const float* coef, float* partial, int* nbx,
                            int64_t npix, int C, int gate_z = 0);
int launch_rn_bn_finalize(hipStream_t st, const float* partial, int nbx, double count, const float* gamma, const float* beta, float* mov_mean,
                          float* mov_var, float* coef, int C, int training, double* sums = nullptr, int phase = 0);
int launch_rn_bn_bwd_finalize(hipStream_t st, const float* partial, int nbx, double count, float* dgamma, float* dbeta, float* coef, int C,
                              double* sums = nullptr, int phase = 0);
// gate4 (optional): one byte per 4 channels, bit j = (out channel j > 0) — what the backward's three readers of the block output's gate take
int launch_rn_bn_apply(hipStream_t st, const float* z, const float* coef, const float* res, float* out, int64_t npix, int C, int relu,
                       unsigned char* gate4 = nullptr);
int launch_rn_add_gated(hipStream_t st, float* dst, const float* dy, const unsigned char* gate4, int64_t n);
// gate_z: 1 = the ReLU gate recomputed from z and the coefficients (a BatchNorm without residual) instead of read from `mask`;
// 2 = `mask` points to gate bytes (launch_rn_bn_apply's gate4)
int launch_rn_bn_bwd_dz(hipStream_t st, const float* z, const float* dy, const float* mask, const float* coef, float* dz, int64_t npix, int C,
                        int gate_z = 0);
int launch_rn_bn_apply2(hipStream_t st, const float* z, const float* coef, const float* zr, const float* coef_r, float* out, int64_t npix, int C,
                        unsigned char* gate4 = nullptr);
int launch_rn_add_masked(hipStream_t st, float* dst, const float* dy, const float* mask, int64_t n);
// the three products of a convolution ([M,K] rows x [K,N] kernel): split-bf16 kernels when the shape allows and pre-split planes are
// given (wsp: launch_gemm_split_b of w, wsp_t: of w^T), else the fp32 MFMA GEMM
int rn_sb_fwd_ok(int K, int N);
int rn_sb_dgrad_ok(int K, int N);
int rn_sb_wgrad_ok(int K, int N);
int launch_rn_product_fwd(hipStream_t st, const float* A, int lda, const float* w, const unsigned short* wsp, float* z, int M, int K, int N,
                          float* stat_part = nullptr, int* nbx = nullptr);
int launch_rn_product_dgrad(hipStream_t st, const float* dz, const float* w, const unsigned short* wsp_t, float* dA, int ldd, int M, int K, int N,
                            int accumulate, const float* addg = nullptr, const unsigned char* gate4 = nullptr);
int launch_rn_product_wgrad(hipStream_t st, const float* A, int lda, const float* dz, float* slab, int64_t slab_cap, float* dw, int M, int K, int N,
                            int split_bf16);
// 3x3 'same' convolution [B*H*W][C] -> [B*H*W][N] with the im2col rows formed on load (C, N powers of two >= 128: rn_conv3_sb_ok);
// wsp = split of w [9 C][N] (transb 0), wsp_flip = split with transb 2 (K = 9 N, N = C): the input-gradient convolution's matrix
// stage 0's 32 -> 32 3x3 as a 64 -> 64 3x3 over pairs of bins (resnet.hip): w [9][32][32] -> W2 [9][64][64]; dW2 -> dw
int launch_rn_w32_embed(hipStream_t st, const float* w, float* w2);
int launch_rn_w32_extract(hipStream_t st, const float* dw2, float* dw);
int rn_conv3_sb_ok(int C, int N);
int launch_rn_conv3_fwd(hipStream_t st, const float* img, const unsigned short* wsp, float* z, int B, int H, int W, int C, int N,
                        float* stat_part = nullptr, int* nbx = nullptr);
int launch_rn_conv3_dgrad(hipStream_t st, const float* dz, const unsigned short* wsp_flip, float* dimg, int B, int H, int W, int C, int N);
int launch_rn_conv3_wgrad(hipStream_t st, const float* img, const float* dz, float* slab, int64_t slab_cap, float* dw, int B, int H, int W, int C, int N);
// xception.hip: middle flow of xception_block (spec/XCEPTION_BLOCK.md)
int launch_xc_unit_fwd(hipStream_t st, const float* x, const float* kdw, const float* wpw, float* dwo, float* z, float* partial, int* npartial,
                       int B, int H, int W, const float* aff = nullptr);
int xc_partial_capacity();
int launch_dw3x3_fwd(hipStream_t st, const float* x, const float* k, float* y, int B, int H, int W, const float* aff = nullptr);
int launch_dw3x3_bwd_data(hipStream_t st, const float* dy, const float* k, const float* xin, const float* add, float* dx, int B, int H, int W,
                          const float* aff = nullptr);
int launch_dw3x3_bwd_w(hipStream_t st, const float* x, const float* dy, float* slab, int* nslab, int B, int H, int W, const float* aff = nullptr);
// input gradient + kernel-gradient slabs in one pass (W = 16): slab [xc_dw_fused_slabs(B, H)][576]
int xc_dw_fused_slabs(int B, int H);
// bn_partial (aff units only): + the backward sums of the BatchNormalization whose pre-BN tensor is xin, [xc_dw_fused_slabs][128]; fold with launch_xc_fold_partials
int launch_dw3x3_bwd_fused(hipStream_t st, const float* dy, const float* k, const float* xin, const float* add, float* dx, float* slab, int* nslab,
                           int B, int H, int W, const float* aff = nullptr, const float* bn_mean = nullptr, const float* bn_invstd = nullptr,
                           float* bn_partial = nullptr);
int launch_xc_fold_partials(hipStream_t st, const float* partial, int n, float* out, int* nout);
int launch_xc_bn_stats(hipStream_t st, const float* z, float* partial, int* npartial, int64_t npix);
int launch_xc_bn_bwd_reduce(hipStream_t st, const float* z, const float* dy, const float* mean, const float* invstd, float* partial,
                            int* npartial, int64_t npix);
int launch_xc_bn_apply(hipStream_t st, const float* z, const float* scale, const float* shift, const float* res, float* out, int64_t npix);
// BatchNorm' + both pointwise products of a unit's backward in one pass (xception.hip): f1 = dz W^T, slab[nslab][4096] = partial dwo^T dz
int xc_pw_bwd_slabs();
int launch_xc_pw_bwd(hipStream_t st, const float* z, const float* gy, const float* dwo, const float* wpw, const float* mean, const float* invstd,
                     const float* scale, const float* c1c2, float* f1, float* slab, int* nslab, int64_t npix);
int launch_xc_bn_bwd_dz(hipStream_t st, const float* z, const float* dy, const float* mean, const float* invstd, const float* scale,
                        const float* c1c2, float* dz, int64_t npix);
int launch_xc_ident(hipStream_t st, float* ident);
int launch_bn_partials_to_sums(hipStream_t st, const float* partial, int npartial, double* sums, double local_count);   // sums[128] = local_count
int launch_bn_finalize_sums(hipStream_t st, const double* sums, double count, const float* gamma, const float* beta,
                            float* mov_mean, float* mov_var, float* mean, float* invstd, float* scale, float* shift);
int launch_bn_bwd_local(hipStream_t st, const double* sums, float* dgamma, float* dbeta);
int launch_bn_bwd_c1c2(hipStream_t st, const double* sums, double count, float* c1c2);
int launch_pool_routing_patch(hipStream_t st, float* z, float* p, unsigned char* amax, const float* scale, const float* shift,
                              const int64_t* idx, const unsigned char* val, int64_t n, int H, int W, int pt, int pf);
int launch_relu_gate_patch_z(hipStream_t st, float* z, float* y, const float* scale, const float* shift, int C, const int64_t* idx,
                             const unsigned char* val, int64_t n);
int launch_affine_copy(hipStream_t st, const float* x, const float* aff, float* dst, int64_t n, int C);
int launch_relu_gate_patch(hipStream_t st, float* y, unsigned char* gate_bits, const int64_t* idx, const unsigned char* val, int64_t n);
int launch_pool_routing(hipStream_t st, const float* z, const float* p, const unsigned char* amax, const float* scale,
                        const float* shift, unsigned char* pos, unsigned char* gate, int B, int H, int W, int pt, int pf);
int launch_bn_pool_bwd_dz(hipStream_t st, const float* z, const float* dp, const float* mean, const float* invstd,
                          const float* scale, const float* shift, const float* c1c2, float* dz,
                          int B, int H, int W, int C, int pt, int pf);
int bn_partial_capacity();

// Optional epilogue extras of the NEXT launch_gemm / launch_gemm_sb call of this host thread (resnet50_block, round 5), set by a GemmEpiScope
// around the call the way BwdFourScope selects the four-product form:
//   stat_part : the product's BatchNorm statistics leave with its epilogue — per (row block, column) [sum | sum of squares] in the layout
//               rn_bn_finalize reads (partial[(chunk * nbx + row block) * 128 + {c, 64 + c}], chunk = column / 64, nbx = the launch's row
//               blocks: gemm_epi_row_blocks) — instead of a separate pass over z.  Requires no bias, no activation, mode 0, no accumulate.
//   addg, gate4: C += addg [gate bit]  (the identity shortcut's gated gradient added in the reduce convolution's input-gradient epilogue:
//               gate4[(row * ldc + col) / 4] bit (col & 3), what rn_bn_apply wrote beside the block output; ldc % 4 == 0)
struct GemmEpi { float* stat_part = nullptr; const float* addg = nullptr; const unsigned char* gate4 = nullptr; };
extern thread_local GemmEpi g_gemm_epi;
struct GemmEpiScope {
    GemmEpiScope(float* stat_part, const float* addg = nullptr, const unsigned char* gate4 = nullptr) { g_gemm_epi.stat_part = stat_part; g_gemm_epi.addg = addg; g_gemm_epi.gate4 = gate4; }
    ~GemmEpiScope() { g_gemm_epi = GemmEpi(); }
};
int gemm_epi_row_blocks(int M, int split_bf16);      // row blocks (= statistics partials per 64-channel chunk) of a launch: 128-row tiles (split-bf16) / 64-row tiles
int launch_gemm(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, const float* bias, float* C,
                int ldc, int M, int N, int K, int transb, int act, int accumulate);
int launch_gemm_dual_n(hipStream_t st, const float* A, int lda, const float* B0, const float* B1, int ldb, const float* bias0,
                       const float* bias1, float* C0, float* C1, int ldc, int M, int N, int K, int transb, int act);
int launch_gemm_mirror(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, const float* bias, float* C, float* mirror,
                       int ldc, int M, int N, int K, int transb, int act);
int launch_gemm_heads(hipStream_t st, const float* A, int lda, const float* Bm, const float* bias, float* C0, float* C1, float* M0,
                      float* M1, int M, int n0, int n1, int K, int act0, int act1);
int launch_heads_weff(hipStream_t st, const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2,
                      const int* n, int K, int Hd, float* weff);
int launch_heads_grad(hipStream_t st, const float* const* w1, const float* const* b1, const float* const* w2, float* const* dw1,
                      float* const* db1, float* const* dw2, float* const* db2, const int* n, int K, int Hd, const float* F,
                      const float* cs);
int launch_gemm_dual_k(hipStream_t st, const float* A0, const float* A1, int lda, const float* B0, const float* B1, int ldb,
                       const float* bias, float* C, int ldc, int M, int N, int K, int transb, int act, int accumulate);
// split-bf16 GEMM (gemm_sb.hip): weights pre-split into bf16 planes by launch_gemm_split_b (up to 16 operands per launch),
// then C = act(A B + bias) with mode 0 / 1 (two products sharing A) / 2 (one product over a concatenated K).
// Usable when K % 32 == 0, N % 128 == 0, lda % 4 == 0 and A is 16-byte aligned (gemm_sb_usable).
extern int g_gsb_dbg;   // tools/tune_gemm.py: 1 no loads in the 4-wave loop, 2 no stores, 4 force the 4-wave form, 8 the 16-wave form
size_t gemm_sb_split_elems(int K, int N);
int gemm_sb_usable(const void* A, int lda, int N, int K);
int launch_gemm_split_b(hipStream_t st, int njobs, const float* const* src, unsigned short* const* dst, const int* ldb,
                        const int* transb, const int* K, const int* N);
int launch_gemm_sb(hipStream_t st, const float* A0, const float* A1, int lda, const unsigned short* Bs0, const unsigned short* Bs1,
                   const float* bias0, const float* bias1, float* C0, float* C1, int ldc, int M, int N, int K, int act, int mode,
                   int accum = 0,       // accum: C += (the shortcut's input gradient lands on the reduce convolution's)
                   int conv_C = 0, int conv_H = 0, int conv_W = 0);   // conv_C > 0: A0 = NHWC image [M px][C], the product is its 3x3 'same'
                                                                       // convolution (K = 9 C, im2col rows formed on load, never stored)
// C[K1,N] = sum_m A[rowmap(m),K1]^T B[m,N]; rows are (b,t) with t in [0,S): A row uses t+shift (zero if outside)
// slab layout per split: [K1*N main | N column sums of B (valid if want_bias)]
int launch_gemm_tn(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, float* slab, int* nslab,
                   int M, int K1, int N, int S, int shift, int want_bias, int max_splits = 0, int chunk_rows = 32);
// split-bf16 form with transposed LDS reads (gemm_tn_sb.hip): K1 = 128, N % 128 == 0; same slabs as launch_gemm_tn
int gemm_tn_sb_usable(const void* A, int lda, const void* Bm, int ldb, int K1, int N);
// several products of one shape in one launch (+ one combine launch): job j's slabs follow job j-1's (nslab each)
#define TN_MAX_JOBS 4
struct TnJobs {
    const float* A[TN_MAX_JOBS];
    const float* B[TN_MAX_JOBS];
    int lda[TN_MAX_JOBS], shift[TN_MAX_JOBS];
    float* out_w[TN_MAX_JOBS];      // combine targets (launch_reduce_slabs2_batch)
    float* out_b[TN_MAX_JOBS];
};
int launch_gemm_tn_sb_batch(hipStream_t st, const TnJobs& jobs, int njobs, int ldb, float* slab, int* nslab, int M, int N, int S, int want_bias);
int launch_reduce_slabs2_batch(hipStream_t st, const float* slab, int nslab, int64_t stride, const TnJobs& jobs, int njobs, int64_t n_w, int64_t n_b);
int launch_gemm_tn_sb(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, float* slab, int* nslab, int M, int N, int S,
                      int shift, int want_bias);
// one product with K1 % 128 == 0 (resnet50_block's kernel gradients): slabs of K1 * N floats, `slab_cap` floats available
int launch_gemm_tn_sb_tiles(hipStream_t st, const float* A, int lda, const float* Bm, int ldb, float* slab, int64_t slab_cap, int* nslab,
                            int M, int K1, int N, int conv_C = 0, int conv_H = 0, int conv_W = 0);   // conv_C > 0: A = NHWC image, implicit im2col (K1 = 9 C)
int launch_reduce_slabs2(hipStream_t st, const float* slab, int nslab, int64_t stride, float* out_w, int64_t n_w,
                         float* out_b, int64_t n_b);
int gemm_tn_max_splits();
// floats of the kernel-gradient slab buffer a model context holds (the GRU's largest product, gemm_tn_max_splits() slabs of it)
inline int64_t tn_slab_capacity() { return (int64_t)gemm_tn_max_splits() * (384 * 384 + 384); }
int launch_colsum(hipStream_t st, const float* X, int ld, float* slab, int* nslab, int M, int N);

int launch_gru_fwd(hipStream_t st, const float* gx_f, const float* gx_b, const float* U_f, const float* U_b,
                   const float* brec_f, const float* brec_b, float* h_f, float* h_b, float* sv_f, float* sv_b,
                   int B, int S, const float* rm_f = nullptr, const float* rm_b = nullptr, float* hm_f = nullptr, float* hm_b = nullptr);
// gru_df.hip: the same recurrences without a workgroup barrier between steps (two wave groups, LDS step counters); option "gru_var" bit 4 / 5
int launch_gru_fwd_df(hipStream_t st, const float* gx_f, const float* gx_b, const float* U_f, const float* U_b,
                      const float* brec_f, const float* brec_b, float* h_f, float* h_b, float* sv_f, float* sv_b, int B, int S);
int gru_df_trace_read(int which, unsigned long long* out);
int launch_gru_bwd(hipStream_t st, const float* dout, const float* h_f, const float* h_b, const float* sv_f,
                   const float* sv_b, const float* U_f, const float* U_b, float* dgx_f, float* dgx_b,
                   float* dgh_f, float* dgh_b, int B, int S, const float* rm_f = nullptr, const float* rm_b = nullptr, const float* hm_f = nullptr,
                   const float* hm_b = nullptr);
int gru_timing_read(int which, unsigned long long* out, int blocks);
int launch_mul(hipStream_t st, const float* a, const float* b, float* out, int64_t n);

struct seld_loss_cfg;
int launch_act_bwd(hipStream_t st, const float* y, float* dy, int64_t n, int act);   // loss_adam.hip: dy *= act'(.) from y = act(.)
// loss_adam.hip: Conv1D('same') over a clip's frames as a dense product (rows laid side by side), its transpose, and counter-based dropout
int launch_time_expand(hipStream_t st, const float* x, float* xe, int B, int S, int C, int ks);
int launch_time_fold(hipStream_t st, const float* dxe, float* dx, int B, int S, int C, int ks, int accumulate);
int launch_dropout(hipStream_t st, const float* in, float* out, int64_t n, float rate, uint64_t seed, unsigned layer, unsigned step);
// a per-clip mask over the feature axis, constant over the clip's S rows (Keras GRU dropout masks): out[r][f] (+)= in[r][f] * mask[r / S][f]; F % 4 == 0
int launch_mask_rows(hipStream_t st, const float* in, const float* mask, float* out, int64_t rows, int S, int F, int accumulate);
int launch_fill(hipStream_t st, float* out, int64_t n, float v);
// loss_adam.hip: models.seldnet_v1's output coupling tanh(doa * [sed | sed | sed]) and its gradient (in place on the losses' gradients)
int launch_v1_couple_fwd(hipStream_t st, const float* sed, const float* doa1, float* out, float* out2, int rows, int nc);
int launch_v1_couple_bwd(hipStream_t st, const float* sed, const float* doa1, float* dsed_pre, int ld_sed, float* ddoa_pre, int ld_doa, int rows, int nc);
int launch_mmse_den(hipStream_t st, const float* y_doa, float* den, float* scratch, int rows, int nc);
int launch_losses(hipStream_t st, const float* sed, const float* doa, const float* y_sed, const float* y_doa,
                  int doa_loss, float w_sed, float w_doa, float sed_grad_scale, const float* den_dev,
                  float* sloss, float* dloss, float* dsed_pre, float* ddoa_pre, float* scratch, int B, int S, int nc, int ld_sed = 0,
                  int ld_doa = 0, int defer_finalize = 0);
int launch_losses_finalize(hipStream_t st, int doa_loss, const float* den_dev, float* sloss, float* dloss, float* scratch, int B, int S,
                           int nc);
int loss_scratch_floats(int rows);
int launch_adam(hipStream_t st, float* theta, const float* g, float* m, float* v, int64_t n, float lr_t,
                float beta1, float beta2, float eps);
int launch_agc(hipStream_t st, const float* theta, float* g, int64_t off, int rank, const int64_t* shape,
               float* scratch);
