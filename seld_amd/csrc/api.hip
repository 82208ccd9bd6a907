// api.hip — C ABI of libseld_hip.so (include/seld_hip.h): context, variable layout, and the
// orchestration of the SELDnet forward / backward / optimizer kernels on one HIP stream.
#include "common.h"
#include "../../include/seld_hip.h"

#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the functions are bound with dlsym (see struct Rccl)
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

namespace {

struct Var { std::string name; int64_t off; int rank; int64_t shape[4]; };

struct Timer { std::string name; std::vector<hipEvent_t> ev; int64_t launches = 0; double ms = 0.0; };

struct ConvL {
    int H, W, Cin, pt, pf;            // input geometry of this conv, pooling
    int64_t w_off, b_off, g_off, be_off;   // trainable offsets
    int64_t mm_off, mv_off;           // state offsets
    float *z = nullptr, *p = nullptr, *dp = nullptr;
    float* pd = nullptr;              // seld_arch.conv_dropout > 0: the block's output after Dropout (p stays the pooled tensor the backward reads)
    unsigned char* amax = nullptr;    // first block only: position of each pooling window's extreme [B,H/pt,W/pf,64]
    float* zext = nullptr;            // first block only: the windows' extreme z (kept next to p for the z-free backward)
    float *mean, *invstd, *scale, *shift, *c1c2;   // into small buffer
};

struct GruL {
    int in_feat;
    int64_t k_off[2], u_off[2], b_off[2];
    float *gx[2], *sv[2], *h[2], *out, *din;   // din: gradient w.r.t. this layer's input
    // seld_arch.gru_dropout > 0 (training): per direction the input mask [B][in_feat] and the state mask [B][128] (0 | 1/(1-rate)), the masked
    // input rows xm [rows][in_feat] (the kernel product's and the kernel gradient's operand), the masked state sequence hm [rows][128]
    // (h_prev of the backward pass and the recurrent-kernel gradient's operand); dtmp: the second direction's input gradient before its mask
    float *imask[2] = {}, *rmask[2] = {}, *xm[2] = {}, *hm[2] = {}, *dtmp = nullptr;
};

struct DenseL {
    int in, out; int64_t w_off, b_off; float* y; float* dy;      // in = the product's K (= ks * in_base)
    // simple_dense_block's hidden layers (modules.py:355-374): Conv1D kernel_size, Dropout rate; xe = the input rows laid side by side
    // [rows][ks * in_base] (ks > 1), yd = the layer's output after dropout (rate > 0), drop_id = the layer's dropout stream
    int ks = 1, in_base = 0; float rate = 0.f; float *xe = nullptr, *yd = nullptr; unsigned drop_id = 0;
};

// Conv2D(k in {1, 3}, strides (1, stride_f), use_bias=False) + BatchNormalization of resnet50_block (spec/RESNET50_BLOCK.md)
struct RnConv {
    int k = 1, Cin = 0, Cout = 0;
    int64_t w_off = 0, g_off = 0, be_off = 0, mm_off = 0, mv_off = 0;
    float *col = nullptr, *z = nullptr, *coef = nullptr;     // im2col of the input (k = 3), pre-BN output, [mean|invstd|scale|shift|c1|c2] x Cout
    unsigned short *wsp = nullptr, *wsp_t = nullptr;         // pre-split bf16 planes of the kernel / its transpose (shapes the split-bf16 GEMM takes)
    unsigned short *wsp9 = nullptr, *wsp9_flip = nullptr;    // 3x3, 64 -> 64 (stage 1): tap planes for the implicit-GEMM kernels of conv_sb.hip
    float *w2 = nullptr, *dw2 = nullptr;                     // 3x3, 32 -> 32 (stage 0): the kernel embedded as 64 -> 64 over pairs of bins, its gradient
};
struct RnBlock {
    int Cin, w, stride_f, Win, Wout;
    bool proj;
    RnConv c[3], sc;
    float *y0 = nullptr, *y1 = nullptr, *out = nullptr;      // ReLU(BN(c0)), ReLU(BN(c1)) [M, w]; block output [M, 4w]
    unsigned char* gate = nullptr;                          // [M, w]: bit j of byte q = (out[4 q + j] > 0), written by the forward's last pass
};

// one  ReLU -> SeparableConv2D(64, 3, use_bias=False) -> BatchNormalization  unit of xception_block's middle flow (spec/XCEPTION_BLOCK.md)
struct XcUnit {
    int64_t dw_off, pw_off, g_off, be_off;   // trainable offsets: depthwise_kernel [3,3,64,1], pointwise_kernel [1,1,64,64], gamma, beta
    int64_t mm_off, mv_off;                  // state offsets
    float *dwo = nullptr, *z = nullptr, *a = nullptr;   // depthwise output, pointwise output (pre-BN), unit output (units 0, 1)
    float *mean, *invstd, *scale, *shift, *c1c2;
};

struct Head {
    std::vector<DenseL> layers;   // dense chain, last = output layer with activation
    int act;
    int hidden_act = 0;           // simple_dense_block's dense_activation on the hidden layers (SELD_ACT_*; 0 = linear)
};

}  // namespace

struct seld_ctx {
    seld_arch arch;
    int B, Bmax, T, S, device;
    hipStream_t stream = nullptr;
    std::vector<Var> tr, nt;
    int64_t nparam = 0, nstate = 0;
    float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr, *state = nullptr;
    int64_t adam_step = 0;
    std::vector<ConvL> conv;
    std::vector<GruL> gru;
    // test aid (seld_debug_set_routing / seld_debug_set_relu_gates): decisions the NEXT backward passes are told to take
    struct Override { int kind, block, which; int64_t n; int64_t* idx; unsigned char* val; };
    std::vector<Override> overrides;
    Head heads[2];
    // xception_block (arch.first_kind == SELD_FIRST_XCEPTION): conv[0] is the entry block, then 3 * xc_blocks units on [B,S,16,64]
    std::vector<XcUnit> xc;
    std::vector<float*> xc_x;                // [xc_blocks + 1] module inputs: xc_x[0] = conv[0].p, xc_x[b + 1] = xc_x[b] + y
    float *xc_small = nullptr, *xc_ident = nullptr, *xc_feat = nullptr, *xc_part = nullptr, *xc_slab = nullptr;
    float* xc_unit_slab = nullptr;    // xc_nowait: per unit [pointwise slabs | depthwise slabs | first-stage sums]
    size_t xc_unit_slab_per = 0, xc_unit_slab_pw = 0, xc_unit_slab_dw = 0;
    int xc_nowait = 1;
    float* xc_slab_tmp = nullptr;     // first-stage sums of the fused pass's slabs (launch_reduce_slabs_2stage)
    float* xc_part_dw = nullptr;      // BatchNorm-backward partials left by the fused depthwise input-gradient pass, one [128] per workgroup
    size_t xc_slab_per = 0;      // floats per depthwise-slab buffer (xc_slab holds two)
    float *xc_g[4] = {}, *xc_dz2 = nullptr;  // gradient ping-pong buffers [B,S,16,64] (X, F1, F2, second F1); second dz buffer
    int xc_fused_pw_bwd = 1;                 // a unit's BatchNorm' + pointwise input / kernel gradients in one kernel (xc_pw_bwd)
    int xc_wgrad_side = 1;                   // xception_block backward: kernel gradients on the side stream (as rn_wgrad_side)
    // resnet50_block (arch.first_kind == SELD_FIRST_RESNET50): conv[0] is the entry block, then the bottleneck blocks
    std::vector<RnBlock> rn;
    float *rn_part = nullptr, *rn_part_side = nullptr, *rn_gx[2] = {}, *rn_bz[2] = {}, *rn_ba = nullptr, *rn_bb[3] = {}, *rn_bcol = nullptr;
    // resnet50_block backward: the kernel gradients run on the side stream beside the input-gradient chain; the dz buffers rotate
    // (ev_rn_free[slot]: the side stream's product that read the slot is done; slots 0-1 = rn_bz, 2-4 = rn_bb)
    hipEvent_t ev_rn_ready = nullptr, ev_rn_free[5] = {};
    float* rn_w9_slab = nullptr;           // slabs of the stage-1 3x3 kernel gradients (wgrad_slab belongs to the main stream's first block)
    int rn_wgrad_side = 1;
    size_t rn_col_elems = 0;
    int rn_implicit3x3 = 1;                // stages 2-3: the 3x3 products read im2col rows formed on load (0: materialised im2col / col2im)
    int rn_feat = 0;                         // features per label frame into the first GRU layer (2 x 32 rn_filters)
    float *feat_grad = nullptr;       // gradient w.r.t. the last pooled conv output ([B,S,128])
    float *dzbuf = nullptr, *small = nullptr, *stat_partial = nullptr, *bn_partial = nullptr;
    float *wgrad_slab = nullptr, *tn_slab = nullptr, *cs_slab = nullptr, *wflip = nullptr;
    float *wgrad_slab_side = nullptr, *dzbuf_alt = nullptr;      // conv_wgrad_side: the side stream's own slabs, the second dz buffer (allocated when the option is set)
    int conv_wgrad_side = 1;
    int dgrad_r8 = 1;      // conv_sb.hip g_sbd_dgrad_r8: the W = 16 four-product input gradient on 8-row tiles (round 5: 2.510 -> 2.496 ms same box)
    float *dgx[SELD_MAX_LAYERS][2] = {}, *dgh[SELD_MAX_LAYERS][2] = {};   // per GRU layer: the side stream reads them later
    float* tn_slab_side = nullptr;
    unsigned short* wsplit = nullptr;      // per 64->64 conv layer i: [2 i] forward, [2 i + 1] flipped; each [9][3][64][64] bf16 planes
    unsigned short *wsp_fwd[SELD_MAX_LAYERS] = {}, *wsp_bwd[SELD_MAX_LAYERS] = {};
    int conv1_gram = 1;                    // 1: first block's kernel gradient from the patch Gram matrix, no pre-BN tensor (conv_gram.hip)
    bool xc_fused_fwd = true;              // xception_block: depthwise + pointwise + BN statistics of a unit in one kernel
    int gram_parts = 2;                    // 2: the background Gram launch in two halves, one under each of the first two GRU layers' forward recurrences
    bool conv3_pre_fused = true;           // ... and the second block's (1,4) pooling pass: window extremes in its epilogue, BatchNorm + ReLU in the third block's loader
    bool conv2_pre_fused = true;           // the first block's BatchNorm + ReLU pass over its pooled tensor folded into the second block's region load
    bool gru_din_first = false;            // backward: a GRU layer's input-gradient product ahead of the side stream's release (measured: no gain, see backward_impl)
    bool gru_wgrad_batch = true;           // a GRU layer's four weight-gradient products in one launch (+ one combine)
    bool gram_active = false;              // the last training forward took that path
    float *gram_slab = nullptr, *gram = nullptr, *mmat = nullptr;
    hipEvent_t ev_gram = nullptr;
    int conv1_split_bf16 = 1;              // 1: the z-free first-block forward on bf16 MFMA with exactly split operands (conv_pool_sb.hip)
    int conv1_pool_fused = 1;              // 1: first block's (5,4) pool window reduction inside the conv epilogue (conv_pool.hip)
    int heads_fused = 1;                   // 1: heads of two LINEAR-then-activated layers run as one product with W1 W2 (see heads_lin)
    float *weff = nullptr, *dy_all = nullptr, *headF = nullptr;   // [K + 1][NT], [rows][NT], [K][NT] + [NT]
    int rn_split_bf16 = 1;                 // resnet50_block: products with N % 128 == 0 (stages 2-3, the expand / shortcut convolutions of
                                           // stages 0-1) on the split-bf16 kernels; 0: everything on the fp32 MFMA GEMM
    int gemm_split_bf16 = 1;               // 1: GRU input projections / heads' first Conv1D (and their input gradients) on the
                                           //    split-bf16 GEMM (gemm_sb.hip) where the shapes allow; 0: exact-fp32 MFMA GEMM
    unsigned short* gsplit = nullptr;      // pre-split weight operands of those products, refreshed by every forward
    unsigned short *ksp_fwd[SELD_MAX_LAYERS][2] = {}, *ksp_bwd[SELD_MAX_LAYERS][2] = {}, *h0sp_fwd[2] = {}, *h0sp_bwd[2] = {};
    int conv64_split_bf16 = 1;             // 1: conv2/conv3 forward + input gradient on bf16 MFMA with exact 3-way split operands
    hipStream_t side = nullptr;            // weight-gradient GEMMs run here, under the BPTT chain of the main stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_prep = nullptr;
    int prep_side = 0;      // option (off by default): the step's weight pre-pass on the side stream beside the first block's forward
    hipEvent_t ev_bucket[SELD_MAX_LAYERS] = {};   // side stream: GRU layer n_gru-1-k's (and, k = 0, the heads') gradients are final
    seld_allreduce_fn sync_fn = nullptr;          // synchronised BatchNorm (seld_set_sync_bn)
    void* sync_user = nullptr;
    int sync_world = 1;
    double* sync_buf = nullptr;                   // [128] (resnet50_block: [16][128]) sums handed to sync_fn
    bool sync_failed = false;                     // the all-reduce callback failed inside a helper: reported at the end of the pass
    int bf16_single = 0;                          // SELD_DTYPE_BF16 / option "bf16_single": one bf16 MFMA product per fp32 product (common.h g_mfma_one)
    // kernel choices the launchers read from process-wide variables (common.h): kept PER CONTEXT here and copied into those variables at the
    // start of every forward / backward pass (apply_kernel_choices), so that setting one on a context never changes another context's arithmetic
    int bwd_four_products = 1, gru_var = 11, conv64_dbuf = 1, tn_tile_blocks = 384, tn_lds_floor = 0, gram_bg_blocks = 192;
    int xc_fused_bn_sums = 1;              // ... and, for a folded unit, the previous BatchNormalization's backward sums too (0: xc_reduce's pass over (z, gY))
    int xc_fused_dw_bwd = 1;               // xception_block: the depthwise kernel gradient's slabs come out of the input-gradient pass (round 5; 0: dw3x3_bwd_w on the side stream)
    int xc_w16 = 1;                        // xception_block: the row-per-workgroup depthwise kernels for W = 16 (0: the generic kernel)
    int xc_xcd_map = 1;                    // xception_block: XCD-contiguous row ranges in the depthwise kernels (xception.hip; 0: identity map, for A/B)
    int rn_epi_stats = 1;                  // resnet50_block: a convolution's BatchNorm statistics leave with its product's epilogue (round 5; 0: the separate pass over z)
    int rn_epi_add = 1;                    // ... and the identity shortcut's gated gradient is added in the reduce convolution's input-gradient epilogue
    // data parallelism inside the library (seld_dp_*): one RCCL communicator, a communication stream, two events
    void* dp_comm = nullptr;                      // ncclComm_t
    int dp_rank = 0, dp_world = 1;
    hipStream_t dp_stream = nullptr;
    hipEvent_t ev_dp_main = nullptr, ev_dp_done = nullptr;
    float *dsed_pre = nullptr, *ddoa_pre = nullptr, *sed_int = nullptr, *doa_int = nullptr;
    float *doa_v1 = nullptr;                   // models.seldnet_v1 (models.py:36-52): tanh(doa * [sed | sed | sed]), the prediction the losses see
    float *head_tmp = nullptr;                 // [rows][max ks * in_base]: a Conv1D head layer's input gradient before it is folded back over the taps
    float* ones = nullptr;    // [B * 2048] of 1.0: the GRU dropout masks are launch_dropout of it
    uint64_t dropout_seed = 0x5e1d5e1d5e1d5e1dull; unsigned dropout_step = 0, dropout_cur = 0; int last_training = 0;   // dropout_cur: the counter the LAST training forward drew its masks with (its backward recomputes them)
    float *loss_scratch = nullptr, *den_dev = nullptr, *loss_out = nullptr;
    float *fin_sl = nullptr, *fin_dl = nullptr;   // deferred loss finalize of the running training step
    int fin_doa_loss = 0;
    std::vector<void*> allocs;
    std::string err;
    int prof = 0;   // 0 off, 1 major kernel groups, 2 every group
    std::vector<Timer> timers;
    std::vector<hipEvent_t> ev_pool;   // timing events are created once and recycled: no hipEventCreate inside a timed step
};

namespace {

std::string g_create_err;

int fail(seld_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(c, SELD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

template <typename T>
int dalloc(seld_ctx* c, T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T) + 256);
    if (e != hipSuccess) return fail(c, SELD_ERR_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}

void add_var(std::vector<Var>& v, int64_t& off, const std::string& name, std::initializer_list<int64_t> shape) {
    Var x;
    x.name = name;
    x.off = off;
    x.rank = (int)shape.size();
    int64_t n = 1;
    int i = 0;
    for (int k = 0; k < 4; ++k) x.shape[k] = 1;
    for (auto s : shape) { x.shape[i++] = s; n *= s; }
    off += n;
    v.push_back(x);
}

struct ProfScope {
    seld_ctx* c; int idx;
    ProfScope(seld_ctx* c_, const char* name, int level = 1) : c(c_), idx(-1) {
        if (c->prof < level) return;
        for (size_t i = 0; i < c->timers.size(); ++i) if (c->timers[i].name == name) idx = (int)i;
        if (idx < 0) { Timer t; t.name = name; c->timers.push_back(t); idx = (int)c->timers.size() - 1; }
        hipEvent_t e = take(c); hipEventRecord(e, c->stream); c->timers[idx].ev.push_back(e);
    }
    ~ProfScope() {
        if (idx < 0) return;
        hipEvent_t e = take(c); hipEventRecord(e, c->stream); c->timers[idx].ev.push_back(e);
        c->timers[idx].launches++;
    }
    static hipEvent_t take(seld_ctx* c) {
        if (c->ev_pool.empty()) { hipEvent_t e; hipEventCreate(&e); return e; }
        hipEvent_t e = c->ev_pool.back(); c->ev_pool.pop_back(); return e;
    }
};
#define PROF_CAT2(a, b) a##b
#define PROF_CAT(a, b) PROF_CAT2(a, b)
#define PROF(c, name) ProfScope PROF_CAT(prof_scope_, __LINE__)(c, name, 1)
#define PROF2(c, name) ProfScope PROF_CAT(prof_scope_, __LINE__)(c, name, 2)
// level 3: per-kernel-kind scopes INSIDE the level-1 groups of the block models (hundreds of event pairs per step: a separate
// profile pass of bench.py, never the pass that is timed for `value`)
#define PROF3(c, name) ProfScope PROF_CAT(prof_scope_, __LINE__)(c, name, 3)

int check_launch(seld_ctx* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, SELD_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return 0;
}

// ---- RCCL, bound at run time
struct Rccl {
    bool ok = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    Rccl() {
        void* h = nullptr;
        for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        ok = GetUniqueId && CommInitRank && CommDestroy && AllReduce && GetErrorString;
    }
};
Rccl& rccl() { static Rccl r; return r; }

// in-place SUM over the ranks of the library's communicator; 0 = enqueued
int dp_allreduce(seld_ctx* c, void* buf, int64_t count, int dtype, hipStream_t st) {
    if (!c->dp_comm || count < 0) return 1;
    if (count == 0) return 0;
    return rccl().AllReduce(buf, buf, (size_t)count, dtype == SELD_DTYPE_F64 ? ncclFloat64 : ncclFloat32, ncclSum,
                            static_cast<ncclComm_t>(c->dp_comm), st) == ncclSuccess ? 0 : 1;
}
int dp_sync_bn_fn(void* user, void* buf, int64_t count, int dtype, void* hip_stream) {
    return dp_allreduce(static_cast<seld_ctx*>(user), buf, count, dtype, static_cast<hipStream_t>(hip_stream));
}

}  // namespace

extern "C" {

int seld_abi_sizes(int32_t* out, int n) {
    const int32_t v[2] = {(int32_t)sizeof(seld_arch), (int32_t)sizeof(seld_loss_cfg)};
    for (int i = 0; out && i < n && i < 2; ++i) out[i] = v[i];
    return 2;
}

const char* seld_last_error(const seld_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int seld_create(const seld_arch* a, int B, int T, int dtype, int device, seld_ctx** out) {
    if (!a || !out) return fail(nullptr, SELD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (dtype != SELD_DTYPE_F32 && dtype != SELD_DTYPE_BF16) return fail(nullptr, SELD_ERR_UNSUPPORTED, "dtype must be SELD_DTYPE_F32 or SELD_DTYPE_BF16");
    if (B <= 0 || T <= 0) return fail(nullptr, SELD_ERR_INVALID, "B and T must be positive");
    if (a->n_conv < 1 || a->n_conv > SELD_MAX_LAYERS || a->n_gru < 1 || a->n_gru > SELD_MAX_LAYERS ||
        a->n_sed_dense < 0 || a->n_sed_dense > SELD_MAX_LAYERS || a->n_doa_dense < 0 || a->n_doa_dense > SELD_MAX_LAYERS)
        return fail(nullptr, SELD_ERR_INVALID, "layer counts out of range");
    if (a->in_ch != 7 && a->in_ch != 10)
        return fail(nullptr, SELD_ERR_UNSUPPORTED, "first conv kernels are built for in_ch = 7 (foa) and 10 (mic)");
    if (a->n_freq != 64) return fail(nullptr, SELD_ERR_UNSUPPORTED, "first conv kernel is built for n_freq = 64");
    if (a->n_classes <= 0) return fail(nullptr, SELD_ERR_INVALID, "n_classes must be positive");
    const bool xcep = a->first_kind == SELD_FIRST_XCEPTION, resn = a->first_kind == SELD_FIRST_RESNET50;
    if (a->first_kind != SELD_FIRST_SIMPLE_CONV && !xcep && !resn) return fail(nullptr, SELD_ERR_UNSUPPORTED, "unknown FIRST block kind");
    if (!(a->conv_dropout >= 0.f && a->conv_dropout < 1.f) || !(a->gru_dropout >= 0.f && a->gru_dropout < 1.f))
        return fail(nullptr, SELD_ERR_INVALID, "conv_dropout / gru_dropout: 0 <= rate < 1");
    if (a->conv_dropout > 0.f && a->first_kind != SELD_FIRST_SIMPLE_CONV)
        return fail(nullptr, SELD_ERR_UNSUPPORTED, "conv_dropout: simple_conv_block only (the other FIRST blocks' specs have no Dropout)");
    if (resn) {
        if (a->n_conv != 1 || a->pool_t[0] != 5 || a->pool_f[0] != 4 || a->rn_filters != 32)
            return fail(nullptr, SELD_ERR_UNSUPPORTED, "resnet50_block: one entry conv2d_bn(64) with pool (5,4), filters 32 (spec/RESNET50_BLOCK.md)");
        for (int s_ = 0; s_ < 4; ++s_)
            if (a->rn_blocks[s_] < 1 || a->rn_blocks[s_] > 8) return fail(nullptr, SELD_ERR_UNSUPPORTED, "resnet50_block: 1..8 blocks per stage");
    }
    if (xcep && (a->n_conv != 1 || a->pool_t[0] != 5 || a->pool_f[0] != 4 || a->xc_blocks < 1 || a->xc_blocks > SELD_MAX_XC_BLOCKS))
        return fail(nullptr, SELD_ERR_UNSUPPORTED, "xception_block: one entry conv2d_bn(64) with pool (5,4) and 1..16 middle modules (spec/XCEPTION_BLOCK.md)");
    int H = T, W = a->n_freq;
    for (int i = 0; i < a->n_conv; ++i) {
        if (a->filters[i] != 64) return fail(nullptr, SELD_ERR_UNSUPPORTED, "conv kernels are built for 64 filters");
        if (a->pool_t[i] <= 0 || a->pool_f[i] <= 0 || H % a->pool_t[i] || W % a->pool_f[i])
            return fail(nullptr, SELD_ERR_UNSUPPORTED, "time/frequency extents must be divisible by the pool sizes");
        if (i > 0 && !(W == 2 || W == 4 || W == 8 || W == 16 || W == 32))
            return fail(nullptr, SELD_ERR_UNSUPPORTED, "inner conv width must be a power of two <= 32");
        H /= a->pool_t[i];
        W /= a->pool_f[i];
    }
    if (xcep) {
        if (W != 16) return fail(nullptr, SELD_ERR_UNSUPPORTED, "xception_block: 16 frequency bins after the entry pool (n_freq 64)");
        W /= 8;       // exit MaxPooling2D((1, 8))
    }
    if (resn && W != 16) return fail(nullptr, SELD_ERR_UNSUPPORTED, "resnet50_block: 16 frequency bins after the entry pool (n_freq 64)");
    const int S = H, feat = resn ? (W / 8) * 32 * a->rn_filters : W * 64;
    for (int i = 0; i < a->n_gru; ++i)
        if (a->gru_units[i] != 128) return fail(nullptr, SELD_ERR_UNSUPPORTED, "GRU kernels are built for 128 units");
    if (feat <= 0 || feat % 128) return fail(nullptr, SELD_ERR_UNSUPPORTED, "GRU input projection expects a multiple of 128 features (seldnet.json: F'*C' = 2*64)");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, SELD_ERR_HIP, "no HIP device");
    if (device < 0 || device >= ndev) return fail(nullptr, SELD_ERR_INVALID, "bad device index");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, SELD_ERR_HIP, "hipSetDevice failed");

    seld_ctx* c = new seld_ctx();
    c->arch = *a; c->B = B; c->Bmax = B; c->T = T; c->S = S; c->device = device;
    c->bf16_single = dtype == SELD_DTYPE_BF16;

    // ---- variable layout (Keras creation order; oracle/seldnet_oracle.py::variable_specs is the twin)
    int64_t off = 0, soff = 0;
    int cin = a->in_ch;
    H = T; W = a->n_freq;
    for (int i = 0; i < a->n_conv; ++i) {
        ConvL L;
        L.H = H; L.W = W; L.Cin = cin; L.pt = a->pool_t[i]; L.pf = a->pool_f[i];
        char nm[64];
        snprintf(nm, sizeof nm, "conv%d.kernel", i); L.w_off = off; add_var(c->tr, off, nm, {3, 3, cin, 64});
        snprintf(nm, sizeof nm, "conv%d.bias", i);   L.b_off = off; add_var(c->tr, off, nm, {64});
        snprintf(nm, sizeof nm, "bn%d.gamma", i);    L.g_off = off; add_var(c->tr, off, nm, {64});
        snprintf(nm, sizeof nm, "bn%d.beta", i);     L.be_off = off; add_var(c->tr, off, nm, {64});
        snprintf(nm, sizeof nm, "bn%d.moving_mean", i);     L.mm_off = soff; add_var(c->nt, soff, nm, {64});
        snprintf(nm, sizeof nm, "bn%d.moving_variance", i); L.mv_off = soff; add_var(c->nt, soff, nm, {64});
        c->conv.push_back(L);
        cin = 64; H /= L.pt; W /= L.pf;
    }
    if (xcep)
        for (int b = 0; b < a->xc_blocks; ++b)
            for (int u = 0; u < 3; ++u) {
                XcUnit U;
                char nm[64];
                snprintf(nm, sizeof nm, "xc%d.%d.depthwise_kernel", b, u); U.dw_off = off; add_var(c->tr, off, nm, {3, 3, 64, 1});
                snprintf(nm, sizeof nm, "xc%d.%d.pointwise_kernel", b, u); U.pw_off = off; add_var(c->tr, off, nm, {1, 1, 64, 64});
                snprintf(nm, sizeof nm, "xc%d.%d.gamma", b, u); U.g_off = off; add_var(c->tr, off, nm, {64});
                snprintf(nm, sizeof nm, "xc%d.%d.beta", b, u); U.be_off = off; add_var(c->tr, off, nm, {64});
                snprintf(nm, sizeof nm, "xc%d.%d.moving_mean", b, u); U.mm_off = soff; add_var(c->nt, soff, nm, {64});
                snprintf(nm, sizeof nm, "xc%d.%d.moving_variance", b, u); U.mv_off = soff; add_var(c->nt, soff, nm, {64});
                c->xc.push_back(U);
            }
    if (resn) {
        int cin_b = 64, wcur = 16;
        for (int s_ = 0; s_ < 4; ++s_) {
            const int wd = a->rn_filters << s_;
            for (int b = 0; b < a->rn_blocks[s_]; ++b) {
                RnBlock R;
                R.Cin = cin_b; R.w = wd; R.stride_f = (b == 0 && s_ > 0) ? 2 : 1; R.Win = wcur; R.Wout = wcur / R.stride_f; R.proj = b == 0;
                auto mk = [&](RnConv& cv, const char* tag, int k, int ci, int co) {
                    char nm[96];
                    cv.k = k; cv.Cin = ci; cv.Cout = co;
                    snprintf(nm, sizeof nm, "rn%d.%d.%s.kernel", s_, b, tag); cv.w_off = off; add_var(c->tr, off, nm, {k, k, ci, co});
                    snprintf(nm, sizeof nm, "rn%d.%d.%s.gamma", s_, b, tag); cv.g_off = off; add_var(c->tr, off, nm, {co});
                    snprintf(nm, sizeof nm, "rn%d.%d.%s.beta", s_, b, tag); cv.be_off = off; add_var(c->tr, off, nm, {co});
                    snprintf(nm, sizeof nm, "rn%d.%d.%s.moving_mean", s_, b, tag); cv.mm_off = soff; add_var(c->nt, soff, nm, {co});
                    snprintf(nm, sizeof nm, "rn%d.%d.%s.moving_variance", s_, b, tag); cv.mv_off = soff; add_var(c->nt, soff, nm, {co});
                };
                mk(R.c[0], "c0", 1, cin_b, wd); mk(R.c[1], "c1", 3, wd, wd); mk(R.c[2], "c2", 1, wd, 4 * wd);
                if (R.proj) mk(R.sc, "sc", 1, cin_b, 4 * wd);
                c->rn.push_back(R);
                cin_b = 4 * wd; wcur = R.Wout;
            }
        }
        c->rn_feat = feat;
    }
    int fin = feat;
    for (int i = 0; i < a->n_gru; ++i) {
        GruL G;
        G.in_feat = fin;
        const char* dn[2] = {"fwd", "bwd"};
        for (int d = 0; d < 2; ++d) {
            char nm[64];
            snprintf(nm, sizeof nm, "gru%d.%s.kernel", i, dn[d]);           G.k_off[d] = off; add_var(c->tr, off, nm, {fin, 384});
            snprintf(nm, sizeof nm, "gru%d.%s.recurrent_kernel", i, dn[d]); G.u_off[d] = off; add_var(c->tr, off, nm, {128, 384});
            snprintf(nm, sizeof nm, "gru%d.%s.bias", i, dn[d]);             G.b_off[d] = off; add_var(c->tr, off, nm, {2, 384});
        }
        c->gru.push_back(G);
        fin = 128;
    }
    for (int hd = 0; hd < 2; ++hd) {
        const char* hn = hd == 0 ? "sed" : "doa";
        const int nd = hd == 0 ? a->n_sed_dense : a->n_doa_dense;
        const int32_t* units = hd == 0 ? a->sed_units : a->doa_units;
        int in = fin;
        for (int j = 0; j < nd; ++j) {
            if (units[j] <= 0 || (units[j] & 3)) { delete c; return fail(nullptr, SELD_ERR_UNSUPPORTED, "dense units must be a positive multiple of 4"); }
            DenseL D; D.out = units[j];
            D.ks = std::max(1, hd == 0 ? a->sed_kernel_size : a->doa_kernel_size);
            D.rate = hd == 0 ? a->sed_dropout : a->doa_dropout;
            if (D.ks > 15 || !(D.rate >= 0.f && D.rate < 1.f)) { delete c; return fail(nullptr, SELD_ERR_UNSUPPORTED, "simple_dense_block: kernel_size 1..15, 0 <= dropout_rate < 1"); }
            D.in_base = in; D.in = D.ks * in; D.drop_id = (unsigned)(16 * hd + j);
            char nm[64];
            snprintf(nm, sizeof nm, "%s.dense%d.kernel", hn, j); D.w_off = off; add_var(c->tr, off, nm, {D.ks, in, units[j]});
            snprintf(nm, sizeof nm, "%s.dense%d.bias", hn, j);   D.b_off = off; add_var(c->tr, off, nm, {units[j]});
            c->heads[hd].layers.push_back(D);
            in = units[j];
        }
        DenseL D; D.in = D.in_base = in; D.out = (hd == 0 ? 1 : 3) * a->n_classes;
        char nm[64];
        snprintf(nm, sizeof nm, "%s.out.kernel", hn); D.w_off = off; add_var(c->tr, off, nm, {in, D.out});
        snprintf(nm, sizeof nm, "%s.out.bias", hn);   D.b_off = off; add_var(c->tr, off, nm, {D.out});
        c->heads[hd].layers.push_back(D);
        c->heads[hd].act = hd == 0 ? 1 : 2;
        c->heads[hd].hidden_act = hd == 0 ? a->sed_dense_act : a->doa_dense_act;
        if (c->heads[hd].hidden_act < SELD_ACT_NONE || c->heads[hd].hidden_act > SELD_ACT_RELU) { delete c; return fail(nullptr, SELD_ERR_UNSUPPORTED, "dense_activation: none, sigmoid, tanh or relu"); }
    }
    c->nparam = off; c->nstate = soff;

    // ---- device memory
#define ALLOC(ptr, n) do { int rc_ = dalloc(c, &(ptr), (size_t)(n)); if (rc_) { g_create_err = c->err; seld_destroy(c); return rc_; } } while (0)
    ALLOC(c->params, c->nparam); ALLOC(c->grads, c->nparam); ALLOC(c->adam_m, c->nparam); ALLOC(c->adam_v, c->nparam);
    ALLOC(c->state, c->nstate);
    hipMemset(c->params, 0, c->nparam * 4); hipMemset(c->grads, 0, c->nparam * 4);
    hipMemset(c->adam_m, 0, c->nparam * 4); hipMemset(c->adam_v, 0, c->nparam * 4); hipMemset(c->state, 0, c->nstate * 4);
    ALLOC(c->small, (size_t)a->n_conv * 64 * 6);
    size_t zmax = 0;
    for (int i = 0; i < a->n_conv; ++i) {
        ConvL& L = c->conv[i];
        const size_t nz = (size_t)B * L.H * L.W * 64;
        const size_t np = (size_t)B * (L.H / L.pt) * (L.W / L.pf) * 64;
        ALLOC(L.z, nz); ALLOC(L.p, np); ALLOC(L.dp, np);
        if (a->conv_dropout > 0.f) ALLOC(L.pd, np);
        if (i == 0) { float* am = nullptr; ALLOC(am, (np + 3) / 4); L.amax = reinterpret_cast<unsigned char*>(am); ALLOC(L.zext, np); }
        if (i == 1 && L.W == 16 && L.pt == 1 && L.pf == 4) ALLOC(L.zext, np);      // the (1,4) windows' extremes of z (conv_sb.hip EXT), when the third block's loader pools
        if (nz > zmax) zmax = nz;
        float* sm = c->small + (size_t)i * 64 * 6;
        L.mean = sm; L.invstd = sm + 64; L.scale = sm + 128; L.shift = sm + 192; L.c1c2 = sm + 256;
    }
    ALLOC(c->dzbuf, zmax);
    {
        const size_t kp = (size_t)conv_gram_dim(c->conv[0].Cin);
        ALLOC(c->gram_slab, (size_t)conv_gram_slab_capacity() * kp * kp);
        ALLOC(c->gram, kp * kp);
        ALLOC(c->mmat, kp * 64);
    }
    if (xcep) {
        const size_t npx = (size_t)B * S * 16 * 64;     // elements of one [B,S,16,64] tensor
        ALLOC(c->xc_small, c->xc.size() * 64 * 6);
        ALLOC(c->xc_ident, 64 * 6);
        launch_xc_ident(0, c->xc_ident);
        for (size_t i = 0; i < c->xc.size(); ++i) {
            XcUnit& U = c->xc[i];
            ALLOC(U.dwo, npx); ALLOC(U.z, npx);
            if (i % 3 != 2) ALLOC(U.a, npx);
            float* sm = c->xc_small + i * 64 * 6;
            U.mean = sm; U.invstd = sm + 64; U.scale = sm + 128; U.shift = sm + 192; U.c1c2 = sm + 256;
        }
        c->xc_x.resize(a->xc_blocks + 1);
        c->xc_x[0] = c->conv[0].p;
        for (int b = 1; b <= a->xc_blocks; ++b) ALLOC(c->xc_x[b], npx);
        for (int k = 0; k < 4; ++k) ALLOC(c->xc_g[k], npx);
        ALLOC(c->xc_dz2, npx);
        ALLOC(c->xc_feat, (size_t)B * S * 128);
        ALLOC(c->xc_part, (size_t)xc_partial_capacity() * 128);
        {   // depthwise kernel-gradient slabs: dw3x3_bwd_w's (<= xc_partial_capacity()) or, with xc_fused_dw_bwd, one per 4 image rows, two buffers
            // (a unit's combine on the side stream reads one while the next unit's input-gradient kernel fills the other)
            const size_t per = (size_t)std::max(xc_partial_capacity(), xc_dw_fused_slabs(c->Bmax, c->S)) * 576;
            ALLOC(c->xc_slab, 2 * per);
            c->xc_slab_per = per;
            ALLOC(c->xc_part_dw, (size_t)xc_dw_fused_slabs(c->Bmax, c->S) * 128);
            ALLOC(c->xc_slab_tmp, (size_t)reduce_slabs_groups(xc_dw_fused_slabs(c->Bmax, c->S)) * 576);
            // xc_nowait (round 5): every unit its OWN slab buffers — [pointwise slabs | depthwise slabs | first-stage sums], ~20 MB per unit — so that no buffer
            // is written twice in a step and the backward loop needs no hand-over events (each wait costs the main stream ~5-10 us of bubble, 2 per unit)
            c->xc_unit_slab_pw = (size_t)xc_pw_bwd_slabs() * 4096;
            c->xc_unit_slab_dw = (size_t)xc_dw_fused_slabs(c->Bmax, c->S) * 576;
            c->xc_unit_slab_per = c->xc_unit_slab_pw + c->xc_unit_slab_dw + (size_t)reduce_slabs_groups(xc_dw_fused_slabs(c->Bmax, c->S)) * 576 +
                                  (size_t)reduce_slabs_groups(xc_pw_bwd_slabs()) * 4096;      // + the two combines' first-stage sums
            ALLOC(c->xc_unit_slab, c->xc.size() * c->xc_unit_slab_per);
        }
    }
    if (resn) {
        size_t mx_out = 0, mx_w = 0, mx_col = 0, mx_in = (size_t)B * S * 16 * 64;
        for (auto& R : c->rn) {
            const size_t M = (size_t)B * S * R.Wout;
            for (int i = 0; i < 3; ++i) { ALLOC(R.c[i].z, M * R.c[i].Cout); ALLOC(R.c[i].coef, (size_t)6 * R.c[i].Cout); }
            if (R.proj) { ALLOC(R.sc.z, M * 4 * R.w); ALLOC(R.sc.coef, (size_t)6 * 4 * R.w); }
            ALLOC(R.y0, M * R.w); ALLOC(R.y1, M * R.w); ALLOC(R.out, M * 4 * R.w); ALLOC(R.gate, M * R.w);
            mx_out = std::max(mx_out, M * 4 * R.w); mx_w = std::max(mx_w, M * R.w); mx_col = std::max(mx_col, M * 9 * R.w);
            mx_in = std::max(mx_in, (size_t)B * S * R.Win * R.Cin);
        }
        for (auto& R : c->rn)
            for (RnConv* cv : {&R.c[0], &R.c[1], &R.c[2], &R.sc}) {
                const int K = cv->k * cv->k * cv->Cin, N = cv->Cout;
                if (!N) continue;
                if (rn_sb_fwd_ok(K, N)) ALLOC(cv->wsp, gemm_sb_split_elems(K, N));
                if (rn_sb_dgrad_ok(K, N)) ALLOC(cv->wsp_t, gemm_sb_split_elems(K, N));
                if (cv->k == 3 && ((cv->Cin == 64 && N == 64) || (cv->Cin == 32 && N == 32))) {
                    ALLOC(cv->wsp9, (size_t)9 * 3 * 4096); ALLOC(cv->wsp9_flip, (size_t)9 * 3 * 4096);
                    if (N == 32) { ALLOC(cv->w2, (size_t)9 * 4096); ALLOC(cv->dw2, (size_t)9 * 4096); }
                }
            }
        ALLOC(c->rn_part, (size_t)rn_partial_capacity() * 16 * 128);
        ALLOC(c->rn_part_side, (size_t)rn_partial_capacity() * 16 * 128);
        ALLOC(c->rn_gx[0], mx_in); ALLOC(c->rn_gx[1], mx_in);
        for (auto& b_ : c->rn_bz) ALLOC(b_, mx_out);
        for (auto& b_ : c->rn_bb) ALLOC(b_, mx_w);
        ALLOC(c->rn_ba, mx_w);
        c->rn_col_elems = mx_col;      // the im2col tensors (a block's col, the shared dcol) are allocated on first use: no default path needs them
        ALLOC(c->rn_w9_slab, (size_t)conv_wgrad_slab_capacity() * (9 * 4096 + 64));
    }
    if (resn || a->first_kind == SELD_FIRST_XCEPTION) {
        bool ok_ = hipEventCreateWithFlags(&c->ev_rn_ready, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
        for (auto& e_ : c->ev_rn_free) ok_ = ok_ && hipEventCreateWithFlags(&e_, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
        if (!ok_) { seld_destroy(c); return fail(nullptr, SELD_ERR_HIP, "event creation failed"); }
    }
    ALLOC(c->stat_partial, (size_t)conv_stat_partial_capacity() * 128);
    ALLOC(c->bn_partial, (size_t)bn_partial_capacity() * 128);
    ALLOC(c->wgrad_slab, (size_t)conv_wgrad_slab_capacity() * (9 * 4096 + 64));
    if (a->n_conv >= 2 && c->xc.empty() && c->rn.empty()) {      // option conv_wgrad_side (simple_conv_block: the second / third block's kernel gradients beside the main chain)
        ALLOC(c->wgrad_slab_side, (size_t)conv_wgrad_slab_capacity() * (9 * 4096 + 64));
        ALLOC(c->dzbuf_alt, zmax);
        for (int k_ = 0; k_ < 2; ++k_)      // ev_rn_free[0 / 1]: the side-stream reader of dzbuf / dzbuf_alt is done (the block models create all five for their own slots)
            if (!c->ev_rn_free[k_] && hipEventCreateWithFlags(&c->ev_rn_free[k_], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
                seld_destroy(c); return fail(nullptr, SELD_ERR_HIP, "event creation failed");
            }
    }
    ALLOC(c->tn_slab, (size_t)tn_slab_capacity());
    ALLOC(c->cs_slab, (size_t)256 * 512);
    ALLOC(c->wflip, 9 * 4096);
    ALLOC(c->wsplit, (size_t)2 * c->conv.size() * 9 * 3 * 4096);
    for (size_t i = 0; i < c->conv.size(); ++i) {
        c->wsp_fwd[i] = c->wsplit + (2 * i) * 9 * 3 * 4096;
        c->wsp_bwd[i] = c->wsplit + (2 * i + 1) * 9 * 3 * 4096;
    }
    const size_t rows = (size_t)B * S;
    for (int i = 0; i < a->n_gru; ++i) {
        GruL& G = c->gru[i];
        for (int d = 0; d < 2; ++d) { ALLOC(G.gx[d], rows * 384); ALLOC(G.sv[d], rows * 512); ALLOC(G.h[d], rows * 128); }
        ALLOC(G.out, rows * 128); ALLOC(G.din, rows * (size_t)G.in_feat);
        if (a->gru_dropout > 0.f) {
            for (int d = 0; d < 2; ++d) {
                ALLOC(G.imask[d], (size_t)B * G.in_feat); ALLOC(G.rmask[d], (size_t)B * 128);
                ALLOC(G.xm[d], rows * (size_t)G.in_feat); ALLOC(G.hm[d], rows * 128);
            }
            ALLOC(G.dtmp, rows * (size_t)G.in_feat);
            if (!c->ones) { ALLOC(c->ones, (size_t)B * 2048); launch_fill(0, c->ones, (int64_t)B * 2048, 1.f); }
        }
    }
    ALLOC(c->feat_grad, rows * 128);
    for (int i = 0; i < a->n_gru; ++i)
        for (int d = 0; d < 2; ++d) { ALLOC(c->dgx[i][d], rows * 384); ALLOC(c->dgh[i][d], rows * 384); }
    // the side stream's slabs: one 384 x 384 product over gemm_tn_max_splits() splits, or a GRU layer's four 128 x 384 ones
    ALLOC(c->tn_slab_side, (size_t)gemm_tn_max_splits() * std::max<size_t>(384 * 384 + 384, 4 * (128 * 384 + 384)));
    // lowest priority: the side stream only carries work nobody waits for soon (weight-gradient GEMMs, the patch Gram
    // matrix); whenever the main stream has a kernel ready it should get the CUs
    int prio_lo = 0, prio_hi = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    // the events only order work between streams of THIS device: no system-scope fence on record (it cost ~4 us of main-stream
    // bubble per fork); host reads go through hipStreamSynchronize, peers through RCCL kernels that run on this device
    if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_lo) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
        // ev_join and the bucket events cross to a caller's communication stream (RCCL reads the gradients there and writes
        // them to peers): they keep the default system-scope release
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        // ev_prep carries DATA from the side stream to the main stream and is waited for microseconds after its record: it keeps the default release, to be on
        // the safe side of DESIGN.md section 6 item 8 (ev_gram, the other side -> main data event, is waited for a millisecond after its record)
        hipEventCreateWithFlags(&c->ev_prep, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_gram, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
        seld_destroy(c);
        return fail(nullptr, SELD_ERR_HIP, "side stream / event creation failed");
    }
    for (int i = 0; i < a->n_gru; ++i)
        if (hipEventCreateWithFlags(&c->ev_bucket[i], hipEventDisableTiming) != hipSuccess) {
            seld_destroy(c);
            return fail(nullptr, SELD_ERR_HIP, "event creation failed");
        }
    ALLOC(c->sync_buf, (resn ? 16 * 128 : 128) + 1);      // + this rank's element count, all-reduced with the sums
    for (int hd = 0; hd < 2; ++hd)
        for (auto& D : c->heads[hd].layers) {
            ALLOC(D.y, rows * (size_t)D.out); ALLOC(D.dy, rows * (size_t)D.out);
            if (D.ks > 1) ALLOC(D.xe, rows * (size_t)D.in);
            if (D.rate > 0.f) ALLOC(D.yd, rows * (size_t)D.out);
        }
    {
        size_t tmp = 0;
        for (int hd = 0; hd < 2; ++hd) for (auto& D : c->heads[hd].layers) if (D.ks > 1) tmp = std::max(tmp, rows * (size_t)D.in);
        if (tmp) ALLOC(c->head_tmp, tmp);
        if (a->output_coupling) ALLOC(c->doa_v1, rows * (size_t)(3 * a->n_classes));
    }
    {
        // pre-split bf16 planes: every GRU kernel and the heads' first layers, in the forward ([n][k]) and the
        // input-gradient ([in][out]) orientation
        size_t ne = 0;
        for (int i = 0; i < a->n_gru; ++i) ne += 4 * gemm_sb_split_elems(c->gru[i].in_feat, 384);
        for (int hd = 0; hd < 2; ++hd) ne += 2 * gemm_sb_split_elems(c->heads[hd].layers[0].in, c->heads[hd].layers[0].out);
        ALLOC(c->gsplit, ne);
        unsigned short* q = c->gsplit;
        for (int i = 0; i < a->n_gru; ++i)
            for (int d = 0; d < 2; ++d) {
                c->ksp_fwd[i][d] = q; q += gemm_sb_split_elems(c->gru[i].in_feat, 384);
                c->ksp_bwd[i][d] = q; q += gemm_sb_split_elems(c->gru[i].in_feat, 384);
            }
        for (int hd = 0; hd < 2; ++hd) {
            const DenseL& D = c->heads[hd].layers[0];
            c->h0sp_fwd[hd] = q; q += gemm_sb_split_elems(D.in, D.out);
            c->h0sp_bwd[hd] = q; q += gemm_sb_split_elems(D.in, D.out);
        }
    }
    {
        const int nt = c->heads[0].layers.back().out + c->heads[1].layers.back().out, k = c->heads[0].layers[0].in;
        ALLOC(c->weff, (size_t)(k + 1) * nt);
        ALLOC(c->dy_all, rows * (size_t)nt);
        ALLOC(c->headF, (size_t)k * nt + nt);
    }
    ALLOC(c->loss_scratch, (size_t)loss_scratch_floats((int)rows));
    ALLOC(c->den_dev, 4); ALLOC(c->loss_out, rows + 4);
#undef ALLOC
    if (hipDeviceSynchronize() != hipSuccess) { seld_destroy(c); return fail(nullptr, SELD_ERR_HIP, "device sync after allocation failed"); }
    *out = c;
    return SELD_OK;
}

void seld_destroy(seld_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (auto& t : c->timers) for (auto e : t.ev) hipEventDestroy(e);
    for (auto e : c->ev_pool) hipEventDestroy(e);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ev_prep) hipEventDestroy(c->ev_prep);
    if (c->ev_gram) hipEventDestroy(c->ev_gram);
    if (c->ev_rn_ready) hipEventDestroy(c->ev_rn_ready);
    for (auto e_ : c->ev_rn_free) if (e_) hipEventDestroy(e_);
    for (auto e : c->ev_bucket) if (e) hipEventDestroy(e);
    seld_dp_destroy(c);
    if (c->side) hipStreamDestroy(c->side);
    for (void* p : c->allocs) hipFree(p);
    delete c;
}

int seld_set_stream(seld_ctx* c, void* s) { if (!c) return SELD_ERR_INVALID; c->stream = (hipStream_t)s; return SELD_OK; }
int seld_set_batch(seld_ctx* c, int B) {
    if (!c) return SELD_ERR_INVALID;
    if (B < 1 || B > c->Bmax) return fail(c, SELD_ERR_INVALID, "batch exceeds the size given to seld_create");
    c->B = B;
    return SELD_OK;
}
int seld_set_option(seld_ctx* c, const char* key, int value) {
    if (!c || !key) return SELD_ERR_INVALID;
    if (!strcmp(key, "conv64_split_bf16")) { c->conv64_split_bf16 = value != 0; return SELD_OK; }
    if (!strcmp(key, "gemm_split_bf16")) { c->gemm_split_bf16 = value != 0; return SELD_OK; }
    if (!strcmp(key, "heads_fused")) { c->heads_fused = value != 0; return SELD_OK; }
    if (!strcmp(key, "dropout_seed")) { c->dropout_seed = 0x5e1d5e1d00000000ull ^ (uint64_t)(unsigned)value; return SELD_OK; }      // the masks are a function of (seed, step, layer, element)
    if (!strcmp(key, "dropout_step")) { c->dropout_step = (unsigned)value; return SELD_OK; }                                         // the NEXT training forward's step counter
    if (!strcmp(key, "conv1_split_bf16")) { c->conv1_split_bf16 = value != 0; return SELD_OK; }
    if (!strcmp(key, "gram_bg_blocks") && value >= 16 && value <= 512) { c->gram_bg_blocks = value; return SELD_OK; }   // tuning knob
    if (!strcmp(key, "conv1_pool_fused")) { c->conv1_pool_fused = value != 0; return SELD_OK; }
    if (!strcmp(key, "conv1_gram")) { c->conv1_gram = value != 0; return SELD_OK; }
    if (!strcmp(key, "gru_wgrad_batch")) { c->gru_wgrad_batch = value != 0; return SELD_OK; }
    if (!strcmp(key, "gru_din_first")) { c->gru_din_first = value != 0; return SELD_OK; }
    if (!strcmp(key, "conv2_pre_fused")) { c->conv2_pre_fused = value != 0; return SELD_OK; }
    if (!strcmp(key, "conv3_pre_fused")) { c->conv3_pre_fused = value != 0; return SELD_OK; }
    if (!strcmp(key, "gram_parts") && (value == 1 || value == 2)) { c->gram_parts = value; return SELD_OK; }
    if (!strcmp(key, "xc_fused_fwd")) { c->xc_fused_fwd = value != 0; return SELD_OK; }
    if (!strcmp(key, "rn_split_bf16")) { c->rn_split_bf16 = value != 0; return SELD_OK; }
    if (!strcmp(key, "rn_wgrad_side")) { c->rn_wgrad_side = value != 0; return SELD_OK; }
    if (!strcmp(key, "rn_implicit3x3")) { c->rn_implicit3x3 = value != 0; return SELD_OK; }
    if (!strcmp(key, "rn_epi_stats")) { c->rn_epi_stats = value != 0; return SELD_OK; }
    if (!strcmp(key, "rn_epi_add")) { c->rn_epi_add = value != 0; return SELD_OK; }
    if (!strcmp(key, "xc_wgrad_side")) { c->xc_wgrad_side = value != 0; return SELD_OK; }
    if (!strcmp(key, "xc_fused_pw_bwd")) { c->xc_fused_pw_bwd = value != 0; return SELD_OK; }
    // per-context kernel choices (apply_kernel_choices copies them into the launchers' variables at the start of each pass)
    if (!strcmp(key, "bwd_four_products")) { c->bwd_four_products = value != 0; return SELD_OK; }
    if (!strcmp(key, "tn_tile_blocks") && value >= 64 && value <= 4096) { c->tn_tile_blocks = value; return SELD_OK; }     // gemm_tn_sb.hip
    if (!strcmp(key, "gru_var")) {      // gru.hip: bit 0 forward VAR 1, bit 1 backward VAR 1, bit 3 falling priority, bits 4.. experimental bodies
        if (value < 0 || value > 255) return fail(c, SELD_ERR_INVALID, "gru_var: 0..255");
        c->gru_var = value; return SELD_OK;
    }
    if (!strcmp(key, "tn_lds_floor") && value >= 0 && value <= 100) { c->tn_lds_floor = value; return SELD_OK; }   // experiment: gemm_tn_sb.hip
    if (!strcmp(key, "conv64_dbuf")) { c->conv64_dbuf = value != 0; return SELD_OK; }     // conv_sb.hip
    if (!strcmp(key, "conv_wgrad_side")) { c->conv_wgrad_side = value != 0; return SELD_OK; }
    if (!strcmp(key, "dgrad_r8")) { c->dgrad_r8 = value != 0; return SELD_OK; }
    if (!strcmp(key, "prep_side")) { c->prep_side = value != 0; return SELD_OK; }
    if (!strcmp(key, "xc_nowait")) { c->xc_nowait = value != 0; return SELD_OK; }
    if (!strcmp(key, "xc_fused_bn_sums")) { c->xc_fused_bn_sums = value != 0; return SELD_OK; }
    if (!strcmp(key, "xc_fused_dw_bwd")) { c->xc_fused_dw_bwd = value != 0; return SELD_OK; }
    if (!strcmp(key, "xc_w16")) { c->xc_w16 = value != 0; return SELD_OK; }               // xception.hip
    if (!strcmp(key, "xc_xcd_map")) { c->xc_xcd_map = value != 0; return SELD_OK; }       // xception.hip
    if (!strcmp(key, "bf16_single")) { c->bf16_single = value != 0; return SELD_OK; }     // = SELD_DTYPE_BF16 at seld_create
    return fail(c, SELD_ERR_INVALID, std::string("unknown option: ") + key);
}
int seld_sync(seld_ctx* c) {
    if (!c) return SELD_ERR_INVALID;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SELD_OK;
}

int64_t seld_param_count(const seld_ctx* c) { return c ? c->nparam : -1; }
int64_t seld_state_count(const seld_ctx* c) { return c ? c->nstate : -1; }
int seld_variable_count(const seld_ctx* c, int trainable) { return c ? (int)(trainable ? c->tr.size() : c->nt.size()) : -1; }
int seld_variable_info(const seld_ctx* c, int trainable, int index, char* name, int name_cap, int64_t* offset,
                       int32_t* rank, int64_t shape[4]) {
    if (!c) return SELD_ERR_INVALID;
    const std::vector<Var>& v = trainable ? c->tr : c->nt;
    if (index < 0 || index >= (int)v.size()) return SELD_ERR_INVALID;
    if (name && name_cap > 0) { strncpy(name, v[index].name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = v[index].off;
    if (rank) *rank = v[index].rank;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = v[index].shape[k];
    return SELD_OK;
}

static int copy_h2d(seld_ctx* c, float* dst, const float* src, int64_t n, int64_t expect) {
    if (!c || !src || n != expect) return fail(c, SELD_ERR_INVALID, "size mismatch in host->device copy");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, src, (size_t)n * 4, hipMemcpyHostToDevice));
    return SELD_OK;
}
static int copy_d2h(seld_ctx* c, float* dst, const float* src, int64_t n, int64_t expect) {
    if (!c || !dst || n != expect) return fail(c, SELD_ERR_INVALID, "size mismatch in device->host copy");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, src, (size_t)n * 4, hipMemcpyDeviceToHost));
    return SELD_OK;
}
int seld_set_weights_host(seld_ctx* c, const float* w, int64_t n) { return c ? copy_h2d(c, c->params, w, n, c->nparam) : SELD_ERR_INVALID; }
int seld_get_weights_host(seld_ctx* c, float* w, int64_t n) { return c ? copy_d2h(c, w, c->params, n, c->nparam) : SELD_ERR_INVALID; }
int seld_set_state_host(seld_ctx* c, const float* s, int64_t n) { return c ? copy_h2d(c, c->state, s, n, c->nstate) : SELD_ERR_INVALID; }
int seld_get_state_host(seld_ctx* c, float* s, int64_t n) { return c ? copy_d2h(c, s, c->state, n, c->nstate) : SELD_ERR_INVALID; }
int seld_get_grads_host(seld_ctx* c, float* g, int64_t n) { return c ? copy_d2h(c, g, c->grads, n, c->nparam) : SELD_ERR_INVALID; }
int seld_get_adam_host(seld_ctx* c, float* m, float* v, int64_t n) {
    if (!c) return SELD_ERR_INVALID;
    int rc = copy_d2h(c, m, c->adam_m, n, c->nparam);
    if (rc) return rc;
    return copy_d2h(c, v, c->adam_v, n, c->nparam);
}
int seld_set_adam_host(seld_ctx* c, const float* m, const float* v, int64_t n, int64_t step) {
    if (!c || step < 0) return SELD_ERR_INVALID;
    int rc = copy_h2d(c, c->adam_m, m, n, c->nparam);
    if (rc) return rc;
    rc = copy_h2d(c, c->adam_v, v, n, c->nparam);
    if (rc) return rc;
    c->adam_step = step;
    return SELD_OK;
}
void* seld_param_ptr(seld_ctx* c) { return c ? c->params : nullptr; }
void* seld_grad_ptr(seld_ctx* c) { return c ? c->grads : nullptr; }

// ---------------------------------------------------------------------------------------------- forward
// side stream: everything enqueued on it after this call starts once the main stream has reached this point
static void fork_side(seld_ctx* c) {
    hipEventRecord(c->ev_fork, c->stream);
    hipStreamWaitEvent(c->side, c->ev_fork, 0);
}
// Which products run on the split-bf16 GEMM: shapes gemm_sb.hip handles (K % 32 == 0, N % 128 == 0); anything else stays
// on the exact-fp32 MFMA GEMM.  The same predicates gate the forward product and its input gradient.
static bool gru_sb(const seld_ctx* c, const GruL& G) { return c->gemm_split_bf16 && (G.in_feat % 128) == 0; }
// simple_dense_block with kernel_size > 1 or dropout_rate > 0 on a hidden layer: the heads run layer by layer (no shared first product,
// no W1 W2 fold)
static bool heads_general(const seld_ctx* c) {
    for (int hd = 0; hd < 2; ++hd)
        for (const DenseL& D : c->heads[hd].layers) if (D.ks > 1 || D.rate > 0.f) return true;
    return false;
}

static bool heads_sb(const seld_ctx* c) {
    const DenseL &S0 = c->heads[0].layers[0], &D0 = c->heads[1].layers[0];
    return !heads_general(c) && c->gemm_split_bf16 && c->heads[0].layers.size() > 1 && c->heads[1].layers.size() > 1 && S0.in == D0.in && S0.out == D0.out &&
           (S0.in % 128) == 0 && (S0.out % 128) == 0;
}

// Heads of the form Dense(h, linear) -> Dense(n, act) on shared features (seldnet.json: Conv1D(128, 1) then the output Dense):
// y = act(feat (W1 W2) + (b1 W2 + b2)).  The 128-wide hidden tensor is never formed, forward or backward: the products
// shrink from K = N = 128 to N = 12 + 36 columns in one launch, and all four weight gradients of a head follow from
// F = feat^T dy and colsum(dy) (gemm.hip, heads_grad_kernel).  Same mathematics, different association of the fp32 sums.
static bool heads_lin(const seld_ctx* c) {
    const Head &Hs = c->heads[0], &Hdo = c->heads[1];
    if (!c->heads_fused || Hs.layers.size() != 2 || Hdo.layers.size() != 2 || Hs.hidden_act || Hdo.hidden_act || heads_general(c)) return false;   // W1 W2 folds only without an activation between them
    const DenseL &S0 = Hs.layers[0], &D0 = Hdo.layers[0];
    return S0.in == D0.in && S0.out == D0.out && Hs.layers[1].out + Hdo.layers[1].out <= 64 && (S0.in & 3) == 0 &&
           ((Hs.layers[1].out + Hdo.layers[1].out) & 3) == 0;
}
static int prepare_heads_weff(seld_ctx* c, hipStream_t st) {
    const float *w1[2], *b1[2], *w2[2], *b2[2];
    int n[2];
    for (int hd = 0; hd < 2; ++hd) {
        const DenseL &L0 = c->heads[hd].layers[0], &L1 = c->heads[hd].layers[1];
        w1[hd] = c->params + L0.w_off; b1[hd] = c->params + L0.b_off; w2[hd] = c->params + L1.w_off; b2[hd] = c->params + L1.b_off;
        n[hd] = L1.out;
    }
    return launch_heads_weff(st, w1, b1, w2, b2, n, c->heads[0].layers[0].in, c->heads[0].layers[0].out, c->weff);
}

// one launch splits every weight operand the split-bf16 GEMMs of this step will read (the weights change every step)
static int prepare_gemm_splits(seld_ctx* c, hipStream_t st, bool with_grad_orientation) {
    const float* src[16]; unsigned short* dst[16]; int ldb[16], tb[16], K[16], N[16];
    int n = 0;
    auto flush = [&]() { int rc = n ? launch_gemm_split_b(st, n, src, dst, ldb, tb, K, N) : 0; n = 0; return rc; };
    auto add = [&](const float* w, unsigned short* d, int ld, int transb, int k, int nn) {
        src[n] = w; dst[n] = d; ldb[n] = ld; tb[n] = transb; K[n] = k; N[n] = nn;
        return ++n == 16 ? flush() : 0;
    };
    for (size_t i = 0; i < c->gru.size(); ++i) {
        const GruL& G = c->gru[i];
        if (!gru_sb(c, G)) continue;
        for (int d = 0; d < 2; ++d) {
            if (add(c->params + G.k_off[d], c->ksp_fwd[i][d], 384, 0, G.in_feat, 384)) return -1;     // gx = feat K
            if (with_grad_orientation && add(c->params + G.k_off[d], c->ksp_bwd[i][d], 384, 1, 384, G.in_feat)) return -1;   // din = dgx K^T
        }
    }
    if (heads_sb(c) && !heads_lin(c))
        for (int hd = 0; hd < 2; ++hd) {
            const DenseL& D = c->heads[hd].layers[0];
            if (add(c->params + D.w_off, c->h0sp_fwd[hd], D.out, 0, D.in, D.out)) return -1;
            if (with_grad_orientation && add(c->params + D.w_off, c->h0sp_bwd[hd], D.out, 1, D.out, D.in)) return -1;
        }
    return flush();
}

static void rn_bn(seld_ctx* c, hipStream_t st, RnConv& cv, int64_t M, int training, int nbx_have = 0, float* part = nullptr);
// the 3x3 convolution of a stage-1 bottleneck (64 -> 64 channels on a width conv_sb.hip / conv_wgrad_sb.hip have kernels for)
// ... of a stage-2 / 3 bottleneck: split-bf16 products on im2col rows formed on load (no col tensor)
static bool rn_c1_implicit(const seld_ctx* c, const RnBlock& R) { return c->rn_implicit3x3 && R.c[1].wsp && R.c[1].wsp_t && rn_conv3_sb_ok(R.w, R.w); }
// (stage 0: 32 -> 32 channels as 64 -> 64 over pairs of bins, rn_c1_width = the width the kernels see)
static int rn_c1_width(const RnBlock& R) { return R.c[1].w2 ? R.Wout / 2 : R.Wout; }
static bool rn_c1_direct(const RnBlock& R) {
    const int W = rn_c1_width(R);
    return R.c[1].wsp9 && (!R.c[1].w2 || (R.Wout & 1) == 0) && (W == 16 || W == 8 || W == 4);
}

// models.seldnet_v1 (models.py:36-52): doa <- tanh(doa * [sed | sed | sed]) after the two heads; the plain model returns as it is
static int heads_couple(seld_ctx* c, float* doa, int rows) {
    if (c->arch.output_coupling)
        launch_v1_couple_fwd(c->stream, c->heads[0].layers.back().y, c->heads[1].layers.back().y, c->doa_v1, doa, rows, c->arch.n_classes);
    return check_launch(c, "forward");
}

// the launchers' kernel-choice variables (common.h) take THIS context's values for the pass that starts here
static void apply_kernel_choices(const seld_ctx* c) {
    g_mfma_one = c->bf16_single;
    g_bwd_four = c->bwd_four_products;
    g_gru_var = c->gru_var;
    g_conv64_dbuf = c->conv64_dbuf;
    g_tn_tile_blocks = c->tn_tile_blocks;
    g_tn_lds_floor_kb = c->tn_lds_floor;
    g_gram_bg_blocks = c->gram_bg_blocks;
    g_xc_xcd_map = c->xc_xcd_map;
    g_xc_w16 = c->xc_w16;
    g_sbd_dgrad_r8 = c->dgrad_r8;
}

// resnet50_block: this step's pre-split weight planes (16 operands per launch), on `st`.  They depend on the parameters only: with `prep_side` they are made
// on the side stream beside the entry convolution and taken back (ev_prep) in front of the first stage (round 5: 0.15 ms of the 14.4-ms step).
static void rn_weight_prep(seld_ctx* c, hipStream_t st, bool save) {
    const float* src[16]; unsigned short* dst[16]; int ldb[16], tb[16], Ks[16], Ns[16];
    int n = 0;
    auto add = [&](const float* w, unsigned short* d, int ld, int transb, int k, int nn) {
        src[n] = w; dst[n] = d; ldb[n] = ld; tb[n] = transb; Ks[n] = k; Ns[n] = nn;
        if (++n == 16) { launch_gemm_split_b(st, n, src, dst, ldb, tb, Ks, Ns); n = 0; }
    };
    for (auto& R : c->rn)
        for (RnConv* cv : {&R.c[0], &R.c[1], &R.c[2], &R.sc}) {
            const int K = cv->k * cv->k * cv->Cin, N = cv->Cout;
            if (cv->wsp) add(c->params + cv->w_off, cv->wsp, N, 0, K, N);
            if (cv->wsp_t && save) {
                if (cv == &R.c[1] && rn_c1_implicit(c, R)) add(c->params + cv->w_off, cv->wsp_t, N, 2, 9 * N, cv->Cin);    // flipped taps
                else add(c->params + cv->w_off, cv->wsp_t, N, 1, N, K);
            }
        }
    if (n) launch_gemm_split_b(st, n, src, dst, ldb, tb, Ks, Ns);
    const float* w9[8]; unsigned short* d9[8]; int f9[8];
    n = 0;
    for (auto& R : c->rn) {
        if (!rn_c1_direct(R)) continue;
        const float* wsrc = c->params + R.c[1].w_off;
        if (R.c[1].w2) { launch_rn_w32_embed(st, wsrc, R.c[1].w2); wsrc = R.c[1].w2; }
        w9[n] = wsrc; d9[n] = R.c[1].wsp9; f9[n++] = 0;
        if (save) { w9[n] = wsrc; d9[n] = R.c[1].wsp9_flip; f9[n++] = 1; }
        if (n >= 7) { launch_split_weights_batch(st, n, w9, d9, f9); n = 0; }
    }
    if (n) launch_split_weights_batch(st, n, w9, d9, f9);
}

static int forward_impl(seld_ctx* c, const float* x, float* sed, float* doa, int training, bool save) {
    apply_kernel_choices(c);
    c->last_training = training;
    if (training) c->dropout_cur = c->dropout_step++;      // every training forward draws new masks (Keras), backward or not
    const bool conv_drop = training && c->arch.conv_dropout > 0.f, gru_drop = training && c->arch.gru_dropout > 0.f;
    if (gru_drop && !save) return fail(c, SELD_ERR_UNSUPPORTED, "gru_dropout: a training forward without saved gates");
    hipStream_t st = c->stream;
    const int B = c->B, S = c->S;
    const int rows = B * S;
    bool prep_on_side = false;
    const bool rn_prep_on_side = !c->rn.empty() && c->rn_split_bf16 && c->prep_side && c->ev_prep;
    if (rn_prep_on_side) { fork_side(c); rn_weight_prep(c, c->side, save); hipEventRecord(c->ev_prep, c->side); }
    // every weight-only pre-pass of the step in ONE launch (prep.hip): the split-bf16 planes of the GEMM and 64 -> 64 conv
    // weights (with the gradient orientations / flipped taps when a backward follows) and the folded head weights
    {
        GemmSplitJobs a; SplitWeightJobs b; HeadsLin h;
        a.one = b.one = g_mfma_one;     // bf16 single-product mode: plane 0 = round-to-nearest bf16 (prep.h)
        int na = 0, nb = 0;
        bool fits = true;
        auto adda = [&](const float* w, unsigned short* d, int ld, int transb, int k, int nn) {
            if (na == GSB_MAX_JOBS) { fits = false; return; }
            a.src[na] = w; a.dst[na] = d; a.ldb[na] = ld; a.transb[na] = transb; a.K[na] = k; a.N[na] = nn; ++na;
        };
        for (size_t i = 0; i < c->gru.size(); ++i) {
            const GruL& G = c->gru[i];
            if (!gru_sb(c, G)) continue;
            for (int d = 0; d < 2; ++d) {
                adda(c->params + G.k_off[d], c->ksp_fwd[i][d], 384, 0, G.in_feat, 384);                // gx = feat K
                if (save) adda(c->params + G.k_off[d], c->ksp_bwd[i][d], 384, 1, 384, G.in_feat);      // din = dgx K^T
            }
        }
        if (heads_sb(c) && !heads_lin(c))
            for (int hd = 0; hd < 2; ++hd) {
                const DenseL& D = c->heads[hd].layers[0];
                adda(c->params + D.w_off, c->h0sp_fwd[hd], D.out, 0, D.in, D.out);
                if (save) adda(c->params + D.w_off, c->h0sp_bwd[hd], D.out, 1, D.out, D.in);
            }
        a.njobs = na;
        if (c->conv64_split_bf16)
            for (size_t i = 1; i < c->conv.size(); ++i) {
                if (nb + 2 > 8) { fits = false; break; }
                b.w[nb] = c->params + c->conv[i].w_off; b.dst[nb] = c->wsp_fwd[i]; b.flip[nb++] = 0;
                if (save) { b.w[nb] = c->params + c->conv[i].w_off; b.dst[nb] = c->wsp_bwd[i]; b.flip[nb++] = 1; }
            }
        const bool lin = heads_lin(c);
        if (lin)
            for (int hd = 0; hd < 2; ++hd) {
                const DenseL &L0 = c->heads[hd].layers[0], &L1 = c->heads[hd].layers[1];
                h.w1[hd] = c->params + L0.w_off; h.b1[hd] = c->params + L0.b_off; h.w2[hd] = c->params + L1.w_off;
                h.b2[hd] = c->params + L1.b_off; h.n[hd] = L1.out;
                h.K = L0.in; h.Hd = L0.out;
            }
        if (fits && (!lin || h.K + 1 <= 4 * 144)) {
            // prep_side (round 5): none of these planes is read by the FIRST block's forward (it splits its own 7-channel kernel on load), so the pre-pass runs
            // on the side stream beside it; the main stream takes it back (ev_prep) behind the first block's launch.  The side stream's later work of the step
            // (the Gram launches, the kernel gradients) is ordered behind it by the stream itself.
            prep_on_side = c->prep_side && c->ev_prep && c->xc.empty() && c->rn.empty() && c->conv.size() >= 2;
            if (prep_on_side) fork_side(c);
            if (launch_weight_prep(prep_on_side ? c->side : st, a, na, b, nb, h, lin ? c->weff : nullptr)) return fail(c, SELD_ERR_UNSUPPORTED, "weight_prep");
            if (prep_on_side) hipEventRecord(c->ev_prep, c->side);
        } else {      // more jobs than one launch takes (not a seldnet.json shape): the stand-alone kernels
            if (prepare_gemm_splits(c, st, save)) return fail(c, SELD_ERR_UNSUPPORTED, "gemm_split_b");
            if (lin && prepare_heads_weff(c, st)) return fail(c, SELD_ERR_UNSUPPORTED, "heads_weff");
            if (c->conv64_split_bf16) {
                const float* w[8]; unsigned short* dst[8]; int flip[8];
                int n = 0;
                for (size_t i = 1; i < c->conv.size(); ++i) {
                    if (n + 2 > 8) { launch_split_weights_batch(st, n, w, dst, flip); n = 0; }
                    w[n] = c->params + c->conv[i].w_off; dst[n] = c->wsp_fwd[i]; flip[n++] = 0;
                    if (save) { w[n] = c->params + c->conv[i].w_off; dst[n] = c->wsp_bwd[i]; flip[n++] = 1; }
                }
                if (n && launch_split_weights_batch(st, n, w, dst, flip)) return fail(c, SELD_ERR_UNSUPPORTED, "split_weights");
            }
        }
    }
    const float* in = x;
    bool pre_pending = false;
    for (size_t i = 0; i < c->conv.size(); ++i) {
        ConvL& L = c->conv[i];
        bool ext_now = false;
        int npart = 0;
        float* stat = training ? c->stat_partial : nullptr;
        char tn[32];
        snprintf(tn, sizeof tn, "conv%d_fwd", (int)i + 1);
        // first block with the seldnet.json (5,4) pool: the conv epilogue reduces every pooling window of z
        // (conv_pool.hip), BN+ReLU+MaxPool becomes an elementwise pass over 1/20 of the data, z is stored
        // only when the backward pass will read it
        const bool fused_pool = i == 0 && c->conv1_pool_fused && L.pt == 5 && L.pf == 4 && L.W == 64;
        const bool gram = fused_pool && save && c->conv1_gram;      // backward without the pre-BN tensor: z is not stored
        if (i == 0) c->gram_active = gram;
        // the pooled tensor's BatchNorm + ReLU pass folded into the next block's loader (conv_sb.hip PRE): training with the Gram backward (zext kept
        // beside p), and inference (nobody reads p: the extremes go to zext and p is not written at all)
        const bool pre_next = fused_pool && (gram || !save) && c->conv2_pre_fused && !conv_drop && c->arch.first_kind == SELD_FIRST_SIMPLE_CONV &&
                              i + 1 < c->conv.size() && c->conv64_split_bf16 && !g_mfma_one && conv64_fwd_sb_takes_pre(c->conv[i + 1].W);
        if (fused_pool) {
            PROF(c, tn);   // level 1
            if (launch_conv_first_fwd_pool(st, in, c->params + L.w_off, c->params + L.b_off, c->params + L.g_off,
                                           (save && !gram) ? L.z : nullptr, (gram || pre_next) ? L.zext : L.p, save ? L.amax : nullptr, stat,
                                           &npart, B, L.H, L.Cin, c->conv1_split_bf16))
                return fail(c, SELD_ERR_UNSUPPORTED, "conv_first_fwd_pool");
        } else if (i == 0) {
            PROF(c, tn);   // level 1
            if (launch_conv_first_fwd(st, in, c->params + L.w_off, c->params + L.b_off, L.z, stat, &npart, B, L.H, L.Cin))
                return fail(c, SELD_ERR_UNSUPPORTED, "conv_first_fwd");
            if (save && L.pf == 4) launch_pool_argext(st, L.z, c->params + L.g_off, L.amax, B, L.H, L.W, L.pt, L.pf);   // for the fused backward
        } else {
            PROF2(c, tn);
            if (c->conv64_split_bf16) {
                // pre_pending: the first block's BatchNorm + ReLU ride in this block's region load (conv_sb.hip PRE), which also writes its pooled tensor
                const ConvL& P = c->conv[i - 1];
                // ext_now (option "conv3_pre_fused"): this block's (1,4) pooling is split the same way — the epilogue keeps every window's extreme of z
                // (EXT), the NEXT block's loader applies BatchNorm + ReLU to them and writes this block's pooled tensor: no pooling pass over z
                ext_now = pre_pending && c->conv3_pre_fused && L.zext && L.W == 16 && L.pt == 1 && L.pf == 4 && !conv_drop && i + 1 < c->conv.size() &&
                          c->conv[i + 1].W == 4 && conv64_fwd_sb_takes_pre(4);
                // (inference: the previous block's activated tensor is read by nobody -> not written)
                if (launch_conv64_fwd_sb(st, pre_pending ? P.zext : in, c->wsp_fwd[i], c->params + L.b_off, L.z, stat, &npart, B, L.H, L.W,
                                         pre_pending ? P.scale : nullptr, pre_pending ? P.shift : nullptr, (pre_pending && save) ? P.p : nullptr,
                                         ext_now ? c->params + L.g_off : nullptr, ext_now ? L.zext : nullptr))
                    return fail(c, SELD_ERR_UNSUPPORTED, "conv64_fwd_sb");
                pre_pending = false;
            } else if (launch_conv64_fwd(st, in, c->params + L.w_off, c->params + L.b_off, L.z, stat, &npart, B, L.H, L.W))
                return fail(c, SELD_ERR_UNSUPPORTED, "conv64_fwd");
        }
        if (i == 0 && prep_on_side) hipStreamWaitEvent(st, c->ev_prep, 0);      // everything behind the first block's convolution may read the pre-split planes
        if (training && c->sync_fn) {
            // synchronised BatchNorm: this rank's [sum z | sum z^2] -> the host's all-reduce -> coefficients of the GLOBAL batch
            launch_bn_partials_to_sums(st, c->stat_partial, npart, c->sync_buf, (double)B * L.H * L.W);
            if (c->sync_fn(c->sync_user, c->sync_buf, 129, SELD_DTYPE_F64, st)) return fail(c, SELD_ERR_HIP, "sync_bn all-reduce callback failed");
            launch_bn_finalize_sums(st, c->sync_buf, 0.0 /* the all-reduced count */, c->params + L.g_off, c->params + L.be_off,
                                    c->state + L.mm_off, c->state + L.mv_off, L.mean, L.invstd, L.scale, L.shift);
        } else if (training)
            launch_bn_finalize(st, c->stat_partial, npart, (double)B * L.H * L.W, c->params + L.g_off, c->params + L.be_off,
                               c->state + L.mm_off, c->state + L.mv_off, L.mean, L.invstd, L.scale, L.shift, 64, 1);
        else
            launch_bn_eval_coeffs(st, c->params + L.g_off, c->params + L.be_off, c->state + L.mm_off, c->state + L.mv_off,
                                  L.scale, L.shift, 64);
        snprintf(tn, sizeof tn, "pool%d_fwd", (int)i + 1);
        // options "conv2_pre_fused" / "conv3_pre_fused" (default 1): the pass is folded into the NEXT block's region load (pre_next: the first block's
        // BatchNorm + ReLU over its window extremes; ext_now: the second block's (1,4) pooling, whose extremes its own epilogue kept) — no launch here
        if (pre_next || ext_now)
            pre_pending = true;
        else {
            PROF2(c, tn);
            if (fused_pool)     // elementwise over zext (in place unless the backward keeps zext)
                launch_bn_relu_ext(st, gram ? L.zext : L.p, L.scale, L.shift, L.p, (int64_t)B * (L.H / 5) * 16 * 64);
            else if (launch_bn_relu_pool_fwd(st, L.z, L.scale, L.shift, L.p, B, L.H, L.W, 64, L.pt, L.pf))
                return fail(c, SELD_ERR_UNSUPPORTED, "bn_relu_pool_fwd");
        }
        in = L.p;
        if (conv_drop) {      // Dropout behind the pool (stream 64 + i); the backward masks the gradient arriving at this block with the same draws
            launch_dropout(st, L.p, L.pd, (int64_t)B * (L.H / L.pt) * (L.W / L.pf) * 64, c->arch.conv_dropout, c->dropout_seed, 64u + (unsigned)i, c->dropout_cur);
            in = L.pd;
        }
    }
    if (c->arch.first_kind == SELD_FIRST_XCEPTION) {
        // ---- xception_block middle flow + exit (spec/XCEPTION_BLOCK.md) on [B,S,16,64]
        const int64_t npix = (int64_t)B * S * 16;
        for (size_t i = 0; i < c->xc.size(); ++i) {
            XcUnit& U = c->xc[i];
            const size_t b = i / 3, u = i % 3;
            // with the fused unit kernel the previous unit's BatchNormalization is applied on load (relu(z scale + shift) of ITS pre-BN
            // tensor): units 0 and 1 of a module then never materialise their normalised output
            const bool fold = c->xc_fused_fwd && u > 0;
            const float* uin = u == 0 ? c->xc_x[b] : (fold ? c->xc[i - 1].z : c->xc[i - 1].a);
            const float* aff = fold ? c->xc[i - 1].scale : nullptr;      // [scale 64 | shift 64]
            int np = 0;
            if (c->xc_fused_fwd) {
                // depthwise + pointwise + BatchNorm statistics in one pass over the unit's input (xception.hip: xc_unit_fwd_kernel)
                PROF2(c, "xc_unit_fwd");
                if (launch_xc_unit_fwd(st, uin, c->params + U.dw_off, c->params + U.pw_off, U.dwo, U.z, training ? c->xc_part : nullptr, &np, B, S, 16, aff))
                    return fail(c, SELD_ERR_UNSUPPORTED, "xc_unit_fwd");
                if (np > xc_partial_capacity()) return fail(c, SELD_ERR_INVALID, "xception_block: more BatchNorm partials than xc_part holds");
            } else {
            {
                PROF2(c, "xc_depthwise_fwd");
                launch_dw3x3_fwd(st, uin, c->params + U.dw_off, U.dwo, B, S, 16);       // ReLU on load, no bias
            }
            {
                PROF2(c, "xc_pointwise_fwd");
                launch_gemm(st, U.dwo, 64, c->params + U.pw_off, 64, nullptr, U.z, 64, (int)npix, 64, 64, 0, 0, 0);
            }
            }
            PROF2(c, "xc_bn_fwd");
            if (training) {
                if (!c->xc_fused_fwd) launch_xc_bn_stats(st, U.z, c->xc_part, &np, npix);
                if (c->sync_fn) {      // synchronised BatchNorm: global sums through the host's all-reduce (see the conv blocks above)
                    launch_bn_partials_to_sums(st, c->xc_part, np, c->sync_buf, (double)npix);
                    if (c->sync_fn(c->sync_user, c->sync_buf, 129, SELD_DTYPE_F64, st)) return fail(c, SELD_ERR_HIP, "sync_bn all-reduce callback failed");
                    launch_bn_finalize_sums(st, c->sync_buf, 0.0 /* the all-reduced count */, c->params + U.g_off, c->params + U.be_off,
                                            c->state + U.mm_off, c->state + U.mv_off, U.mean, U.invstd, U.scale, U.shift);
                } else
                    launch_bn_finalize(st, c->xc_part, np, (double)npix, c->params + U.g_off, c->params + U.be_off, c->state + U.mm_off,
                                       c->state + U.mv_off, U.mean, U.invstd, U.scale, U.shift, 64, 1);
            } else {
                launch_bn_eval_coeffs(st, c->params + U.g_off, c->params + U.be_off, c->state + U.mm_off, c->state + U.mv_off, U.scale,
                                      U.shift, 64);
            }
            if (u == 2 || !c->xc_fused_fwd)
                launch_xc_bn_apply(st, U.z, U.scale, U.shift, u == 2 ? c->xc_x[b] : nullptr, u == 2 ? c->xc_x[b + 1] : U.a, npix);
        }
        // exit: ReLU -> MaxPooling2D((1, 8)) = the BN+ReLU+pool kernel with identity coefficients
        PROF2(c, "xc_exit_pool");
        if (launch_bn_relu_pool_fwd(st, c->xc_x.back(), c->xc_ident + 128, c->xc_ident + 192, c->xc_feat, B, S, 16, 64, 1, 8))
            return fail(c, SELD_ERR_UNSUPPORTED, "xception exit pool");
        in = c->xc_feat;
    }
    if (c->arch.first_kind == SELD_FIRST_RESNET50) {
        // ---- resnet50_block stages (spec/RESNET50_BLOCK.md): every convolution a product (resnet.hip: launch_rn_product_*)
        PROF(c, "rn_stages_fwd");
        const bool sb = c->rn_split_bf16 != 0;
        const bool epi_stats = training && c->rn_epi_stats;      // BatchNorm statistics in the products' epilogues (common.h GemmEpi)
        if (sb) {      // this step's weight planes: made at the start of the forward on the side stream (rn_prep_on_side), or here
            PROF3(c, "rn_weight_prep");
            if (rn_prep_on_side) hipStreamWaitEvent(st, c->ev_prep, 0);
            else rn_weight_prep(c, st, save);
        }
        const float* X = in;      // [B,S,Win,Cin]
        for (auto& R : c->rn) {
            if (c->sync_failed) break;     // a failed SyncBN collective: enqueue nothing further (the error is reported below)
            const int64_t M = (int64_t)B * S * R.Wout;
            const int w = R.w;
            // the projection shortcut (first block of a stage) depends on the block input only: side stream, joined before the add
            const bool sc_side = R.proj && c->rn_wgrad_side && !c->sync_fn;
            if (sc_side) {
                hipEventRecord(c->ev_rn_ready, st); hipStreamWaitEvent(c->side, c->ev_rn_ready, 0);
                int nb_ = 0;
                launch_rn_product_fwd(c->side, X, R.Cin * R.stride_f, c->params + R.sc.w_off, sb ? R.sc.wsp : nullptr, R.sc.z, (int)M, R.Cin, 4 * w,
                                      epi_stats ? c->rn_part_side : nullptr, &nb_);
                rn_bn(c, c->side, R.sc, M, training, nb_, c->rn_part_side);
                hipEventRecord(c->ev_rn_free[0], c->side);
            }
            // 1x1 (frequency stride = doubled row stride of the operand), BN, ReLU
            int nb0 = 0;
            { PROF3(c, "rn_products_fwd"); launch_rn_product_fwd(st, X, R.Cin * R.stride_f, c->params + R.c[0].w_off, sb ? R.c[0].wsp : nullptr, R.c[0].z, (int)M, R.Cin, w,
                                                                 epi_stats ? c->rn_part : nullptr, &nb0); }
            { PROF3(c, "rn_bn_fwd"); rn_bn(c, st, R.c[0], M, training, nb0); }
            { PROF3(c, "rn_bn_fwd"); launch_rn_bn_apply(st, R.c[0].z, R.c[0].coef, nullptr, R.y0, M, w, 1); }
            // 3x3, BN, ReLU: 64 -> 64 (stage 1) on the implicit-GEMM kernel of the conv blocks (BatchNorm's sums from its epilogue),
            // the other widths as a product on im2col rows
            if (sb && rn_c1_direct(R)) {
                int npart = 0;
                if (R.c[1].w2) {      // stage 0: the epilogue's sums are per (bin parity, channel): the statistics pass instead
                    { PROF3(c, "rn_products_fwd"); launch_conv64_fwd_sb(st, R.y0, R.c[1].wsp9, nullptr, R.c[1].z, nullptr, nullptr, B, S, rn_c1_width(R)); }
                    { PROF3(c, "rn_bn_fwd"); rn_bn(c, st, R.c[1], M, training); }
                } else {
                    { PROF3(c, "rn_products_fwd"); launch_conv64_fwd_sb(st, R.y0, R.c[1].wsp9, nullptr, R.c[1].z, training ? c->rn_part : nullptr, &npart, B, S, R.Wout); }
                    // the conv epilogue's partial sums land in rn_part ([rn_partial_capacity()][16][128] floats): the producer's count must fit
                    if (npart > rn_partial_capacity() * 16) return fail(c, SELD_ERR_INVALID, "resnet50_block: more BatchNorm partials than rn_part holds");
                    { PROF3(c, "rn_bn_fwd"); rn_bn(c, st, R.c[1], M, training, npart); }
                }
            } else if (sb && rn_c1_implicit(c, R)) {
                int nb1 = 0;
                { PROF3(c, "rn_products_fwd"); launch_rn_conv3_fwd(st, R.y0, R.c[1].wsp, R.c[1].z, B, S, R.Wout, w, w, epi_stats ? c->rn_part : nullptr, &nb1); }
                { PROF3(c, "rn_bn_fwd"); rn_bn(c, st, R.c[1], M, training, nb1); }
            } else {
                // (only with rn_split_bf16 / rn_implicit3x3 off, or a width no direct kernel takes: the col tensor is allocated here, once)
                if (!R.c[1].col && dalloc(c, &R.c[1].col, (size_t)M * 9 * w)) return fail(c, SELD_ERR_NOMEM, "im2col tensor");
                launch_im2col3x3(st, R.y0, R.c[1].col, B, S, R.Wout, w);
                int nb1 = 0;
                { PROF3(c, "rn_products_fwd"); launch_rn_product_fwd(st, R.c[1].col, 9 * w, c->params + R.c[1].w_off, sb ? R.c[1].wsp : nullptr, R.c[1].z, (int)M, 9 * w, w,
                                                                     epi_stats ? c->rn_part : nullptr, &nb1); }
                { PROF3(c, "rn_bn_fwd"); rn_bn(c, st, R.c[1], M, training, nb1); }
            }
            { PROF3(c, "rn_bn_fwd"); launch_rn_bn_apply(st, R.c[1].z, R.c[1].coef, nullptr, R.y1, M, w, 1); }
            // 1x1 expand, BN; shortcut; out = ReLU(y + r)
            int nb2 = 0;
            { PROF3(c, "rn_products_fwd"); launch_rn_product_fwd(st, R.y1, w, c->params + R.c[2].w_off, sb ? R.c[2].wsp : nullptr, R.c[2].z, (int)M, w, 4 * w,
                                                                 epi_stats ? c->rn_part : nullptr, &nb2); }
            { PROF3(c, "rn_bn_fwd"); rn_bn(c, st, R.c[2], M, training, nb2); }
            if (R.proj) {
                if (sc_side) hipStreamWaitEvent(st, c->ev_rn_free[0], 0);
                else {
                    int nbs = 0;
                    { PROF3(c, "rn_products_fwd"); launch_rn_product_fwd(st, X, R.Cin * R.stride_f, c->params + R.sc.w_off, sb ? R.sc.wsp : nullptr, R.sc.z, (int)M, R.Cin, 4 * w,
                                                                         epi_stats ? c->rn_part : nullptr, &nbs); }
                    { PROF3(c, "rn_bn_fwd"); rn_bn(c, st, R.sc, M, training, nbs); }
                }
                { PROF3(c, "rn_bn_fwd"); launch_rn_bn_apply2(st, R.c[2].z, R.c[2].coef, R.sc.z, R.sc.coef, R.out, M, 4 * w, save ? R.gate : nullptr); }
            } else {
                { PROF3(c, "rn_bn_fwd"); launch_rn_bn_apply(st, R.c[2].z, R.c[2].coef, X, R.out, M, 4 * w, 1, save ? R.gate : nullptr); }
            }
            X = R.out;
        }
        if (c->sync_failed) { c->sync_failed = false; return fail(c, SELD_ERR_HIP, "sync_bn all-reduce callback failed"); }
        in = X;       // [B,S,2,1024] = [B,S,2048]
    }
    const float* feat = in;  // [B,S,128] (force_1d_inputs: feature = f*64 + c)
    const int gparts = !c->gram_active ? 0 : (c->gram_parts == 2 && c->gru.size() >= 2 ? 2 : 1);
    int gram_ns = 0;
    for (size_t i = 0; i < c->gru.size(); ++i) {
        GruL& G = c->gru[i];
        if (gru_drop) {
            // Keras GRU dropout / recurrent_dropout (modules.py:312-314): each direction's cell draws its own input mask (stream 96 + 4 i + d) and
            // state mask (98 + 4 i + d); the directions no longer share their input rows, so the projections are two products
            PROF(c, "gru_fwd");
            const float rate = c->arch.gru_dropout;
            for (int d = 0; d < 2; ++d) {
                launch_dropout(st, c->ones, G.imask[d], (int64_t)B * G.in_feat, rate, c->dropout_seed, 96u + 4u * (unsigned)i + d, c->dropout_cur);
                launch_dropout(st, c->ones, G.rmask[d], (int64_t)B * 128, rate, c->dropout_seed, 98u + 4u * (unsigned)i + d, c->dropout_cur);
                launch_mask_rows(st, feat, G.imask[d], G.xm[d], rows, S, G.in_feat, 0);
                if (gru_sb(c, G) && gemm_sb_usable(G.xm[d], G.in_feat, 384, G.in_feat))
                    launch_gemm_sb(st, G.xm[d], nullptr, G.in_feat, c->ksp_fwd[i][d], nullptr, c->params + G.b_off[d], nullptr, G.gx[d], nullptr, 384, rows, 384,
                                   G.in_feat, 0, 0);
                else
                    launch_gemm(st, G.xm[d], G.in_feat, c->params + G.k_off[d], 384, c->params + G.b_off[d], G.gx[d], 384, rows, 384, G.in_feat, 0, 0, 0);
            }
            if (launch_gru_fwd(st, G.gx[0], G.gx[1], c->params + G.u_off[0], c->params + G.u_off[1], c->params + G.b_off[0] + 384,
                               c->params + G.b_off[1] + 384, G.h[0], G.h[1], G.sv[0], G.sv[1], B, S, G.rmask[0], G.rmask[1], G.hm[0], G.hm[1]))
                return fail(c, SELD_ERR_UNSUPPORTED, "gru_fwd (dropout)");
        } else {
        {
            PROF2(c, "gru_inproj_gemm");
            // both directions' projections of the same input in one launch
            if (gru_sb(c, G) && gemm_sb_usable(feat, G.in_feat, 384, G.in_feat))
                launch_gemm_sb(st, feat, nullptr, G.in_feat, c->ksp_fwd[i][0], c->ksp_fwd[i][1], c->params + G.b_off[0],
                               c->params + G.b_off[1], G.gx[0], G.gx[1], 384, rows, 384, G.in_feat, 0, 1);
            else
                launch_gemm_dual_n(st, feat, G.in_feat, c->params + G.k_off[0], c->params + G.k_off[1], 384, c->params + G.b_off[0],
                                   c->params + G.b_off[1], G.gx[0], G.gx[1], 384, rows, 384, G.in_feat, 0, 0);
        }
        if ((int)i < gparts) fork_side(c);
        {
            PROF(c, "gru_fwd");
            launch_gru_fwd(st, G.gx[0], G.gx[1], c->params + G.u_off[0], c->params + G.u_off[1], c->params + G.b_off[0] + 384,
                           c->params + G.b_off[1] + 384, G.h[0], G.h[1], save ? G.sv[0] : nullptr, save ? G.sv[1] : nullptr, B, S);
        }
        }
        if ((int)i < gparts && gru_drop) fork_side(c);
        if ((int)i < gparts) {
            // Gram matrix of the input patches (conv_gram.hip): depends on x alone -> side stream, under the GRU
            // recurrences (2B of the 256 CUs): eligible when the first GRU kernel is (fork event recorded in front of it) but
            // enqueued after it, on a lower-priority stream, so that the recurrence gets its CUs first
            // option "gram_parts" = 2: half of the tiles under each of the first two layers' recurrences (each part released by its own fork)
            int ns = 0;
            const int kp = conv_gram_dim(c->conv[0].Cin);
            if (i == 0) gram_ns = 0;
            if (launch_conv_first_gram(c->side, x, c->gram_slab + (size_t)gram_ns * kp * kp, &ns, B, c->conv[0].H, c->conv[0].Cin, 1, (int)i, gparts))
                return fail(c, SELD_ERR_UNSUPPORTED, "conv_first_gram");
            gram_ns += ns;
            if ((int)i == gparts - 1) {
                launch_reduce_slabs(c->side, c->gram_slab, gram_ns, (int64_t)kp * kp, c->gram, (int64_t)kp * kp, 0);
                hipEventRecord(c->ev_gram, c->side);
            }
        }
        launch_mul(st, G.h[0], G.h[1], G.out, (int64_t)rows * 128);
        feat = G.out;
    }
    {
        PROF2(c, "heads_fwd");
        // the first layers of the two heads read the same features: one launch when their shapes agree (seldnet.json:
        // Conv1D(128) in both) and neither is the head's output layer
        DenseL &S0 = c->heads[0].layers[0], &D0 = c->heads[1].layers[0];
        if (heads_lin(c)) {
            DenseL &S1 = c->heads[0].layers[1], &D1 = c->heads[1].layers[1];
            const int nt = S1.out + D1.out;
            if (launch_gemm_heads(st, feat, S0.in, c->weff, c->weff + (size_t)S0.in * nt, S1.y, D1.y, sed, doa, rows, S1.out, D1.out,
                                  S0.in, c->heads[0].act, c->heads[1].act))
                return fail(c, SELD_ERR_UNSUPPORTED, "gemm_heads");
            return heads_couple(c, doa, rows);
        }
        if (heads_general(c)) {
            // layer by layer: [rows laid side by side ->] product + bias + activation [-> dropout]
            for (int hd = 0; hd < 2; ++hd) {
                const float* a = feat;
                Head& Hd = c->heads[hd];
                float* outp = hd == 0 ? sed : doa;
                for (size_t j = 0; j < Hd.layers.size(); ++j) {
                    DenseL& D = Hd.layers[j];
                    const bool lastl = (j + 1 == Hd.layers.size());
                    if (D.ks > 1) { launch_time_expand(st, a, D.xe, c->B, c->S, D.in_base, D.ks); a = D.xe; }
                    launch_gemm_mirror(st, a, D.in, c->params + D.w_off, D.out, c->params + D.b_off, D.y, lastl ? outp : nullptr, D.out,
                                       rows, D.out, D.in, 0, lastl ? Hd.act : Hd.hidden_act);
                    a = D.y;
                    if (!lastl && D.rate > 0.f && training) {
                        launch_dropout(st, D.y, D.yd, (int64_t)rows * D.out, D.rate, c->dropout_seed, D.drop_id, c->dropout_cur);
                        a = D.yd;
                    }
                }
            }
            return heads_couple(c, doa, rows);
        }
        const bool merged0 = c->heads[0].layers.size() > 1 && c->heads[1].layers.size() > 1 && S0.in == D0.in && S0.out == D0.out &&
                             c->heads[0].hidden_act == c->heads[1].hidden_act;      // one launch, one epilogue activation
        const int hact0 = c->heads[0].hidden_act;
        if (merged0 && heads_sb(c) && gemm_sb_usable(feat, S0.in, S0.out, S0.in))
            launch_gemm_sb(st, feat, nullptr, S0.in, c->h0sp_fwd[0], c->h0sp_fwd[1], c->params + S0.b_off, c->params + D0.b_off, S0.y,
                           D0.y, S0.out, rows, S0.out, S0.in, hact0, 1);
        else if (merged0)
            launch_gemm_dual_n(st, feat, S0.in, c->params + S0.w_off, c->params + D0.w_off, S0.out, c->params + S0.b_off,
                               c->params + D0.b_off, S0.y, D0.y, S0.out, rows, S0.out, S0.in, 0, hact0);
        for (int hd = 0; hd < 2; ++hd) {
            const float* a = feat;
            Head& Hd = c->heads[hd];
            float* outp = hd == 0 ? sed : doa;
            for (size_t j = 0; j < Hd.layers.size(); ++j) {
                DenseL& D = Hd.layers[j];
                const bool lastl = (j + 1 == Hd.layers.size());
                float* y = D.y;
                // the head's output layer also writes the caller's copy (no device-to-device copy afterwards)
                if (!(merged0 && j == 0))
                    launch_gemm_mirror(st, a, D.in, c->params + D.w_off, D.out, c->params + D.b_off, y, lastl ? outp : nullptr, D.out,
                                       rows, D.out, D.in, 0, lastl ? Hd.act : Hd.hidden_act);
                a = y;
            }
        }
    }
    return heads_couple(c, doa, rows);
}

int seld_forward(seld_ctx* c, const float* x, float* sed, float* doa, int training) {
    if (!c || !x) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    return forward_impl(c, x, sed, doa, training, false);
}

static int run_losses(seld_ctx* c, const float* y_sed, const float* y_doa, const seld_loss_cfg* cfg, float* sloss,
                      float* dloss, bool want_grads, bool defer_finalize = false) {
    hipStream_t st = c->stream;
    const int rows = c->B * c->S, nc = c->arch.n_classes;
    if (cfg->doa_loss < SELD_DOA_MSE || cfg->doa_loss > SELD_DOA_MSLE) return fail(c, SELD_ERR_INVALID, "bad doa_loss");
    if (cfg->doa_loss == SELD_DOA_MMSE) {
        if (cfg->mmse_den > 0.f) {
            if (hipMemcpyAsync(c->den_dev, &cfg->mmse_den, 4, hipMemcpyHostToDevice, st) != hipSuccess)
                return fail(c, SELD_ERR_HIP, "den copy failed");
            hipStreamSynchronize(st);  // cfg may live on the caller's stack
        } else {
            launch_mmse_den(st, y_doa, c->den_dev, c->loss_scratch, rows, nc);
            // data parallel (seld_dp_init): the mask count of the GLOBAL batch, summed in place on this stream — no host round trip
            // TRAINING path only (want_grads): seld_test_step is not a collective — train.teststep knows nothing about process groups, a
            // validation loop may run on one rank or with unequal batch counts, and its loss is this rank's own num / den
            if (want_grads && c->dp_comm && c->dp_world > 1 && dp_allreduce(c, c->den_dev, 1, SELD_DTYPE_F32, st)) return fail(c, SELD_ERR_HIP, "RCCL all-reduce of the MMSE denominator failed");
        }
    }
    float* sl = sloss ? sloss : c->loss_out;
    float* dl = dloss ? dloss : c->loss_out + 4;
    // fused linear heads: both pre-activation gradients side by side in one [rows][n_sed + n_doa] buffer (the K axis of dfeat)
    const bool lin = heads_lin(c);
    const int n0 = c->heads[0].layers.back().out, nt = n0 + c->heads[1].layers.back().out;
    launch_losses(st, c->heads[0].layers.back().y, c->arch.output_coupling ? c->doa_v1 : c->heads[1].layers.back().y, y_sed, y_doa, cfg->doa_loss, cfg->w_sed,
                  cfg->w_doa, cfg->sed_grad_scale, c->den_dev, sl, dl,
                  want_grads ? (lin ? c->dy_all : c->heads[0].layers.back().dy) : nullptr,
                  want_grads ? (lin ? c->dy_all + n0 : c->heads[1].layers.back().dy) : nullptr, c->loss_scratch, c->B, c->S, nc,
                  lin ? nt : 0, lin ? nt : 0, defer_finalize ? 1 : 0);
    // seldnet_v1: the losses left d / d(doa sed) in the DOA slot; through the product to the two heads' pre-activations
    if (want_grads && c->arch.output_coupling)
        launch_v1_couple_bwd(st, c->heads[0].layers.back().y, c->heads[1].layers.back().y, lin ? c->dy_all : c->heads[0].layers.back().dy,
                             lin ? nt : n0, lin ? c->dy_all + n0 : c->heads[1].layers.back().dy, lin ? nt : nt - n0, rows, nc);
    if (defer_finalize) { c->fin_sl = sl; c->fin_dl = dl; c->fin_doa_loss = cfg->doa_loss; }
    return check_launch(c, "losses");
}

int seld_mmse_den(seld_ctx* c, const float* y_doa, float* den) {
    if (!c || !y_doa || !den) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    launch_mmse_den(c->stream, y_doa, den, c->loss_scratch, c->B * c->S, c->arch.n_classes);
    return check_launch(c, "mmse_den");
}

int seld_test_step(seld_ctx* c, const float* x, const float* y_sed, const float* y_doa, const seld_loss_cfg* cfg,
                   float* sed, float* doa, float* sloss, float* dloss) {
    if (!c || !x || !y_sed || !y_doa || !cfg) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = forward_impl(c, x, sed, doa, 0, false);
    if (rc) return rc;
    return run_losses(c, y_sed, y_doa, cfg, sloss, dloss, false);
}

// ---------------------------------------------------------------------------------------------- backward
// dW[K1,N] = A^T B via slabs, into grads at w_off; optional time shift on A rows
// dW[K1,N] = A^T B and db[N] = colsum(B) in one TN launch + one fixed-order slab reduction
static void wgrad_dense(seld_ctx* c, hipStream_t st, float* slab, const float* A, int lda, const float* Bm, int ldb, int M,
                        int K1, int N, int64_t w_off, int64_t b_off, int S, int shift) {
    int ns = 0;
    // both slab buffers hold gemm_tn_max_splits() slabs of 384 x 384 + 384 floats: larger products (a 2048-feature GRU input) take fewer splits
    const int64_t cap = tn_slab_capacity() / ((int64_t)K1 * N + N);
    if (c->gemm_split_bf16 && gemm_tn_sb_usable(A, lda, Bm, ldb, K1, N)) launch_gemm_tn_sb(st, A, lda, Bm, ldb, slab, &ns, M, N, S, shift, 1);
    else launch_gemm_tn(st, A, lda, Bm, ldb, slab, &ns, M, K1, N, S, shift, 1, (int)std::min<int64_t>(cap, gemm_tn_max_splits()));
    launch_reduce_slabs2(st, slab, ns, (int64_t)K1 * N + N, c->grads + w_off, (int64_t)K1 * N, c->grads + b_off, N);
}

// dW[K1,N] = A^T B for a bias-free convolution of a FIRST block (xception / resnet50), as many short splits as the slab buffer holds


// BatchNormalization of a resnet50_block convolution: statistics (training) or moving statistics -> cv.coef
// nbx_have > 0: the convolution's epilogue already left that many [sum | sum of squares] partials in rn_part (Cout = 64)
// part: the partial-sum scratch (default rn_part; the side stream's projection shortcut has its own)
static void rn_bn(seld_ctx* c, hipStream_t st, RnConv& cv, int64_t M, int training, int nbx_have, float* part) {
    int nbx = nbx_have;
    if (!part) part = c->rn_part;
    if (training && !nbx_have) launch_rn_bn_stats(st, cv.z, part, &nbx, M, cv.Cout);
    float *g = c->params + cv.g_off, *be = c->params + cv.be_off, *mm = c->state + cv.mm_off, *mv = c->state + cv.mv_off;
    if (training && c->sync_fn) {
        // synchronised BatchNorm: this rank's per-chunk sums -> the host's all-reduce -> coefficients of the GLOBAL batch
        const int nd = (cv.Cout + 63) / 64 * 128;
        launch_rn_bn_finalize(st, part, nbx, (double)M, g, be, mm, mv, cv.coef, cv.Cout, 1, c->sync_buf, 1);
        if (c->sync_fn(c->sync_user, c->sync_buf, nd + 1, SELD_DTYPE_F64, st)) { c->sync_failed = true; return; }      // + the element count
        launch_rn_bn_finalize(st, part, nbx, 0.0, g, be, mm, mv, cv.coef, cv.Cout, 1, c->sync_buf, 2);
        return;
    }
    launch_rn_bn_finalize(st, part, nbx, (double)M, g, be, mm, mv, cv.coef, cv.Cout, training);
}
// backward of the same: dz = BN'(dy [mask > 0]) into `dz`, dgamma / dbeta into the gradient buffer
// gate4 == nullptr: the BatchNorm feeds a ReLU directly (no residual) and the gate is recomputed from z; else the block output's gate bytes
static void rn_bn_bwd(seld_ctx* c, hipStream_t st, RnConv& cv, const float* dy, const unsigned char* gate4, float* dz, int64_t M) {
    int nbx = 0;
    const int gate_z = gate4 ? 2 : 1;
    const float* mask = reinterpret_cast<const float*>(gate4);
    launch_rn_bn_bwd_reduce(st, cv.z, dy, mask, cv.coef, c->rn_part, &nbx, M, cv.Cout, gate_z);
    if (c->sync_fn) {
        const int nd = (cv.Cout + 63) / 64 * 128;
        launch_rn_bn_bwd_finalize(st, c->rn_part, nbx, (double)M, c->grads + cv.g_off, c->grads + cv.be_off, cv.coef, cv.Cout, c->sync_buf, 1);
        if (c->sync_fn(c->sync_user, c->sync_buf, nd + 1, SELD_DTYPE_F64, st)) { c->sync_failed = true; return; }
        launch_rn_bn_bwd_finalize(st, c->rn_part, nbx, 0.0, c->grads + cv.g_off, c->grads + cv.be_off, cv.coef, cv.Cout, c->sync_buf, 2);
    } else
        launch_rn_bn_bwd_finalize(st, c->rn_part, nbx, (double)M, c->grads + cv.g_off, c->grads + cv.be_off, cv.coef, cv.Cout);
    launch_rn_bn_bwd_dz(st, cv.z, dy, mask, cv.coef, dz, M, cv.Cout, gate_z);
}

// weight gradients of the fused linear heads, on the side stream (the caller has forked): F = feat^T dy and colsum(dy) in one TN
// launch, then the four tensors of each head from them
static void heads_lin_side(seld_ctx* c, int rows) {
    const DenseL& S0 = c->heads[0].layers[0];
    const GruL& Glast = c->gru.back();
    const int nt = c->heads[0].layers[1].out + c->heads[1].layers[1].out, K = S0.in;
    int ns = 0;
    launch_gemm_tn(c->side, Glast.out, K, c->dy_all, nt, c->tn_slab_side, &ns, rows, K, nt, 0, 0, 1);
    launch_reduce_slabs2(c->side, c->tn_slab_side, ns, (int64_t)K * nt + nt, c->headF, (int64_t)K * nt, c->headF + (size_t)K * nt, nt);
    const float *w1[2], *b1[2], *w2[2];
    float *dw1[2], *db1[2], *dw2[2], *db2[2];
    int n[2];
    for (int hd = 0; hd < 2; ++hd) {
        const DenseL &L0 = c->heads[hd].layers[0], &L1 = c->heads[hd].layers[1];
        w1[hd] = c->params + L0.w_off; b1[hd] = c->params + L0.b_off; w2[hd] = c->params + L1.w_off;
        dw1[hd] = c->grads + L0.w_off; db1[hd] = c->grads + L0.b_off; dw2[hd] = c->grads + L1.w_off; db2[hd] = c->grads + L1.b_off;
        n[hd] = L1.out;
    }
    launch_heads_grad(c->side, w1, b1, w2, dw1, db1, dw2, db2, n, K, S0.out, c->headF, c->headF + (size_t)K * nt);
}

static int backward_impl(seld_ctx* c, const float* x) {
    apply_kernel_choices(c);
    hipStream_t st = c->stream;
    // test aid: injected routing decisions edit the tensors the backward kernels read their decisions from (the forward is done with them)
    for (const auto& o : c->overrides) {
        if (o.kind == 0) {
            ConvL& L = c->conv[o.block];
            const bool recorded = o.block == 0 && L.amax && (c->gram_active || (L.pf == 4 && (L.pt == 5 || L.pt == 4 || L.pt == 2 || L.pt == 1)));
            if (!recorded && !L.z) return fail(c, SELD_ERR_UNSUPPORTED, "seld_debug_set_routing: this block keeps neither recorded positions nor its pre-BN tensor");
            launch_pool_routing_patch(st, L.z, L.p, recorded ? L.amax : nullptr, L.scale, L.shift, o.idx, o.val, o.n, L.H, L.W, L.pt, L.pf);
        } else if (o.kind == 2) {      // xception_block: the ReLU in front of unit o.block's depthwise convolution
            const int b = o.block / 3, u = o.block % 3;
            const bool fold = c->xc_fused_fwd && u > 0;
            if (fold) launch_relu_gate_patch_z(st, c->xc[o.block - 1].z, nullptr, c->xc[o.block - 1].scale, c->xc[o.block - 1].scale + 64, 64, o.idx, o.val, o.n);
            else launch_relu_gate_patch(st, u == 0 ? c->xc_x[b] : c->xc[o.block - 1].a, nullptr, o.idx, o.val, o.n);
        } else if (o.kind == 3) {      // xception_block: the exit's MaxPool(ReLU(.)) over (1, 8), scanned from the last module's output
            launch_pool_routing_patch(st, c->xc_x.back(), c->xc_feat, nullptr, c->xc_ident + 128, c->xc_ident + 192, o.idx, o.val, o.n, c->S, 16, 1, 8);
        } else {
            RnBlock& R = c->rn[o.block];
            if (o.which == 2) launch_relu_gate_patch(st, R.out, R.gate, o.idx, o.val, o.n);      // read from the gate bits (and the output's sign)
            else {      // recomputed by the backward kernels from the pre-BN tensor and the forward's scale / shift (coef + 2C, + 3C)
                RnConv& K = R.c[o.which];
                launch_relu_gate_patch_z(st, K.z, o.which == 0 ? R.y0 : R.y1, K.coef + 2 * K.Cout, K.coef + 3 * K.Cout, K.Cout, o.idx, o.val, o.n);
            }
        }
    }
    const int B = c->B, S = c->S, rows = B * S;
    GruL& Glast = c->gru.back();
    // ---- heads: the input-gradient chain runs on the main stream; the weight/bias gradients only
    // feed Adam, so they go to the side stream and overlap with the BPTT chain that follows
    {
        PROF2(c, "heads_bwd");
        float* dfeat = c->feat_grad;
        DenseL &S0 = c->heads[0].layers[0], &D0 = c->heads[1].layers[0];
        if (heads_lin(c)) {
            // dfeat = [dy_sed | dy_doa] Weff^T (K = 48), then on the side stream F = feat^T dy, colsum(dy) and the four
            // gradients of each head from them
            const int nt = c->heads[0].layers[1].out + c->heads[1].layers[1].out, K = S0.in;
            launch_gemm(st, c->dy_all, nt, c->weff, nt, nullptr, dfeat, K, rows, K, nt, 1, 0, 0);
            // their weight gradients (side stream) are enqueued behind the fork that follows the last GRU layer's BPTT: one
            // cross-stream event (a ~7 us bubble on the main stream) fewer
        } else if (heads_general(c)) {
            for (int hd = 0; hd < 2; ++hd) {
                Head& Hd = c->heads[hd];
                for (int j = (int)Hd.layers.size() - 1; j >= 0; --j) {
                    DenseL& D = Hd.layers[j];
                    float* din = j == 0 ? dfeat : Hd.layers[j - 1].dy;
                    const int accumulate = (j == 0 && hd == 1) ? 1 : 0;
                    if (D.ks > 1) {
                        launch_gemm(st, D.dy, D.out, c->params + D.w_off, D.out, nullptr, c->head_tmp, D.in, rows, D.in, D.out, 1, 0, 0);
                        launch_time_fold(st, c->head_tmp, din, c->B, c->S, D.in_base, D.ks, accumulate);
                    } else {
                        launch_gemm(st, D.dy, D.out, c->params + D.w_off, D.out, nullptr, din, D.in, rows, D.in, D.out, 1, 0, accumulate);
                    }
                    if (j > 0) {
                        const DenseL& P = Hd.layers[j - 1];
                        const int64_t n = (int64_t)rows * P.out;
                        // the previous layer's dropout (the mask recomputed from the counters of the forward pass), then its activation
                        if (P.rate > 0.f && c->last_training) launch_dropout(st, din, din, n, P.rate, c->dropout_seed, P.drop_id, c->dropout_cur);
                        if (Hd.hidden_act) launch_act_bwd(st, P.y, din, n, Hd.hidden_act);
                    }
                }
            }
            fork_side(c);
            for (int hd = 0; hd < 2; ++hd) {
                Head& Hd = c->heads[hd];
                for (int j = (int)Hd.layers.size() - 1; j >= 0; --j) {
                    DenseL& D = Hd.layers[j];
                    const float* ain = D.ks > 1 ? D.xe
                                     : (j == 0 ? Glast.out : (Hd.layers[j - 1].rate > 0.f && c->last_training ? Hd.layers[j - 1].yd : Hd.layers[j - 1].y));
                    wgrad_dense(c, c->side, c->tn_slab_side, ain, D.in, D.dy, D.out, rows, D.in, D.out, D.w_off, D.b_off, 0, 0);
                }
            }
        } else {
        // the gradient w.r.t. the shared features is the sum over the two heads' first layers: one product over the
        // concatenated K axis when their shapes agree (out % 32 == 0), otherwise two launches with accumulation
        const bool merged0 = S0.in == D0.in && S0.out == D0.out && (S0.out & 31) == 0;
        for (int hd = 0; hd < 2; ++hd) {
            Head& Hd = c->heads[hd];
            for (int j = (int)Hd.layers.size() - 1; j >= (merged0 ? 1 : 0); --j) {
                DenseL& D = Hd.layers[j];
                float* din = j == 0 ? dfeat : Hd.layers[j - 1].dy;
                const int accumulate = (j == 0 && hd == 1) ? 1 : 0;
                launch_gemm(st, D.dy, D.out, c->params + D.w_off, D.out, nullptr, din, D.in, rows, D.in, D.out, 1, 0, accumulate);
                // through the hidden layer's dense_activation: the gradient w.r.t. its pre-activation, from its stored output
                if (j > 0 && Hd.hidden_act) launch_act_bwd(st, Hd.layers[j - 1].y, din, (int64_t)rows * D.in, Hd.hidden_act);
            }
        }
        if (merged0 && heads_sb(c) && gemm_sb_usable(S0.dy, S0.out, S0.in, S0.out) && gemm_sb_usable(D0.dy, S0.out, S0.in, S0.out)) {
            BwdFourScope four_;
            launch_gemm_sb(st, S0.dy, D0.dy, S0.out, c->h0sp_bwd[0], c->h0sp_bwd[1], nullptr, nullptr, dfeat, nullptr, S0.in, rows, S0.in,
                           S0.out, 0, 2);
        } else if (merged0)
            launch_gemm_dual_k(st, S0.dy, D0.dy, S0.out, c->params + S0.w_off, c->params + D0.w_off, S0.out, nullptr, dfeat, S0.in, rows,
                               S0.in, S0.out, 1, 0, 0);
        fork_side(c);
        for (int hd = 0; hd < 2; ++hd) {
            Head& Hd = c->heads[hd];
            for (int j = (int)Hd.layers.size() - 1; j >= 0; --j) {
                DenseL& D = Hd.layers[j];
                const float* ain = j == 0 ? Glast.out : Hd.layers[j - 1].y;
                wgrad_dense(c, c->side, c->tn_slab_side, ain, D.in, D.dy, D.out, rows, D.in, D.out, D.w_off, D.b_off, 0, 0);
            }
        }
        }
    }
    // ---- GRU layers, last to first
    const float* dout = c->feat_grad;
    for (int i = (int)c->gru.size() - 1; i >= 0; --i) {
        GruL& G = c->gru[i];
        const bool conv_drop = c->last_training && c->arch.conv_dropout > 0.f, gru_drop = c->last_training && c->arch.gru_dropout > 0.f;
        const float* lin = i == 0 ? (c->arch.first_kind == SELD_FIRST_XCEPTION ? c->xc_feat : (c->arch.first_kind == SELD_FIRST_RESNET50 ? c->rn.back().out : (conv_drop ? c->conv.back().pd : c->conv.back().p))) : c->gru[i - 1].out;
        {
            PROF(c, "gru_bwd");
            if (gru_drop) {
                if (launch_gru_bwd(st, dout, G.h[0], G.h[1], G.sv[0], G.sv[1], c->params + G.u_off[0], c->params + G.u_off[1], c->dgx[i][0],
                                   c->dgx[i][1], c->dgh[i][0], c->dgh[i][1], B, S, G.rmask[0], G.rmask[1], G.hm[0], G.hm[1]))
                    return fail(c, SELD_ERR_UNSUPPORTED, "gru_bwd (dropout)");
            } else
            launch_gru_bwd(st, dout, G.h[0], G.h[1], G.sv[0], G.sv[1], c->params + G.u_off[0], c->params + G.u_off[1], c->dgx[i][0],
                           c->dgx[i][1], c->dgh[i][0], c->dgh[i][1], B, S);
        }
        // the input gradient the next BPTT (or the conv backward) waits for: main stream.  Option "gru_din_first" (experiment, default 0) enqueues it BEFORE the
        // side stream is released for this layer's weight gradients, so that they do not share the card with it: same box 2.651 / 2.650 ms per step with,
        // 2.639 / 2.635 without — what the product gains the weight gradients lose under the next BPTT
        auto din_gemm = [&]() {
        {
            PROF2(c, "gru_bwd_gemms");   // main stream: the input gradient the next BPTT waits for
            // din = dgx_f K_f^T + dgx_b K_b^T: one product over the concatenated K axis (no read-modify-write of din)
            if (gru_drop) {      // din = (dgx_f K_f^T) * imask_f + (dgx_b K_b^T) * imask_b: each direction's input rows had their own mask
                for (int d = 0; d < 2; ++d) {
                    float* t_ = d == 0 ? G.din : G.dtmp;
                    if (gru_sb(c, G)) {
                        BwdFourScope four_;
                        launch_gemm_sb(st, c->dgx[i][d], nullptr, 384, c->ksp_bwd[i][d], nullptr, nullptr, nullptr, t_, nullptr, G.in_feat, rows, G.in_feat, 384, 0, 0);
                    } else
                        launch_gemm(st, c->dgx[i][d], 384, c->params + G.k_off[d], 384, nullptr, t_, G.in_feat, rows, G.in_feat, 384, 1, 0, 0);
                    launch_mask_rows(st, t_, G.imask[d], G.din, rows, S, G.in_feat, d);
                }
            } else if (gru_sb(c, G)) {
                BwdFourScope four_;
                launch_gemm_sb(st, c->dgx[i][0], c->dgx[i][1], 384, c->ksp_bwd[i][0], c->ksp_bwd[i][1], nullptr, nullptr, G.din, nullptr,
                               G.in_feat, rows, G.in_feat, 384, 0, 2);
            } else
                launch_gemm_dual_k(st, c->dgx[i][0], c->dgx[i][1], 384, c->params + G.k_off[0], c->params + G.k_off[1], 384, nullptr,
                                   G.din, G.in_feat, rows, G.in_feat, 384, 1, 0, 0);
        }
        };
        if (c->gru_din_first) din_gemm();
        // weight gradients of this layer: side stream (they overlap with the next layer's BPTT, which uses 2B of the 256 CUs)
        fork_side(c);
        if (i == (int)c->gru.size() - 1 && heads_lin(c)) heads_lin_side(c, rows);
        TnJobs tj = {};
        for (int d = 0; d < 2; ++d) {
            // kernel + input bias (bias row 0); recurrent kernel: H_prev^T dgh (forward direction saw h[t-1], backward direction
            // h[t+1]) + bias row 1
            tj.A[2 * d] = gru_drop ? G.xm[d] : lin; tj.lda[2 * d] = G.in_feat; tj.B[2 * d] = c->dgx[i][d]; tj.shift[2 * d] = 0;
            tj.out_w[2 * d] = c->grads + G.k_off[d]; tj.out_b[2 * d] = c->grads + G.b_off[d];
            tj.A[2 * d + 1] = gru_drop ? G.hm[d] : G.h[d]; tj.lda[2 * d + 1] = 128; tj.B[2 * d + 1] = c->dgh[i][d]; tj.shift[2 * d + 1] = d == 0 ? -1 : 1;
            tj.out_w[2 * d + 1] = c->grads + G.u_off[d]; tj.out_b[2 * d + 1] = c->grads + G.b_off[d] + 384;
        }
        int ns4 = 0;
        if (c->gru_wgrad_batch && c->gemm_split_bf16 && G.in_feat == 128 && launch_gemm_tn_sb_batch(c->side, tj, 4, 384, c->tn_slab_side, &ns4, rows, 384, S, 1) == 0) {
            // the layer's four products in one launch, their slabs combined by one more
            launch_reduce_slabs2_batch(c->side, c->tn_slab_side, ns4, (int64_t)128 * 384 + 384, tj, 4, (int64_t)128 * 384, 384);
        } else
            for (int j = 0; j < 4; ++j)
                wgrad_dense(c, c->side, c->tn_slab_side, tj.A[j], tj.lda[j], tj.B[j], 384, rows, j & 1 ? 128 : G.in_feat, 384,
                            tj.out_w[j] - c->grads, tj.out_b[j] - c->grads, j & 1 ? S : 0, tj.shift[j]);
        hipEventRecord(c->ev_bucket[(int)c->gru.size() - 1 - i], c->side);   // this layer's (and, for the last layer, the heads') gradients are final
        if (!c->gru_din_first) din_gemm();
        dout = G.din;
    }
    // ---- conv blocks, last to first.  dout = gradient w.r.t. the last pooled output
    const float* dp = dout;
    const bool conv_drop = c->last_training && c->arch.conv_dropout > 0.f;
    if (c->arch.first_kind == SELD_FIRST_RESNET50) {
        // ---- resnet50_block backward, blocks last to first; g = gradient w.r.t. the block's output
        PROF(c, "rn_stages_bwd");
        const bool sb = c->rn_split_bf16 != 0;
        // The kernel gradients (a third of the block's products) go to the side stream: one product of these shapes leaves the card
        // part-filled (e.g. 300 row tiles on 256 CUs), and an independent stream fills what the input-gradient chain leaves idle.
        // A dz buffer is handed over by ev_rn_ready and comes back by ev_rn_free[slot] before its next writer starts.
        const bool aside = c->rn_wgrad_side != 0;
        hipStream_t ws = aside ? c->side : st;
        bool busy[5] = {};
        int zi = 1, bbi = 4;        // last slot taken of rn_bz (0-1) / rn_bb (2-4)
        auto take = [&](int first, int n, int& cur) {
            cur = first + (cur - first + 1) % n;
            if (busy[cur]) { hipStreamWaitEvent(st, c->ev_rn_free[cur], 0); busy[cur] = false; }
            return cur < 2 ? c->rn_bz[cur] : c->rn_bb[cur - 2];
        };
        auto fork = [&](int) { if (aside) { hipEventRecord(c->ev_rn_ready, st); hipStreamWaitEvent(c->side, c->ev_rn_ready, 0); } };
        auto done = [&](int slot) { if (aside) { hipEventRecord(c->ev_rn_free[slot], c->side); busy[slot] = true; } };
        auto wgrad = [&](int slot, const float* A, int lda, const float* dz, int M_, int K1, int N, int64_t w_off) {
            fork(slot);
            launch_rn_product_wgrad(ws, A, lda, dz, c->tn_slab, tn_slab_capacity(), c->grads + w_off, M_, K1, N,
                                    c->rn_split_bf16);
            done(slot);
        };
        const float* g = dout;
        int flip = 0;
        for (int bi = (int)c->rn.size() - 1; bi >= 0; --bi) {
            if (c->sync_failed) break;     // a failed SyncBN collective: enqueue nothing further (the error is reported below)
            RnBlock& R = c->rn[bi];
            const int64_t M = (int64_t)B * S * R.Wout;
            const int w = R.w;
            const float* X = bi == 0 ? c->conv[0].p : c->rn[bi - 1].out;
            float* dX = bi == 0 ? c->conv[0].dp : c->rn_gx[flip];
            const int ldx = R.Cin * R.stride_f;
            // main branch: BN2 (behind the block's ReLU: mask = out), 1x1 expand
            float* dz2 = take(0, 2, zi);
            { PROF3(c, "rn_bn_bwd"); rn_bn_bwd(c, st, R.c[2], g, R.gate, dz2, M); }
            wgrad(zi, R.y1, w, dz2, (int)M, w, 4 * w, R.c[2].w_off);
            { PROF3(c, "rn_products_dgrad"); launch_rn_product_dgrad(st, dz2, c->params + R.c[2].w_off, sb ? R.c[2].wsp_t : nullptr, c->rn_ba, w, (int)M, w, 4 * w, 0); }
            // BN1 (mask = y1), 3x3: stage 1 on the conv blocks' kernels, the other widths through im2col / col2im
            float* dz1 = take(2, 3, bbi);
            { PROF3(c, "rn_bn_bwd"); rn_bn_bwd(c, st, R.c[1], c->rn_ba, nullptr, dz1, M); }
            if (sb && rn_c1_direct(R)) {
                fork(bbi);
                int ns = 0;
                launch_conv64_wgrad_sb(ws, R.y0, dz1, c->rn_w9_slab, &ns, B, S, rn_c1_width(R));
                if (R.c[1].w2) {
                    launch_reduce_slabs(ws, c->rn_w9_slab, ns, 9 * 4096 + 64, R.c[1].dw2, 9 * 4096, 0);
                    launch_rn_w32_extract(ws, R.c[1].dw2, c->grads + R.c[1].w_off);
                } else
                    launch_reduce_slabs(ws, c->rn_w9_slab, ns, 9 * 4096 + 64, c->grads + R.c[1].w_off, 9 * 4096, 0);
                done(bbi);
                { PROF3(c, "rn_products_dgrad"); launch_conv64_dgrad_sb(st, dz1, R.c[1].wsp9_flip, c->rn_ba, B, S, rn_c1_width(R)); }
            } else if (sb && rn_c1_implicit(c, R)) {
                fork(bbi);
                launch_rn_conv3_wgrad(ws, R.y0, dz1, c->tn_slab, tn_slab_capacity(), c->grads + R.c[1].w_off, B, S,
                                      R.Wout, w, w);
                done(bbi);
                { PROF3(c, "rn_products_dgrad"); launch_rn_conv3_dgrad(st, dz1, R.c[1].wsp_t, c->rn_ba, B, S, R.Wout, w, w); }
            } else {
                if (!R.c[1].col) return fail(c, SELD_ERR_INVALID, "resnet50_block: the options changed between forward and backward");
                if (!c->rn_bcol && dalloc(c, &c->rn_bcol, c->rn_col_elems)) return fail(c, SELD_ERR_NOMEM, "col2im tensor");
                wgrad(bbi, R.c[1].col, 9 * w, dz1, (int)M, 9 * w, w, R.c[1].w_off);
                { PROF3(c, "rn_products_dgrad"); launch_rn_product_dgrad(st, dz1, c->params + R.c[1].w_off, sb ? R.c[1].wsp_t : nullptr, c->rn_bcol, 9 * w, (int)M, 9 * w, w, 0); }
                launch_col2im3x3(st, c->rn_bcol, c->rn_ba, B, S, R.Wout, w);
            }
            // BN0 (mask = y0), 1x1 reduce; its input gradient lands on the strided rows of dX
            float* dz0 = take(2, 3, bbi);
            { PROF3(c, "rn_bn_bwd"); rn_bn_bwd(c, st, R.c[0], c->rn_ba, nullptr, dz0, M); }
            wgrad(bbi, X, ldx, dz0, (int)M, R.Cin, w, R.c[0].w_off);
            if (R.stride_f > 1) hipMemsetAsync(dX, 0, (size_t)B * S * R.Win * R.Cin * sizeof(float), st);
            // identity block: the shortcut's gated gradient g [gate] is added in this product's epilogue (split-bf16 kernels; 1 = the shape took the
            // fp32 GEMM and the separate pass below still runs)
            const bool epi_add = !R.proj && c->rn_epi_add && R.stride_f == 1;
            int added = 1;
            { PROF3(c, "rn_products_dgrad"); added = launch_rn_product_dgrad(st, dz0, c->params + R.c[0].w_off, sb ? R.c[0].wsp_t : nullptr, dX, ldx, (int)M, R.Cin, w, 0,
                                                                             epi_add ? g : nullptr, epi_add ? R.gate : nullptr); }
            if (added < 0) return fail(c, SELD_ERR_INVALID, "resnet50_block: reduce convolution's input-gradient product");
            // shortcut
            if (R.proj) {
                float* dzs = take(0, 2, zi);
                { PROF3(c, "rn_bn_bwd"); rn_bn_bwd(c, st, R.sc, g, R.gate, dzs, M); }
                wgrad(zi, X, ldx, dzs, (int)M, R.Cin, 4 * w, R.sc.w_off);
                { PROF3(c, "rn_products_dgrad"); launch_rn_product_dgrad(st, dzs, c->params + R.sc.w_off, sb ? R.sc.wsp_t : nullptr, dX, ldx, (int)M, R.Cin, 4 * w, 1); }
            } else if (!epi_add || added == 1) {
                { PROF3(c, "rn_bn_bwd"); launch_rn_add_gated(st, dX, g, R.gate, M * 4 * w); }
            }
            g = dX;
            flip ^= 1;
        }
        if (c->sync_failed) { c->sync_failed = false; return fail(c, SELD_ERR_HIP, "sync_bn all-reduce callback failed"); }
        dp = c->conv[0].dp;
    }
    if (c->arch.first_kind == SELD_FIRST_XCEPTION) {
        // ---- xception_block backward: exit pool, then the modules last to first.  gX = gradient w.r.t. the module's output
        // (= the next module's input); within a module gY walks back through the three units and the residual adds gX to it.
        const int64_t npix = (int64_t)B * S * 16;
        // three [B,S,16,64] gradient buffers: X = gradient w.r.t. the current module's output (kept until its residual add),
        // F1 = gradient w.r.t. a unit's depthwise output, F2 = gradient w.r.t. a unit's input (= the previous unit's output)
        float *X = c->xc_g[0], *F1 = c->xc_g[1], *F2 = c->xc_g[2];
        {
            PROF2(c, "xc_exit_pool_bwd");
            const float* id = c->xc_ident;       // mean 0 | invstd 1 | scale 1 | shift 0 | c1 0 | c2 0
            launch_bn_pool_bwd_dz(st, c->xc_x.back(), dout, id, id + 64, id + 128, id + 192, id + 256, X, B, S, 16, 64, 1, 8);
        }
        // The two kernel gradients of a unit (pointwise: dwo^T dz, depthwise: from the unit's input and F1) are off the input-gradient
        // chain: they run on the side stream; dz and F1 alternate between two buffers each, handed over by ev_rn_ready and taken back
        // by ev_rn_free[slot] (slots 0-1 dz, 2-3 F1) before the buffer's next writer starts.
        const bool aside = c->xc_wgrad_side != 0;
        hipStream_t ws = aside ? c->side : st;
        float* dzb[2] = {c->dzbuf, c->xc_dz2};
        float* f1b[2] = {F1, c->xc_g[3]};
        bool busy[4] = {};
        int di = 1, fi = 1;
        auto take = [&](int first, int& cur) {
            cur ^= 1;
            if (busy[first + cur]) { hipStreamWaitEvent(st, c->ev_rn_free[first + cur], 0); busy[first + cur] = false; }
            return first + cur;
        };
        auto fork = [&]() { if (aside) { hipEventRecord(c->ev_rn_ready, st); hipStreamWaitEvent(c->side, c->ev_rn_ready, 0); } };
        auto done = [&](int slot) { if (aside) { hipEventRecord(c->ev_rn_free[slot], c->side); busy[slot] = true; } };
        bool have_sums = false;      // the running unit's BatchNorm-backward partials are in xc_part_dw (n_dw_part rows)
        int n_dw_part = 0;
        struct { float* slab; int ns_pw, ns_dw; int64_t pw_off, dw_off; } pend[3];      // xc_nowait: a module's combines, launched behind its last unit
        int npend = 0;
        for (int b = (int)c->arch.xc_blocks - 1; b >= 0; --b) {
            const float* gY = X;
            for (int u = 2; u >= 0; --u) {
                XcUnit& U = c->xc[(size_t)b * 3 + u];
                const bool fold = c->xc_fused_fwd && u > 0;       // the forward applied the previous unit's BatchNormalization on load
                const float* uin = u == 0 ? c->xc_x[b] : (fold ? c->xc[(size_t)b * 3 + u - 1].z : c->xc[(size_t)b * 3 + u - 1].a);
                const float* aff = fold ? c->xc[(size_t)b * 3 + u - 1].scale : nullptr;
                int np = 0, ns = 0, ns_pw = 0;
                const bool fpw = c->xc_fused_pw_bwd != 0;
                int sd = -1;
                float* dz = nullptr;
                if (!fpw) { sd = take(0, di); dz = dzb[di]; }
                {
                    PROF2(c, "xc_bn_bwd");
                    // the sums [sum gY | sum gY xhat]: left by the depthwise input-gradient pass that produced gY (have_sums), else a pass over (z, gY)
                    if (have_sums) launch_xc_fold_partials(st, c->xc_part_dw, n_dw_part, c->xc_part, &np);
                    else launch_xc_bn_bwd_reduce(st, U.z, gY, U.mean, U.invstd, c->xc_part, &np, npix);
                    have_sums = false;
                    if (c->sync_fn) {
                        launch_bn_partials_to_sums(st, c->xc_part, np, c->sync_buf, (double)npix);
                        launch_bn_bwd_local(st, c->sync_buf, c->grads + U.g_off, c->grads + U.be_off);
                        if (c->sync_fn(c->sync_user, c->sync_buf, 129, SELD_DTYPE_F64, st)) return fail(c, SELD_ERR_HIP, "sync_bn all-reduce callback failed");
                        launch_bn_bwd_c1c2(st, c->sync_buf, 0.0 /* the all-reduced count */, U.c1c2);
                    } else
                        launch_bn_bwd_finalize(st, c->xc_part, np, (double)npix, c->grads + U.g_off, c->grads + U.be_off, U.c1c2, 64);
                    if (!fpw) launch_xc_bn_bwd_dz(st, U.z, gY, U.mean, U.invstd, U.scale, U.c1c2, dz, npix);
                }
                // xc_nowait: the default path's side-stream work reads slab buffers only, and every unit has its own: no slot to take back
                const bool nowait = fpw && c->xc_fused_dw_bwd && c->xc_nowait && c->xc_unit_slab;
                float* uslab = nowait ? c->xc_unit_slab + ((size_t)b * 3 + u) * c->xc_unit_slab_per : nullptr;
                int sf = -1;
                if (nowait) fi ^= 1; else sf = take(2, fi);
                float* F1c = f1b[fi];
                if (nowait) {
                    PROF2(c, "xc_pointwise_bwd");
                    launch_xc_pw_bwd(st, U.z, gY, U.dwo, c->params + U.pw_off, U.mean, U.invstd, U.scale, U.c1c2, F1c, uslab, &ns_pw, npix);
                } else if (fpw) {
                    PROF2(c, "xc_pointwise_bwd");
                    // dz formed on load; F1 = dz W^T and the slabs of dW = dwo^T dz from one pass (xc_pw_bwd); the combine goes to the side stream
                    if (busy[0]) { hipStreamWaitEvent(st, c->ev_rn_free[0], 0); busy[0] = false; }      // slot 0 = the slab buffer here
                    launch_xc_pw_bwd(st, U.z, gY, U.dwo, c->params + U.pw_off, U.mean, U.invstd, U.scale, U.c1c2, F1c, c->tn_slab, &ns, npix);
                    fork();
                    launch_reduce_slabs(ws, c->tn_slab, ns, 4096, c->grads + U.pw_off, 4096, 0);
                    done(0);
                } else {
                    PROF2(c, "xc_pointwise_bwd");
                    // dW = dwo^T dz (TN product over the pixels, many short splits: the slab is only 64 x 64), d(dwo) = dz W^T
                    fork();
                    launch_gemm_tn(ws, U.dwo, 64, dz, 64, c->tn_slab, &ns, (int)npix, 64, 64, 0, 0, 0, 512);
                    launch_reduce_slabs2(ws, c->tn_slab, ns, 64 * 64 + 64, c->grads + U.pw_off, 64 * 64, nullptr, 0);
                    done(sd);
                    launch_gemm(st, dz, 64, c->params + U.pw_off, 64, nullptr, F1c, 64, (int)npix, 64, 64, 1, 0, 0);
                }
                PROF2(c, "xc_depthwise_bwd");
                // gradient w.r.t. the unit's input, through its ReLU; the module's first unit adds the residual branch's X
                float* gin = (u == 0 && b == 0) ? c->conv[0].dp : F2;
                if (c->xc_fused_dw_bwd) {
                    // ... and the kernel-gradient slabs from the same pass (slab buffer fi: slot `sf` was taken above, i.e. its last combine is done)
                    float* sl = nowait ? uslab + c->xc_unit_slab_pw : c->xc_slab + (size_t)fi * c->xc_slab_per;
                    // a folded unit's input is the previous unit's pre-BN tensor and gin that BatchNormalization's output gradient: its backward sums ride along
                    const bool sums = fold && c->xc_fused_bn_sums;
                    const XcUnit* Pv = sums ? &c->xc[(size_t)b * 3 + u - 1] : nullptr;
                    if (launch_dw3x3_bwd_fused(st, F1c, c->params + U.dw_off, uin, u == 0 ? X : nullptr, gin, sl, &ns, B, S, 16, aff,
                                               sums ? Pv->mean : nullptr, sums ? Pv->invstd : nullptr, sums ? c->xc_part_dw : nullptr))
                        return fail(c, SELD_ERR_UNSUPPORTED, "dw3x3_bwd_fused");
                    if (sums) { have_sums = true; n_dw_part = ns; }
                    if (nowait) {
                        // ONE hand-over per module (an event record costs the main stream ~5 us): the three units' combines go to the side stream behind the
                        // module's last unit, each on buffers of its own
                        pend[npend++] = {uslab, ns_pw, ns, U.pw_off, U.dw_off};
                        if (u == 0) {
                            fork();
                            for (int q = 0; q < npend; ++q) {
                                float* tmp_ = pend[q].slab + c->xc_unit_slab_pw + c->xc_unit_slab_dw;
                                launch_reduce_slabs_2stage(ws, pend[q].slab, pend[q].ns_pw, 4096, c->grads + pend[q].pw_off, 4096, tmp_ + (size_t)reduce_slabs_groups(xc_dw_fused_slabs(c->Bmax, c->S)) * 576);
                                launch_reduce_slabs_2stage(ws, pend[q].slab + c->xc_unit_slab_pw, pend[q].ns_dw, 576, c->grads + pend[q].dw_off, 576, tmp_);
                            }
                            npend = 0;
                        }
                    } else {
                        fork();
                        launch_reduce_slabs_2stage(ws, sl, ns, 576, c->grads + U.dw_off, 576, c->xc_slab_tmp);      // side stream: its launches are ordered, one tmp
                        done(sf);
                    }
                } else {
                    fork();
                    launch_dw3x3_bwd_w(ws, uin, F1c, c->xc_slab, &ns, B, S, 16, aff);
                    launch_reduce_slabs(ws, c->xc_slab, ns, 576, c->grads + U.dw_off, 576, 0);
                    done(sf);
                    launch_dw3x3_bwd_data(st, F1c, c->params + U.dw_off, uin, u == 0 ? X : nullptr, gin, B, S, 16, aff);
                }
                gY = gin;
            }
            if (b > 0) { float* t_ = X; X = F2; F2 = t_; }      // the module's input gradient is the next module's output gradient
        }
        // the first block's backward (main stream) writes dzbuf: not before the side stream's last reader of it is done
        for (int k = 0; k < 4; ++k)
            if (busy[k]) hipStreamWaitEvent(st, c->ev_rn_free[k], 0);
        dp = c->conv[0].dp;
    }
    bool dz_busy[2] = {false, false};      // conv_wgrad_side: a side-stream kernel gradient reads dzbuf / dzbuf_alt (ev_rn_free[0 / 1] marks its end)
    for (int i = (int)c->conv.size() - 1; i >= 0; --i) {
        ConvL& L = c->conv[i];
        int np = 0;
        char tn[32];
        snprintf(tn, sizeof tn, "pool%d_bwd_reduce", i + 1);
        if (conv_drop) {      // through this block's Dropout: the forward's draws again (in place: dp is a buffer of this context)
            float* g_ = const_cast<float*>(dp);
            launch_dropout(st, g_, g_, (int64_t)B * (L.H / L.pt) * (L.W / L.pf) * 64, c->arch.conv_dropout, c->dropout_seed, 64u + (unsigned)i, c->dropout_cur);
        }
        {
            PROF2(c, tn);
            const bool gz = i == 0 && c->gram_active;      // no z: the windows' extreme values stand in
            if (launch_bn_pool_bwd_reduce(st, gz ? L.zext : L.z, L.p, dp, L.mean, L.invstd, L.scale, L.shift, c->bn_partial, &np, B,
                                          L.H, L.W, 64, L.pt, L.pf, gz ? 1 : 0))
                return fail(c, SELD_ERR_UNSUPPORTED, "bn_pool_bwd_reduce");
        }
        if (c->sync_fn) {
            launch_bn_partials_to_sums(st, c->bn_partial, np, c->sync_buf, (double)B * L.H * L.W);
            launch_bn_bwd_local(st, c->sync_buf, c->grads + L.g_off, c->grads + L.be_off);
            if (c->sync_fn(c->sync_user, c->sync_buf, 129, SELD_DTYPE_F64, st)) return fail(c, SELD_ERR_HIP, "sync_bn all-reduce callback failed");
            launch_bn_bwd_c1c2(st, c->sync_buf, 0.0 /* the all-reduced count */, L.c1c2);
        } else
            launch_bn_bwd_finalize(st, c->bn_partial, np, (double)B * L.H * L.W, c->grads + L.g_off, c->grads + L.be_off, L.c1c2, 64);
        int ns = 0;
        // conv_wgrad_side (round 5; same box 2.551 -> 2.523 ms): blocks 2 / 3 put their kernel gradient on the side stream (idle in this part of the step); their dz then
        // alternates between two buffers — the next block's dz is written while the side stream still reads this one's — and the slabs are the side stream's own
        const bool wside = c->conv_wgrad_side && c->prof < 2 && i >= 1 && c->dzbuf_alt && c->wgrad_slab_side;      // (a level-2 profile pass times every kernel alone)
        const int dzpar = (wside && (i & 1)) ? 1 : 0;
        float* dzb = dzpar ? c->dzbuf_alt : c->dzbuf;
        const bool fused_first = (i == 0) && L.pf == 4 && (L.pt == 5 || L.pt == 4 || L.pt == 2 || L.pt == 1);
        if (!fused_first) {
            snprintf(tn, sizeof tn, "pool%d_bwd_dz", i + 1);
            PROF2(c, tn);
            // a kernel gradient on the side stream may still read this buffer (two blocks back, or the third block's when the first block's dz goes here)
            if (dz_busy[dzpar]) { hipStreamWaitEvent(st, c->ev_rn_free[dzpar], 0); dz_busy[dzpar] = false; }
            launch_bn_pool_bwd_dz(st, L.z, dp, L.mean, L.invstd, L.scale, L.shift, L.c1c2, dzb, B, L.H, L.W, 64, L.pt, L.pf);
        }
        if (i == 0 && c->gram_active) {
            PROF(c, "conv1_wgrad");
            // dW = ka (G W + g b) + g kb + M  (conv_gram.hip): M from x, the pooled gradient and the recorded argmax
            const int kp = conv_gram_dim(L.Cin);
            if (launch_conv_first_msparse(st, x, L.p, dp, L.amax, L.scale, c->wgrad_slab, &ns, B, L.H, L.Cin))
                return fail(c, SELD_ERR_UNSUPPORTED, "conv_first_msparse");
            launch_reduce_slabs(st, c->wgrad_slab, ns, (int64_t)kp * 64, c->mmat, (int64_t)kp * 64, 0);
            hipStreamWaitEvent(st, c->ev_gram, 0);
            launch_conv_first_assemble(st, c->gram, c->mmat, c->params + L.w_off, c->params + L.b_off, L.mean, c->grads + L.w_off,
                                       c->grads + L.b_off, L.Cin);
        } else if (i == 0) {
            {
                PROF(c, "conv1_wgrad");
                // fused: dz = BN/ReLU/pool backward formed inside the wgrad kernel (L.mean.. are contiguous: 6 x 64)
                const int rc = fused_first
                    ? launch_conv_first_wgrad_fused(st, x, L.z, L.p, dp, L.amax, L.mean, c->wgrad_slab, &ns, B, L.H, L.Cin, L.pt, L.pf)
                    : launch_conv_first_wgrad(st, x, c->dzbuf, c->wgrad_slab, &ns, B, L.H, L.Cin);
                if (rc) return fail(c, SELD_ERR_UNSUPPORTED, "conv_first_wgrad");
            }
            // slab rows 0..9*Cin-1 = kernel [9*Cin][64], row 9*Cin = bias: contiguous with the flat layout
            launch_reduce_slabs(st, c->wgrad_slab, ns, conv_first_wgrad_slab_stride(L.Cin), c->grads + L.w_off,
                                (int64_t)(9 * L.Cin + 1) * 64, 0);
        } else {
            const float* lin = conv_drop ? c->conv[i - 1].pd : c->conv[i - 1].p;
            snprintf(tn, sizeof tn, "conv%d_wgrad", i + 1);
            {
                PROF2(c, tn);
                hipStream_t wst = wside ? c->side : st;
                float* wsl = wside ? c->wgrad_slab_side : c->wgrad_slab;
                if (wside) { hipEventRecord(c->ev_fork, st); hipStreamWaitEvent(c->side, c->ev_fork, 0); }      // dz (and the block's input) are final on the main stream
                if (c->conv64_split_bf16 && conv64_wgrad_sb_usable(L.W)) {
                    if (launch_conv64_wgrad_sb(wst, lin, dzb, wsl, &ns, B, L.H, L.W))
                        return fail(c, SELD_ERR_UNSUPPORTED, "conv64_wgrad_sb");
                } else if (launch_conv64_wgrad(wst, lin, dzb, wsl, &ns, B, L.H, L.W))
                    return fail(c, SELD_ERR_UNSUPPORTED, "conv64_wgrad");
                launch_reduce_slabs(wst, wsl, ns, 9 * 4096 + 64, c->grads + L.w_off, 9 * 4096 + 64, 0);
                if (wside) { hipEventRecord(c->ev_rn_free[dzpar], c->side); dz_busy[dzpar] = true; }      // this dz buffer's reader on the side stream
            }
            snprintf(tn, sizeof tn, "conv%d_dgrad", i + 1);
            {
                PROF2(c, tn);
                if (c->conv64_split_bf16) {   // flipped + split planes were made by the forward's weight pre-pass
                    launch_conv64_dgrad_sb(st, dzb, c->wsp_bwd[i], c->conv[i - 1].dp, B, L.H, L.W);
                } else {
                    launch_flip_weights(st, c->params + L.w_off, c->wflip);
                    launch_conv64_fwd(st, dzb, c->wflip, nullptr, c->conv[i - 1].dp, nullptr, nullptr, B, L.H, L.W);
                }
            }
            dp = c->conv[i - 1].dp;
        }
    }
    // the deferred loss scalars (run_losses): the side stream is ordered behind the losses kernel by every fork above
    if (c->fin_sl)
        launch_losses_finalize(c->side, c->fin_doa_loss, c->den_dev, c->fin_sl, c->fin_dl, c->loss_scratch, c->B, c->S, c->arch.n_classes);
    c->fin_sl = nullptr;
    // join: the side stream's weight gradients must be complete before Adam / the DP all-reduce
    hipEventRecord(c->ev_join, c->side);
    hipStreamWaitEvent(c->stream, c->ev_join, 0);
    return check_launch(c, "backward");
}

int seld_grads_tail_ready(seld_ctx* c, void* stream, int64_t* offset) {
    if (!c || !offset || c->gru.empty()) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    // the GRU and head variables follow the conv/BN ones in the flat buffer; their gradients are the side stream's
    // work, complete at ev_join (recorded by the last seld_train_fwd_bwd)
    *offset = c->gru[0].k_off[0];
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, c->ev_join, 0));
    return SELD_OK;
}

int seld_grads_bucket_count(const seld_ctx* c) { return c ? (int)c->gru.size() + 1 : -1; }

int seld_grads_bucket_ready(seld_ctx* c, int index, void* stream, int64_t* offset, int64_t* count) {
    if (!c || !offset || !count || index < 0 || index > (int)c->gru.size()) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = (int)c->gru.size();
    if (index == n) {               // conv / BN: the main stream's own work (+ the side stream's join)
        *offset = 0;
        *count = c->gru[0].k_off[0];
        HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, c->ev_join, 0));
        return SELD_OK;
    }
    const int layer = n - 1 - index;
    *offset = c->gru[layer].k_off[0];
    *count = (index == 0 ? c->nparam : c->gru[layer + 1].k_off[0]) - *offset;
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)stream, c->ev_bucket[index], 0));
    return SELD_OK;
}

int seld_set_sync_bn(seld_ctx* c, seld_allreduce_fn fn, void* user, int world) {
    if (!c || world < 1) return SELD_ERR_INVALID;
    c->sync_fn = fn;
    c->sync_user = user;
    c->sync_world = fn ? world : 1;
    return SELD_OK;
}

// ---------------------------------------------------------------------------------------------- data parallelism (RCCL)
// one process per GPU, a full weight replica per rank, clips sharded; the library owns the communicator.
// RCCL is bound at run time (dlopen: a process that already carries an RCCL, e.g. PyTorch's, is joined to THAT copy by its SONAME;
// a plain C host gets /opt/rocm/lib's) so that libseld_hip.so has no link-time dependency on it and loads where RCCL is absent.
int seld_dp_destroy(seld_ctx* c);
int seld_dp_available(void) { return rccl().ok ? 1 : 0; }

int seld_dp_unique_id(void* id_out) {
    if (!id_out) return SELD_ERR_INVALID;
    if (!rccl().ok) return SELD_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return SELD_ERR_HIP;
    memcpy(id_out, &id, sizeof id);
    return SELD_OK;
}

int seld_dp_init(seld_ctx* c, int rank, int world, const void* unique_id) {
    if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return SELD_ERR_INVALID;
    if (c->dp_comm) return fail(c, SELD_ERR_INVALID, "seld_dp_init: this context already has a communicator");
    if (!rccl().ok) return fail(c, SELD_ERR_UNSUPPORTED, "RCCL (librccl.so.1) could not be loaded");
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    ncclResult_t r = rccl().CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return fail(c, SELD_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    c->dp_comm = comm;
    c->dp_rank = rank;
    c->dp_world = world;
    // the communication stream reads gradients the side stream wrote and hands them back to the main stream: events with the default
    // (system-scope) release, as the bucket events have
    if (hipStreamCreateWithFlags(&c->dp_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_dp_main, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_dp_done, hipEventDisableTiming) != hipSuccess) {
        seld_dp_destroy(c);      // the communicator exists already: give it back, the context is as it was before the call
        return fail(c, SELD_ERR_HIP, "seld_dp_init: stream / event creation failed");
    }
    return SELD_OK;
}

int seld_dp_world(const seld_ctx* c) { return c ? (c->dp_comm ? c->dp_world : 1) : -1; }

int seld_dp_destroy(seld_ctx* c) {
    if (!c) return SELD_ERR_INVALID;
    if (c->dp_comm) {
        hipSetDevice(c->device);
        hipDeviceSynchronize();
        if (c->sync_fn == dp_sync_bn_fn) { c->sync_fn = nullptr; c->sync_user = nullptr; c->sync_world = 1; }
        rccl().CommDestroy(static_cast<ncclComm_t>(c->dp_comm));
        c->dp_comm = nullptr;
    }
    if (c->dp_stream) { hipStreamDestroy(c->dp_stream); c->dp_stream = nullptr; }
    if (c->ev_dp_main) { hipEventDestroy(c->ev_dp_main); c->ev_dp_main = nullptr; }
    if (c->ev_dp_done) { hipEventDestroy(c->ev_dp_done); c->ev_dp_done = nullptr; }
    c->dp_world = 1;
    return SELD_OK;
}

// The gradient all-reduce of one step, between seld_train_fwd_bwd and seld_adam_step: TWO collectives, both
// in place on the flat gradient buffer, on the library's communication stream:
//   1. GRU layers + heads (1.74 MB of the 2.06 MB): final when the FIRST GRU layer's weight-gradient products have drained on the side
//      stream (its bucket event; the side stream is in order, so the later layers' and the heads' are final too) — about 0.5 ms before the
//      backward pass ends: the conv backward runs meanwhile;
//   2. conv / BN variables: final when the main stream has drained (an event recorded here).
// The main stream then waits for the communication stream: seld_adam_step sees summed gradients.
int seld_dp_allreduce_grads(seld_ctx* c) {
    if (!c) return SELD_ERR_INVALID;
    if (!c->dp_comm) return fail(c, SELD_ERR_INVALID, "seld_dp_allreduce_grads: no communicator (seld_dp_init)");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t split = c->gru[0].k_off[0];
    HIPCHK(c, hipStreamWaitEvent(c->dp_stream, c->ev_bucket[(int)c->gru.size() - 1], 0));
    if (dp_allreduce(c, c->grads + split, c->nparam - split, SELD_DTYPE_F32, c->dp_stream)) return fail(c, SELD_ERR_HIP, "RCCL all-reduce (GRU + heads gradients) failed");
    HIPCHK(c, hipEventRecord(c->ev_dp_main, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->dp_stream, c->ev_dp_main, 0));
    if (dp_allreduce(c, c->grads, split, SELD_DTYPE_F32, c->dp_stream)) return fail(c, SELD_ERR_HIP, "RCCL all-reduce (conv / BN gradients) failed");
    HIPCHK(c, hipEventRecord(c->ev_dp_done, c->dp_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_dp_done, 0));
    return SELD_OK;
}

// Synchronised BatchNorm through the library's communicator: the per-channel sums are all-reduced on the stream the BatchNorm runs on.
// A failed collective is fatal for the process group (the peers block in theirs): the caller must abort the job.
int seld_dp_set_sync_bn(seld_ctx* c, int on) {
    if (!c) return SELD_ERR_INVALID;
    if (on && !c->dp_comm) return fail(c, SELD_ERR_INVALID, "seld_dp_set_sync_bn: no communicator (seld_dp_init)");
    return seld_set_sync_bn(c, on ? dp_sync_bn_fn : nullptr, on ? c : nullptr, on ? c->dp_world : 1);
}

int seld_train_fwd_bwd(seld_ctx* c, const float* x, const float* y_sed, const float* y_doa, const seld_loss_cfg* cfg,
                       float* sed, float* doa, float* sloss, float* dloss) {
    if (!c || !x || !y_sed || !y_doa || !cfg) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = forward_impl(c, x, sed, doa, 1, true);
    if (rc) return rc;
    // the scalar loss values are finalized on the side stream at the end of the backward pass: nothing in it waits for them
    rc = run_losses(c, y_sed, y_doa, cfg, sloss, dloss, true, true);
    if (rc) return rc;
    return backward_impl(c, x);
}

int seld_adam_step(seld_ctx* c, float lr, float beta1, float beta2, float eps, int agc) {
    if (!c) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (agc)
        for (auto& v : c->tr) launch_agc(c->stream, c->params, c->grads, v.off, v.rank, v.shape, nullptr);
    c->adam_step += 1;
    const double t = (double)c->adam_step;
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    PROF2(c, "adam");
    launch_adam(c->stream, c->params, c->grads, c->adam_m, c->adam_v, c->nparam, lr_t, beta1, beta2, eps);
    return check_launch(c, "adam");
}

int seld_train_step(seld_ctx* c, const float* x, const float* y_sed, const float* y_doa, const seld_loss_cfg* cfg,
                    float lr, int agc, float* sed, float* doa, float* sloss, float* dloss) {
    int rc = seld_train_fwd_bwd(c, x, y_sed, y_doa, cfg, sed, doa, sloss, dloss);
    if (rc) return rc;
    return seld_adam_step(c, lr, 0.9f, 0.999f, 1e-7f, agc);
}

// ---------------------------------------------------------------------------------------------- test aid
int seld_debug_pool_routing(seld_ctx* c, int block, unsigned char* pos, unsigned char* gate) {
    if (c && pos && gate && c->arch.first_kind == SELD_FIRST_XCEPTION && block == (int)c->conv.size()) {      // the exit pool of xception_block
        HIPCHK(c, hipSetDevice(c->device));
        if (launch_pool_routing(c->stream, c->xc_x.back(), c->xc_feat, nullptr, c->xc_ident + 128, c->xc_ident + 192, pos, gate, c->B, c->S, 16, 1, 8))
            return fail(c, SELD_ERR_UNSUPPORTED, "pool_routing");
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return check_launch(c, "pool_routing");
    }
    if (!c || !pos || !gate || block < 0 || block >= (int)c->conv.size()) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const ConvL& L = c->conv[block];
    // the first block routes by the positions its forward recorded (with or without the pre-BN tensor); the others by the
    // scan bn_pool_bwd_dz repeats over the stored pre-BN tensor
    const bool recorded = block == 0 && L.amax && (c->gram_active || (L.pf == 4 && (L.pt == 5 || L.pt == 4 || L.pt == 2 || L.pt == 1)));
    if (launch_pool_routing(c->stream, L.z, L.p, recorded ? L.amax : nullptr, L.scale, L.shift, pos, gate, c->B, L.H, L.W, L.pt, L.pf))
        return fail(c, SELD_ERR_UNSUPPORTED, "pool_routing");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return check_launch(c, "pool_routing");
}

static int set_override(seld_ctx* c, int kind, int block, int which, int64_t n, const int64_t* idx_host, const unsigned char* val_host, int64_t limit) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->overrides.size();)      // replace an earlier list for the same decision tensor (its buffers stay with the ctx)
        if (c->overrides[i].kind == kind && c->overrides[i].block == block && c->overrides[i].which == which) c->overrides.erase(c->overrides.begin() + i);
        else ++i;
    if (n <= 0) return SELD_OK;
    if (!idx_host || !val_host) return SELD_ERR_INVALID;
    for (int64_t k = 0; k < n; ++k)
        if (idx_host[k] < 0 || idx_host[k] >= limit) return fail(c, SELD_ERR_INVALID, "injected decision index out of range");
    seld_ctx::Override o{kind, block, which, n, nullptr, nullptr};
    if (dalloc(c, &o.idx, (size_t)n) || dalloc(c, &o.val, (size_t)n)) return SELD_ERR_NOMEM;
    HIPCHK(c, hipMemcpy(o.idx, idx_host, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(o.val, val_host, (size_t)n, hipMemcpyHostToDevice));
    c->overrides.push_back(o);
    return SELD_OK;
}

int seld_debug_set_routing(seld_ctx* c, int block, int64_t n, const int64_t* idx_host, const unsigned char* val_host) {
    if (c && c->arch.first_kind == SELD_FIRST_XCEPTION && block == (int)c->conv.size()) {      // the exit pool of xception_block
        for (int64_t k = 0; k < n && val_host; ++k)
            if (val_host[k] > 8) return fail(c, SELD_ERR_INVALID, "injected routing value exceeds 1 + the window size");
        return set_override(c, 3, block, 0, n, idx_host, val_host, (int64_t)c->Bmax * c->S * 2 * 64);
    }
    if (!c || block < 0 || block >= (int)c->conv.size()) return SELD_ERR_INVALID;
    const ConvL& L = c->conv[block];
    const int64_t limit = (int64_t)c->Bmax * (L.H / L.pt) * (L.W / L.pf) * 64;
    for (int64_t k = 0; k < n && val_host; ++k)
        if (val_host[k] > L.pt * L.pf) return fail(c, SELD_ERR_INVALID, "injected routing value exceeds 1 + the window size");
    return set_override(c, 0, block, 0, n, idx_host, val_host, limit);
}

int seld_debug_set_relu_gates(seld_ctx* c, int block, int which, int64_t n, const int64_t* idx_host, const unsigned char* val_host) {
    if (!c || which < 0 || which > 2) return SELD_ERR_INVALID;
    if (c->arch.first_kind == SELD_FIRST_XCEPTION) {      // block = unit index 3 b + u: the ReLU in front of that unit
        if (block < 0 || block >= (int)c->xc.size() || which != 0) return SELD_ERR_INVALID;
        return set_override(c, 2, block, 0, n, idx_host, val_host, (int64_t)c->Bmax * c->S * 16 * 64);
    }
    if (c->arch.first_kind != SELD_FIRST_RESNET50 || block < 0 || block >= (int)c->rn.size()) return SELD_ERR_INVALID;
    const RnBlock& R = c->rn[block];
    const int64_t limit = (int64_t)c->Bmax * c->S * R.Wout * (which == 2 ? 4 * R.w : R.w);
    return set_override(c, 1, block, which, n, idx_host, val_host, limit);
}

int seld_debug_relu_output(seld_ctx* c, int block, int which, float* dst, int64_t capacity, int64_t* count) {
    if (!c || !dst || !count || which < 0 || which > 2) return SELD_ERR_INVALID;
    if (c->arch.first_kind == SELD_FIRST_XCEPTION) {      // block = unit index: the value whose sign is the gate of the ReLU in front of that unit
        if (block < 0 || block >= (int)c->xc.size() || which != 0) return SELD_ERR_INVALID;
        HIPCHK(c, hipSetDevice(c->device));
        const int64_t n = (int64_t)c->B * c->S * 16 * 64;
        *count = n;
        if (capacity < n) return fail(c, SELD_ERR_INVALID, "seld_debug_relu_output: destination too small");
        const int b = block / 3, u = block % 3;
        const bool fold = c->xc_fused_fwd && u > 0;
        launch_affine_copy(c->stream, u == 0 ? c->xc_x[b] : (fold ? c->xc[block - 1].z : c->xc[block - 1].a), fold ? c->xc[block - 1].scale : nullptr, dst, n, 64);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return check_launch(c, "relu_output");
    }
    if (c->arch.first_kind != SELD_FIRST_RESNET50 || block < 0 || block >= (int)c->rn.size()) return SELD_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const RnBlock& R = c->rn[block];
    const int64_t M = (int64_t)c->B * c->S * R.Wout, n = M * (which == 2 ? 4 * R.w : R.w);
    *count = n;
    if (capacity < n) return fail(c, SELD_ERR_INVALID, "seld_debug_relu_output: destination too small");
    const float* src = which == 0 ? R.y0 : (which == 1 ? R.y1 : R.out);
    HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SELD_OK;
}

// ---------------------------------------------------------------------------------------------- profiling
int seld_profile_enable(seld_ctx* c, int on) {
    if (!c) return SELD_ERR_INVALID;
    c->prof = on < 0 ? 0 : (on > 3 ? 3 : on);
    if (c->prof) {      // events for a default bench run of scopes up front; prof_resolve (seld_profile_get / _reset) recycles them
        hipSetDevice(c->device);
        while (c->ev_pool.size() < (c->prof == 1 ? 1024u : (c->prof == 2 ? 8192u : 32768u))) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; c->ev_pool.push_back(e); }
    }
    return SELD_OK;
}
int seld_profile_count(const seld_ctx* c) { return c ? (int)c->timers.size() : -1; }
static void prof_resolve(seld_ctx* c) {
    hipStreamSynchronize(c->stream);
    for (auto& t : c->timers) {
        for (size_t i = 0; i + 1 < t.ev.size(); i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]) == hipSuccess) t.ms += ms;
        }
        for (auto e : t.ev) c->ev_pool.push_back(e);     // recycled by the next timed scopes
        t.ev.clear();
    }
}
int seld_profile_get(seld_ctx* c, int index, char* name, int name_cap, int64_t* launches, double* total_ms) {
    if (!c || index < 0 || index >= (int)c->timers.size()) return SELD_ERR_INVALID;
    prof_resolve(c);
    Timer& t = c->timers[index];
    if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = t.launches;
    if (total_ms) *total_ms = t.ms;
    return SELD_OK;
}
int seld_profile_reset(seld_ctx* c) {
    if (!c) return SELD_ERR_INVALID;
    prof_resolve(c);
    c->timers.clear();
    return SELD_OK;
}

}  // extern "C"
