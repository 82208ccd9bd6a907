/* seld_hip.h — C ABI of libseld_hip.so, the MI355X (gfx950) SELDnet hot path.
 *
 * The reference (IRIS-AUDIO/SELD) has no FFI; its seam is Python duck typing.  Every entry
 * point below names the reference interface it replaces (file:line under the reference tree).
 * The Python package `seld_amd` binds these with ctypes and mirrors the reference's
 * models.seldnet / train.trainstep / train.teststep surface on top of them.
 *
 * Conventions
 *  - every call returns int: 0 = ok, <0 = error (SELD_ERR_*); seld_last_error() has the text;
 *    no exception crosses the ABI.
 *  - pointers are DEVICE pointers unless the name ends in _host.  fp32 everywhere on the ABI.
 *  - the ctx owns weights, gradients, Adam slots, BN moving statistics and all workspace
 *    (sized at seld_create for a fixed [B,T,F,C]); the caller owns x / labels / outputs.
 *  - work is enqueued on the ctx's stream (default: the null stream; seld_set_stream to change)
 *    and is asynchronous; seld_sync() waits.  Calls on one ctx are not thread-safe; distinct
 *    ctxs are independent (one ctx per GPU / rank).
 *  - layouts are the reference's: x [B,T,F,C] (NHWC, H=time), conv kernels HWIO, GRU kernels
 *    [in,3u] gate order z|r|h with bias [2,3u], labels sed [B,S,nc], doa [B,S,3nc] as x|y|z blocks.
 */
#ifndef SELD_HIP_H
#define SELD_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SELD_OK 0
#define SELD_ERR_INVALID (-1)     /* bad argument */
#define SELD_ERR_UNSUPPORTED (-2) /* valid config this build has no kernel for */
#define SELD_ERR_HIP (-3)         /* HIP runtime error */
#define SELD_ERR_NOMEM (-4)

#define SELD_DTYPE_F32 0
/* BASELINE.json configs[1]'s literal "bf16": every dense product of the step that has a single-product kernel (the three conv blocks'
 * forward / input gradient / kernel gradient, the GRU input projections and their input and kernel gradients) takes ONE bf16 MFMA product
 * with both operands rounded to nearest-even bf16 and fp32 accumulation, instead of the six products of the exact 3-way split that
 * SELD_DTYPE_F32 uses; tensors, BatchNorm statistics, the GRU recurrence, losses and Adam stay fp32.  Not within north_star's 1e-4: the
 * measured distance from the fp64 oracle is documented in DESIGN.md (outputs ~1e-3, gradients ~1e-2 of a variable's maximum).  The same
 * switch at run time: seld_set_option(ctx, "bf16_single", 0 | 1).  (2: SELD_DTYPE_F64 = 1 names the SyncBN callback's buffer type.) */
#define SELD_DTYPE_BF16 2

#define SELD_DOA_MSE 0  /* tf.keras.losses.MSE function form (train.py:317-320): [B,S] rows, tape sums them */
#define SELD_DOA_MMSE 1 /* losses.MMSE (losses.py:4-13) */
#define SELD_DOA_MAE 2  /* tf.keras.losses.MAE function form (params.py:16-17 choice 'MAE', train.py:317-318) */
#define SELD_DOA_MSLE 3 /* tf.keras.losses.MSLE function form (choice 'MSLE'): mean((log(max(p,1e-7)+1) - log(max(y,1e-7)+1))^2) */

#define SELD_MAX_LAYERS 4

typedef struct seld_ctx seld_ctx;

/* Architecture = what models.seldnet (models.py:18-32) reads from model_config/seldnet.json. */
typedef struct seld_arch {
    int32_t in_ch;                      /* C: 7 (foa) */
    int32_t n_freq;                     /* F: 64 */
    int32_t n_conv;                     /* len(FIRST_ARGS.filters) */
    int32_t filters[SELD_MAX_LAYERS];
    int32_t pool_t[SELD_MAX_LAYERS];    /* FIRST_ARGS.pool_size[i][0] */
    int32_t pool_f[SELD_MAX_LAYERS];    /* FIRST_ARGS.pool_size[i][1] */
    int32_t n_gru;                      /* len(SECOND_ARGS.units) */
    int32_t gru_units[SELD_MAX_LAYERS];
    int32_t n_sed_dense;                /* len(SED_ARGS.units) */
    int32_t sed_units[SELD_MAX_LAYERS];
    int32_t n_doa_dense;
    int32_t doa_units[SELD_MAX_LAYERS];
    int32_t n_classes;                  /* 12 (train.py:306-307) */
    /* FIRST block kind (models.py:24 getattr(modules, model_config['FIRST'])): 0 = simple_conv_block (filters / pool_* above),
     * 1 = xception_block (model_config/xception_gru.json; spec/XCEPTION_BLOCK.md): n_conv = 1 entry conv2d_bn(filters[0] = 2 x
     * FIRST_ARGS.filters) with pool (5,4), then xc_blocks = FIRST_ARGS.block_num residual modules of three
     * SeparableConv2D + BatchNormalization, then ReLU + MaxPooling2D((1,8)). */
    int32_t first_kind;
    int32_t xc_blocks;
    /* 2 = resnet50_block (model_config/resnet50_gru.json; spec/RESNET50_BLOCK.md): the same entry, then four stages of rn_blocks[s]
     * bottleneck blocks of width rn_filters * 2^s, frequency stride 2 at the first block of stages 1..3 -> [B, T/5, 2, 32 rn_filters] */
    int32_t rn_filters;
    int32_t rn_blocks[4];
    /* simple_dense_block's `dense_activation` (modules.py:356, 368-371): the activation of the heads' HIDDEN Conv1D layers (the output
     * Dense layers keep sigmoid / tanh, models.py:28-30).  0 = None / 'linear' (seldnet.json; lets W1 W2 fold into one product),
     * SELD_ACT_SIGMOID, SELD_ACT_TANH, SELD_ACT_RELU. */
    int32_t sed_dense_act;
    int32_t doa_dense_act;
    /* simple_dense_block's `kernel_size` (modules.py:355, 370-372: Conv1D(units, kernel_size, padding='same') over the frames of a clip;
     * 0 or 1 = per-step, seldnet.json) and `dropout_rate` (modules.py:357, 373-374: Dropout after every hidden layer, training only) of the
     * two heads.  The dropout masks are a counter-based function of (option "dropout_seed", the training step counter, layer, element):
     * see loss_adam.hip::dropout_kernel; TensorFlow's generator cannot be reproduced by another program. */
    int32_t sed_kernel_size;
    int32_t doa_kernel_size;
    float sed_dropout;
    float doa_dropout;
    /* 1 = models.seldnet_v1 (models.py:36-52; model_config/seldnet_v1.json): doa_out = tanh(doa * Concatenate([sed] * 3)). */
    int32_t output_coupling;
    /* `dropout_rate` of the FIRST and SECOND blocks (0 in every shipped config), training only; masks drawn like the heads' (dropout_kernel).
     * conv_dropout: simple_conv_block's Dropout(rate) behind every MaxPooling2D (model_config/seldnet.json:7; first_kind 0 only).
     * gru_dropout: bidirectional_GRU_block passes it as BOTH `dropout` and `recurrent_dropout` of every GRU (modules.py:306, 312-314): per
     * direction one input mask [B, in_feat] and one state mask [B, units], values 0 | 1/(1-rate), constant over the sequence (Keras GRUCell,
     * implementation 2: the input is masked before the kernel product, the previous state before the recurrent product AND the blend). */
    float conv_dropout;
    float gru_dropout;
} seld_arch;
#define SELD_ACT_NONE 0
#define SELD_ACT_SIGMOID 1
#define SELD_ACT_TANH 2
#define SELD_ACT_RELU 3
/* sizeof(seld_arch), sizeof(seld_loss_cfg) as the LIBRARY was compiled, into out[0..min(n,2)); returns how many it knows (2).  A binding
 * compares them with its own struct declarations before the first seld_create. */
int seld_abi_sizes(int32_t* out, int n);
#define SELD_FIRST_SIMPLE_CONV 0
#define SELD_FIRST_XCEPTION 1
#define SELD_FIRST_RESNET50 2
#define SELD_MAX_XC_BLOCKS 16

/* Loss configuration = train.py:311-320 + the `loss_weight` flag (params.py:30). */
typedef struct seld_loss_cfg {
    int32_t doa_loss;        /* SELD_DOA_MSE | SELD_DOA_MMSE | SELD_DOA_MAE | SELD_DOA_MSLE */
    float w_sed, w_doa;      /* loss_weight "1,1000" */
    float sed_grad_scale;    /* 1 on a single device; 1/world for data parallel MMSE runs */
    float mmse_den;          /* <=0: sum(mask) of this batch; >0: all-reduced global denominator */
} seld_loss_cfg;

/* ---- lifecycle: replaces getattr(models, 'seldnet')(input_shape, model_config) (train.py:308) */
int seld_create(const seld_arch* arch, int B, int T, int dtype, int device, seld_ctx** out);
void seld_destroy(seld_ctx* ctx);
const char* seld_last_error(const seld_ctx* ctx); /* ctx may be NULL: last create() error */
int seld_set_stream(seld_ctx* ctx, void* hip_stream);
/* batch size of the next calls, 1 <= B <= the B given to seld_create (the reference's Keras model
 * accepts any batch; data_loader.batch(drop_remainder=False) yields a short last batch) */
int seld_set_batch(seld_ctx* ctx, int B);
int seld_sync(seld_ctx* ctx);
/* compute options.  "conv64_split_bf16" (default 1): conv2/conv3 forward and input gradient run on the bf16 matrix
 * cores with every fp32 operand split exactly into three bf16 values and six partial products (fp32-level accuracy);
 * 0 selects the f32-input MFMA kernels.  "gemm_split_bf16" (default 1): the same scheme for the GRU input projections,
 * the heads' first Conv1D and their input gradients (gemm_sb.hip) where K % 32 == 0 and N % 128 == 0; 0 keeps them on
 * the f32-input MFMA GEMM.  "conv1_split_bf16" (default 1): the same scheme for the first
 * block's forward whenever its pre-BN tensor is not stored (conv_pool_sb.hip).  "conv1_pool_fused" / "conv1_gram" (default 1): first block's pooling inside the conv
 * epilogue / its kernel gradient from the patch Gram matrix.  "dropout_seed" / "dropout_step" (values): the key of the heads' dropout
 * draws and the step counter of the next training forward (seld_arch.sed_dropout / doa_dropout; INTEGRATION.md section 6 lists every key).
 * EVERY key is per context: the kernel choices the launchers read from library-wide variables ("bwd_four_products", "gru_var" 0..255,
 * "conv64_dbuf", "tn_tile_blocks", "tn_lds_floor", "gram_bg_blocks", "bf16_single") are stored in the context and copied into those
 * variables at the start of each forward / backward pass, so a six-product context and a four-product context coexist in one process
 * (calls on one context are not thread-safe; two contexts driven from two threads at once are not supported for differing choices). */
int seld_set_option(seld_ctx* ctx, const char* key, int value);

/* ---- variables: replaces model.trainable_variables / get_weights / set_weights
 * (train.py:31-34, evaluator.py:57).  Flat fp32, Keras creation order; seld_variable_info
 * enumerates (name, offset, shape).  "state" = BN moving_mean/moving_variance pairs. */
int64_t seld_param_count(const seld_ctx* ctx);
int64_t seld_state_count(const seld_ctx* ctx);
int seld_variable_count(const seld_ctx* ctx, int trainable);
int seld_variable_info(const seld_ctx* ctx, int trainable, int index, char* name, int name_cap,
                       int64_t* offset, int32_t* rank, int64_t shape[4]);
int seld_set_weights_host(seld_ctx* ctx, const float* w_host, int64_t n);
int seld_get_weights_host(seld_ctx* ctx, float* w_host, int64_t n);
int seld_set_state_host(seld_ctx* ctx, const float* s_host, int64_t n);
int seld_get_state_host(seld_ctx* ctx, float* s_host, int64_t n);
int seld_get_grads_host(seld_ctx* ctx, float* g_host, int64_t n);
int seld_get_adam_host(seld_ctx* ctx, float* m_host, float* v_host, int64_t n);
int seld_set_adam_host(seld_ctx* ctx, const float* m_host, const float* v_host, int64_t n, int64_t step);
void* seld_param_ptr(seld_ctx* ctx); /* device, [param_count] fp32 */
void* seld_grad_ptr(seld_ctx* ctx);  /* device, [param_count] fp32: the DP all-reduce buffer */

/* ---- model(x, training) (models.py:18-32; train.py:25,41).  training!=0: batch statistics,
 * moving statistics updated.  sed [B,S,nc], doa [B,S,3nc], S = T / prod(pool_t). */
int seld_forward(seld_ctx* ctx, const float* x, float* sed, float* doa, int training);

/* Data-parallel overlap: after seld_train_fwd_bwd has been ENQUEUED, make `stream` wait until the gradients of the
 * GRU and head variables — grads[*offset : param_count), 86 % of the buffer — are final.  They are produced on the
 * library's side stream early in the backward pass, so a host can start their all-reduce on a communication stream
 * while the conv backward is still running on the main stream; grads[0 : *offset) (conv/BN) are final when the main
 * stream has drained.  (No counterpart in the reference, which has no distributed path.) */
int seld_grads_tail_ready(seld_ctx* ctx, void* stream, int64_t* offset);
/* The same hand-off at the granularity the backward pass produces gradients in: bucket 0 = the last GRU layer + the heads
 * (final when that layer's BPTT has run and its weight-gradient GEMMs have drained on the side stream, i.e. while the earlier
 * layers' recurrences are still running), bucket k = GRU layer n_gru-1-k, the last bucket = the conv/BN variables (final when
 * the main stream has drained).  seld_grads_bucket_ready makes `stream` wait for bucket `index` of the last ENQUEUED
 * seld_train_fwd_bwd and returns its [offset, offset + count) range of the flat gradient buffer. */
int seld_grads_bucket_count(const seld_ctx* ctx);
int seld_grads_bucket_ready(seld_ctx* ctx, int index, void* stream, int64_t* offset, int64_t* count);
/* ---- synchronised BatchNorm for data parallelism (a B/world-per-rank run then equals the reference's
 * single-device batch, layers.py:33 with params.py:27's batch of 256).  The library calls `fn(user, buf, count, dtype, stream)`
 * — dtype SELD_DTYPE_F64, count 128 * ceil(C / 64) + 1 = [sum z | sum z^2] resp. [sum dy | sum dy xhat] per channel (128 for the 64-channel
 * blocks, up to 2048 for resnet50_block's 1024-channel BatchNormalizations) followed by THIS rank's element count (so ranks may hold
 * different numbers of clips: the global count is the all-reduced one) — once per BatchNormalization in the
 * training forward and once in the backward; fn must enqueue an in-place SUM over the `world` ranks of the device buffer `buf`
 * on `stream` (e.g. ncclAllReduce / torch.distributed.all_reduce) and return 0.  fn == NULL switches back to per-replica
 * statistics.  dgamma / dbeta stay per-rank partial sums like every other gradient (the gradient all-reduce completes them). */
#define SELD_DTYPE_F64 1
typedef int (*seld_allreduce_fn)(void* user, void* buf, int64_t count, int dtype, void* hip_stream);
int seld_set_sync_bn(seld_ctx* ctx, seld_allreduce_fn fn, void* user, int world);
/* ---- data parallelism INSIDE the library: one process per GPU, a full weight replica per rank, the
 * batch sharded by clips; the reference has no counterpart (train.py:22-36 is single device).  The library owns an RCCL communicator
 * (RCCL is bound with dlopen at the first seld_dp_* call: no link-time dependency), a communication stream and the bucket order.
 *   seld_dp_available()              : 1 if RCCL could be bound in this process, else 0 — what every rank OTHER than 0 calls before the
 *                                      host's agreement step (ncclGetUniqueId starts a bootstrap listener: only rank 0 draws an id)
 *   seld_dp_unique_id(id)            : rank 0 fills 128 bytes (ncclUniqueId); the host carries them to every rank by its own means
 *   seld_dp_init(ctx, rank, world, id): ncclCommInitRank on the ctx's device (collective: every rank calls it)
 *   seld_dp_allreduce_grads(ctx)     : between seld_train_fwd_bwd and seld_adam_step — TWO in-place ncclAllReduce(SUM) over the flat
 *                                      gradient buffer: GRU + heads as soon as those gradients are final (the conv backward is still
 *                                      running), then the conv / BN variables; the ctx stream waits for both
 *   seld_dp_set_sync_bn(ctx, on)     : synchronised BatchNorm (see seld_set_sync_bn) through the same communicator
 *   MMSE loss: with a communicator and cfg.mmse_den <= 0 the mask count is all-reduced on the device inside seld_train_fwd_bwd /
 *              seld_test_step; the caller sets cfg.sed_grad_scale = 1 / world (loss-reduction rules: DESIGN.md section 5)
 * A failed collective is fatal for the process group: peers block in theirs — abort the job. */
int seld_dp_available(void);
int seld_dp_unique_id(void* id_out_128_bytes);
int seld_dp_init(seld_ctx* ctx, int rank, int world, const void* unique_id_128_bytes);
int seld_dp_world(const seld_ctx* ctx);
int seld_dp_allreduce_grads(seld_ctx* ctx);
int seld_dp_set_sync_bn(seld_ctx* ctx, int on);
int seld_dp_destroy(seld_ctx* ctx);
/* ---- train.trainstep (train.py:22-36), split so that a data-parallel host can all-reduce
 * seld_grad_ptr() between the two halves:
 *   seld_train_fwd_bwd : forward(training=True) + losses + tape.gradient -> grad buffer
 *   seld_adam_step     : optional AGC (utils.py:86-96) + Adam.apply_gradients (Keras Adam, eps 1e-7)
 * sloss: 1 float; dloss: [B*S] floats (MSE) or 1 float (MMSE); either may be NULL. */
int seld_train_fwd_bwd(seld_ctx* ctx, const float* x, const float* y_sed, const float* y_doa,
                       const seld_loss_cfg* cfg, float* sed, float* doa, float* sloss, float* dloss);
int seld_adam_step(seld_ctx* ctx, float lr, float beta1, float beta2, float eps, int agc);
int seld_train_step(seld_ctx* ctx, const float* x, const float* y_sed, const float* y_doa,
                    const seld_loss_cfg* cfg, float lr, int agc, float* sed, float* doa,
                    float* sloss, float* dloss);
/* ---- train.teststep (train.py:39-44) */
int seld_test_step(seld_ctx* ctx, const float* x, const float* y_sed, const float* y_doa,
                   const seld_loss_cfg* cfg, float* sed, float* doa, float* sloss, float* dloss);
/* sum(mask) of losses.MMSE for this batch (1 float, device) — for the DP denominator all-reduce */
int seld_mmse_den(seld_ctx* ctx, const float* y_doa, float* den);

/* ---- feature stage: replaces feature_extractor.extract_features (feature_extractor.py:53-88) and, for the
 * in-loop variant, apply_normalizer + preprocess_features_labels (:117-149, :226-234).
 * complex_spec (:153-173) = torchaudio spectrogram(hann(win_length) periodic, n_fft, hop, power=None, center, reflect);
 * mode 0 'foa': 4 log-mel (top_db 80 over the clip) + 3 mel intensity vectors; mode 1 'mic': 4 log-mel + 6 GCC-PHAT.
 * wav [4][n_samples] fp32 (device) -> out [1 + n_samples/hop][n_mels][7|10] fp32 (device), the reference's
 * [time, freq, chan] layout.  `stream` is a hipStream_t (NULL = null stream). */
typedef struct seld_feat seld_feat;
int seld_feat_create(int sample_rate, int n_fft, int win_length, int hop_length, int n_mels, int mode, int normalized,
                     int device, seld_feat** out);
void seld_feat_destroy(seld_feat* f);
const char* seld_feat_last_error(const seld_feat* f);
/* kernel selection, for A/B runs and parity of the fallback: "wave_kernel" (default 1): one wave per frame with a radix-4 FFT
 * in registers + LDS (n_fft 256 .. 1024); 0: the workgroup-per-frame radix-2 kernel that serves every other n_fft */
int seld_feat_set_option(seld_feat* f, const char* key, int value);
int64_t seld_feat_frames(const seld_feat* f, int64_t n_samples);
int seld_feat_channels(const seld_feat* f);
int seld_feat_extract(seld_feat* f, const float* wav, int n_ch, int64_t n_samples, float* out, void* stream);
/* The same for `n_clips` clips of ONE length in one pair of launches (what a data loader preprocessing a list of equal-length files,
 * feature_extractor.py:238-262, does clip by clip): wav [n_clips][n_ch][n_samples] -> out [n_clips][frames][n_mels][7|10]; the
 * top_db clamp stays per clip.  Launch overheads and the tail of a 3 001-frame grid amortise over the batch. */
int seld_feat_extract_batch(seld_feat* f, const float* wav, int n_clips, int n_ch, int64_t n_samples, float* out, void* stream);
/* (x - mean)/max(std, eps) per (freq, chan), rows trimmed / zero-padded (before normalising, as the reference does) to T_out */
int seld_feat_normalize(const float* feat, const float* mean, const float* stdv, float* out, int64_t T_in, int64_t T_out,
                        int FC, float eps, void* stream);
/* calculate_statistics (feature_extractor.py:218-224): per-(freq, chan) mean and population std over ALL frames of a list of feature
 * tensors.  `acc` = 2*FC + 1 doubles on the device (sum | sum of squares | row count; zero it to start), folded once per tensor
 * feat [rows][FC] (any number of calls, any row counts: a file, or a batch [n][T][FC] as n*T rows); `scratch` =
 * seld_feat_stats_scratch_doubles(FC) doubles (device).  Fixed-order double sums: bit-reproducible for a given call sequence.
 * seld_feat_stats_finalize writes float mean / std [FC] (the reference's [1, freq, chan] arrays). */
int64_t seld_feat_stats_scratch_doubles(int FC);
int seld_feat_stats_accumulate(const float* feat, int64_t rows, int FC, double* acc, double* scratch, void* stream);
int seld_feat_stats_finalize(const double* acc, int FC, float* mean, float* stdv, void* stream);

/* ---- sliding-window inference: the two tensor ops around model() in evaluator.ensemble_outputs
 * (evaluator.py:16-50, trainv2.py:158-192).
 * seld_frame_windows: tf.signal.frame(x [T,FC], win_size, step, axis=0)[first_window : first_window+n_windows]
 *                     -> windows [n_windows, win_size, FC]   (FC = freq*chan, multiple of 4)
 * seld_overlap_average: tf.signal.overlap_and_add(y [n_windows, L, D], frame_step=1) / window count -> [n_windows-1+L, D] */
int seld_frame_windows(const float* x, float* windows, int T, int FC, int win_size, int step, int first_window, int n_windows,
                       void* stream);
int seld_overlap_average(const float* y, float* out, int n_windows, int L, int D, void* stream);

/* ---- batch augmentation on the device (reference: the tf.data sample/batch transforms of train.get_dataset,
 * train.py:157-165).  The random draws are made by the host (seld_amd/transforms.py mirrors the reference's
 * distributions) and passed as small device arrays; the kernels are pure data movement.
 * seld_aug_mask: transforms.mask (transforms.py:6-44) for both axes in one pass over x [B,T,F,C], in place: for sample
 *   b and segment s = t / period (T % period == 0, else SELD_ERR_INVALID as the reference raises ValueError), frames
 *   [t_off, t_off+t_size) of the segment and bins [f_off, f_off+f_size) are zeroed; arrays are int32 [B * T/period];
 *   either pair may be NULL (that axis is not masked).
 * seld_aug_gather_sign: out[b,o,r,i] = sgn[b,r] * in[b,o,src[b,r],i] in place on x [B,outer,R,inner] (R <= 32): the
 *   channel permutation + sign flips of foa_intensity_vec_aug / acs_aug (transforms.py:73-114,159-207) on the
 *   features (R = channels, inner = 1) and on the labels viewed [B,S,4,n_classes] (R = 4, inner = n_classes). */
int seld_aug_mask(float* x, int B, int T, int F, int C, int period, const int* t_off, const int* t_size, const int* f_off,
                  const int* f_size, void* stream);
int seld_aug_gather_sign(float* x, int B, int64_t outer, int R, int64_t inner, const int* src, const float* sgn, void* stream);

/* ---- SELD metrics on the device: SELDMetrics.update_states (metrics.py:60-154), which the reference runs in TF
 * eager mode on the host after every step (train.py:82-83).  `state` = seld_metrics_state_size(nc) doubles
 * (device, zero to reset): TP FP TN FN S D I Nref Nsys total_DE DE_TP, then class_tp|fp|tn|fn [nc] each;
 * `scratch` = seld_metrics_scratch_floats(...) floats (device).  y_true / y_pred = (sed [B,S,nc], doa [B,S,3nc]). */
int seld_metrics_state_size(int n_classes);
int64_t seld_metrics_scratch_floats(int B, int S, int n_classes, int block_size);
int seld_metrics_update(const float* sed_true, const float* doa_true, const float* sed_pred, const float* doa_pred, int B, int S,
                        int n_classes, int block_size, float doa_threshold, double* state, float* scratch, void* stream);

/* ---- measurement: HIP-event timing of named kernel groups on the ctx stream (bench.py roofline).
 * seld_profile_enable(ctx, level): 0 off, 1 the largest groups (conv1 fwd / conv1 wgrad / GRU fwd / GRU BPTT, the resnet stages), 2 every group,
 * 3 additionally one scope per product / BatchNorm launch of resnet50_block (hundreds of event pairs per step: a profile pass, not a timed one) */
int seld_profile_enable(seld_ctx* ctx, int on);
int seld_profile_count(const seld_ctx* ctx);
int seld_profile_get(seld_ctx* ctx, int index, char* name, int name_cap, int64_t* launches, double* total_ms);
int seld_profile_reset(seld_ctx* ctx);

/* ---- per-kernel entry points (unit parity tests; all device pointers, null stream) --------
 * Each cites what it computes in the reference.  Shapes are checked; SELD_ERR_UNSUPPORTED if
 * the build has no kernel for them. */
/* same keys as seld_set_option, for the seld_k_* entry points: PROCESS-WIDE (there is no context).  Defaults: the product's
 * ("bwd_four_products" 1: the unit entry points' backward products also run on four of the six split terms); a forward / backward pass of
 * any context overwrites them with that context's choices, so set them right before the seld_k_* calls they are meant for. */
int seld_k_set_option(const char* key, int value);
/* Conv2D(64, 3, padding='same', use_bias=True) on NHWC (layers.py:27-32); x [B,H,W,Cin], w HWIO, z [B,H,W,64].
 * stats (may be NULL): [2*64] = per-channel sum(z), sum(z^2) over B*H*W (BatchNormalization batch statistics). */
int seld_k_conv3x3_fwd(const float* x, const float* w, const float* bias, float* z, float* stats,
                       int B, int H, int W, int Cin, int Cout);
/* First conv block forward with the (5,4) MaxPooling2D reduction folded into the convolution (layers.py:27-37,
 * seldnet.json FIRST_ARGS pool_size[0]): x [B,H,64,Cin] (Cin 7 or 10, H % 5 == 0) -> z [B,H,64,64] (may be NULL:
 * not stored), zext [B,H/5,16,64] = per pooling window max(z) where gamma >= 0 / min(z) where gamma < 0, amax
 * [B,H/5,16,64] bytes = position row*4+col of that extreme inside its window (what MaxPoolGrad routes to; the first
 * in column-then-row scan order on ties; may be NULL only if z is NULL), and the batch statistics as in
 * seld_k_conv3x3_fwd.  BN+ReLU is monotone in z, so MaxPool(ReLU(BN(z))) == seld_k_bn_relu_ext(zext) bit for bit. */
int seld_k_conv_first_fwd_pool(const float* x, const float* w, const float* bias, const float* gamma, float* z, float* zext,
                               unsigned char* amax, float* stats, int B, int H, int Cin);
/* The first block trained WITHOUT its pre-BN tensor (conv_gram.hip): z = P W + b is linear in the im2col patches P, so
 * the kernel/bias gradient through BatchNorm(batch statistics)+ReLU+MaxPool(5,4) is
 *   dW = ka (G W + g b) + g kb + M,   G = P^T P (Gram matrix of the input patches), g = column sums of P,
 *   M[k][co] = sum over windows with p > 0 of scale[co] dp P[recorded argmax pixel][k]
 * seld_k_conv1_gram: G alone, [KP x KP] floats with KP = 64 (Cin 7) / 128 (Cin 10); row/column 9*Cin is the ones column
 *   (G[k][9 Cin] = g[k], G[9 Cin][9 Cin] = B*H*64); only the upper 32x32 tiles are written (symmetric).
 * seld_k_conv1_train_gram: the whole block from x: forward (window extremes, positions, statistics, p) and backward
 *   (dgamma, dbeta, dw [9*Cin*64], db [64]) for a given dp [B,H/5,16,64]; replaces tape.gradient through
 *   conv2d_bn + MaxPooling2D of the first block (layers.py:27-37). */
int seld_k_conv1_gram(const float* x, float* G, int B, int H, int Cin);
int seld_k_conv1_train_gram(const float* x, const float* w, const float* bias, const float* gamma, const float* beta,
                            const float* dp, float* p, float* dw, float* db, float* dgamma, float* dbeta, int B, int H, int Cin);
/* p = max(0, zext * scale[c] + shift[c]) (fused multiply-add), n elements, 64 channels innermost */
int seld_k_bn_relu_ext(const float* zext, const float* scale, const float* shift, float* p, int64_t n);
/* input gradient of the same conv (tape.gradient through Conv2D): dz [B,H,W,64] -> dx [B,H,W,Cin=64] */
int seld_k_conv3x3_dgrad(const float* dz, const float* w, float* dx, int B, int H, int W, int Cin, int Cout);
/* kernel+bias gradient of the same conv: dw HWIO, db [Cout] */
int seld_k_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* db, int B, int H, int W, int Cin, int Cout);
/* first conv block backward in one fused pass: given the forward's pre-BN z [B,H,64,64], the batch mean/invstd and
 * the gradient dp w.r.t. the pooled output, returns d(conv kernel) HWIO, d(conv bias), dgamma, dbeta.  Equals
 * seld_k_bn_relu_pool_bwd followed by seld_k_conv3x3_wgrad without materialising dz. */
int seld_k_conv1_bwd_fused(const float* x, const float* z, const float* dp, const float* mean, const float* invstd,
                           const float* gamma, const float* beta, float* dw, float* db, float* dgamma, float* dbeta,
                           int B, int H, int Cin, int pt, int pf);
/* BatchNormalization(training) + ReLU + MaxPooling2D(pt,pf) given scale/shift per channel:
 * p = maxpool(relu(z*scale+shift)) (layers.py:33-35 + simple_conv_block) */
int seld_k_bn_relu_pool_fwd(const float* z, const float* scale, const float* shift, float* p,
                            int B, int H, int W, int C, int pt, int pf);
/* backward of the same: dp -> dz, dgamma, dbeta given batch mean / invstd / gamma / beta */
int seld_k_bn_relu_pool_bwd(const float* z, const float* dp, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, float* dz, float* dgamma, float* dbeta,
                            int B, int H, int W, int C, int pt, int pf);
/* C[M,N] = act(A[M,K] * op(B) + bias); transb=0: B [K,N]; 1: B [N,K]; act 0 none,1 sigmoid,2 tanh */
int seld_k_gemm(const float* A, const float* Bm, const float* bias, float* C, int M, int N, int K,
                int transb, int act, int accumulate);
/* two products sharing A in one launch: C0 = act(A*op(B0)+bias0), C1 = act(A*op(B1)+bias1) (the forward/backward GRU
 * input projections of modules.py:311-316, the first Conv1D of the two heads of modules.py:326-346) */
int seld_k_gemm_pair_n(const float* A, const float* B0, const float* B1, const float* bias0, const float* bias1,
                       float* C0, float* C1, int M, int N, int K, int transb, int act);
/* one product over a concatenated K axis: C = act(A0*op(B0) + A1*op(B1) + bias) (the gradient w.r.t. an input that
 * feeds two layers); K % 32 == 0 is required and anything else is refused */
int seld_k_gemm_pair_k(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias,
                       float* C, int M, int N, int K, int transb, int act, int accumulate);
/* the same products on the split-bf16 path (three exact bf16 terms per operand, 6 bf16 MFMAs per product, fp32
 * accumulation: fp32-level accuracy at 2.7x the fp32 MFMA rate): mode 0 C0 = act(A0 op(B0) + bias0); mode 1 also
 * C1 = act(A0 op(B1) + bias1); mode 2 C0 = act(A0 op(B0) + A1 op(B1) + bias0).  K % 32 == 0 and N % 128 == 0 are
 * required (anything else is refused: the caller uses seld_k_gemm).  This is what the train / predict entry points
 * run for the GRU input projections and the heads' first Conv1D. */
int seld_k_gemm_sb(const float* A0, const float* A1, const float* B0, const float* B1, const float* bias0,
                   const float* bias1, float* C0, float* C1, int M, int N, int K, int transb, int act, int mode);
/* C[K1,N] = A[M,K1]^T * B[M,N] (weight gradients of Dense / GRU kernels; K1 = 128 with N % 128 == 0 runs on the split-bf16
 * kernel with transposed LDS reads, gemm_tn_sb.hip, unless seld_k_set_option("gemm_tn_split_bf16", 0)); colsum (may be NULL): [N] = sum_m B[m,:]
 * (the matching bias gradient, produced by the same launch) */
int seld_k_gemm_tn(const float* A, const float* Bm, float* C, float* colsum, int M, int K1, int N);

/* xception_block's depthwise 3x3 backward on [B,H,16,64] NHWC (xception.hip; spec/XCEPTION_BLOCK.md: the unit is ReLU -> depthwise 3x3 ->
 * pointwise -> BatchNormalization).  dy: the gradient w.r.t. the depthwise OUTPUT; xin: the unit's input (aff == NULL) or, with aff = [scale 64 |
 * shift 64], the PREVIOUS unit's pre-BN tensor z (the activation is then relu(z scale + shift)); add (may be NULL): a residual gradient added to dx;
 * k [3][3][64] (Keras depthwise_kernel [3,3,64,1]).  Outputs: dx [B,H,16,64] = the gradient w.r.t. the (pre-ReLU) input, dk [3][3][64], and — with
 * bn_mean / bn_invstd [64] (aff required, add NULL) — sums [128] = [sum dx | sum dx xhat], xhat = (z - mean) invstd: the previous
 * BatchNormalization's backward sums.  fused = 1: one pass (dw3x3_w16_bwd_fused + two-stage combine + partial fold: the default of the model path,
 * round 5); fused = 0: the separate passes (dw3x3_bwd_data, dw3x3_bwd_w, xc_reduce). */
int seld_k_xc_dw_bwd(const float* dy, const float* k, const float* xin, const float* add, const float* aff, const float* bn_mean,
                     const float* bn_invstd, float* dx, float* dk, float* sums, int B, int H, int fused);
/* Bidirectional(GRU(128, reset_after=True), merge_mode='mul') recurrence (modules.py:311-316).
 * gx_* [B,S,384] = x*kernel + bias[0]; U_* [128,384]; brec_* = bias[1]; h_* [B,S,128]; out = h_f*h_b.
 * saved_* [B,S,128,4] (per unit: z, r, hh, h*U_h+b) may be NULL. */
int seld_k_gru_fwd(const float* gx_f, const float* gx_b, const float* U_f, const float* U_b,
                   const float* brec_f, const float* brec_b, float* h_f, float* h_b,
                   float* saved_f, float* saved_b, float* out, int B, int S, int units);
/* BPTT of the same: dout [B,S,128] -> dgx_* (input-side pre-activation grads), dgh_* (recurrent-side) */
int seld_k_gru_bwd(const float* dout, const float* h_f, const float* h_b, const float* saved_f,
                   const float* saved_b, const float* U_f, const float* U_b, float* dgx_f, float* dgx_b,
                   float* dgh_f, float* dgh_b, int B, int S, int units);
/* BinaryCrossentropy + MSE|MMSE on head outputs (train.py:26-29, losses.py:4-13): losses and the
 * gradients w.r.t. the PRE-activation head outputs (sigmoid / tanh folded in). */
int seld_k_losses(const float* sed, const float* doa, const float* y_sed, const float* y_doa,
                  const seld_loss_cfg* cfg, float* sloss, float* dloss, float* dsed_pre, float* ddoa_pre,
                  int B, int S, int nc);
/* Keras Adam update (train.py:311,34); step is 1-based */
int seld_k_adam(float* theta, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                float beta2, float eps, int64_t step);
/* Test aid (tests/test_model_gpu.py::test_parity_given_identical_routing): after seld_train_fwd_bwd, the routing decision the
 * backward pass took for every pooled element of conv block `block` — MaxPooling2D's argmax as window position row*pf + col and
 * ReLU's gate (pooled value > 0) — as two device byte arrays [B, H/pt, W/pf, 64].  MaxPoolGrad / ReluGrad of the reference
 * (layers.py:33-37 + simple_conv_block) take the same decisions from their own fp32 values; windows whose two largest elements
 * are within one rounding of each other may decide differently.  Synchronises the ctx stream. */
int seld_debug_pool_routing(seld_ctx* ctx, int block, unsigned char* pos, unsigned char* gate);
/* Test aid for resnet50_block (tests/test_model_gpu.py::test_resnet50_gru_train_step): after seld_train_fwd_bwd, the output of one
 * of bottleneck block `block`'s three ReLUs (which = 0: after the 1x1 reduce, 1: after the 3x3, 2: the block output), copied to the
 * device buffer `dst` (capacity floats); *count = its size, [B, T/5, W_block, w or 4w].  output > 0 is the gate the backward pass
 * used.  Synchronises the ctx stream. */
int seld_debug_relu_output(seld_ctx* ctx, int block, int which, float* dst, int64_t capacity, int64_t* count);
/* Test aids, the inverse of the two above (tests/test_model_gpu.py::test_full_size_parity_given_fp64_decisions): make every LATER backward
 * pass of this ctx take GIVEN decisions at a list of elements — host arrays of n flat indices into the decision tensor ([B, H/pt, W/pf,
 * 64] of conv block `block`; the ReLU output of seld_debug_relu_output(block, which)) and n values (routing: 0 = the window passes 0,
 * 1 + pos = it passes the element at window position pos; gate: 0 | 1).  n = 0 clears the list of that tensor.  MaxPoolGrad / ReluGrad of
 * the reference (layers.py:33-37) take these decisions from their own values; among the ~10^7..10^8 decisions of a full-size step a few
 * dozen (conv blocks) to a few thousand (the 16-bottleneck resnet) sit within fp32 rounding of a tie and decide differently in ANY two
 * evaluations.  With the fp64 oracle's decisions injected at exactly those elements (the fixtures carry them:
 * tests/golden/make_golden_*.py), the step's gradients are comparable with the free-running fp64 oracle's at the parity bar itself.
 * Implementation: the stored tensors the backward kernels read a decision from are edited by the smallest amount that makes them
 * decide as told (an ulp walk on one pre-BN value, the recorded argmax byte, a 1e-35 in place of a 0), after the forward pass has
 * finished with them.  Not for production use; calls synchronise the ctx stream. */
/* xception_block (spec/XCEPTION_BLOCK.md): seld_debug_pool_routing / _set_routing take block = 1 for the EXIT's MaxPool(ReLU(.)) over (1, 8)
 * ([B, T/5, 2, 64]); seld_debug_relu_output / _set_relu_gates take block = unit index 3 b + u, which = 0, for the ReLU in front of that
 * unit's SeparableConv2D ([B, T/5, 16, 64]; the read returns the value whose sign is the gate). */
int seld_debug_set_routing(seld_ctx* ctx, int block, int64_t n, const int64_t* idx_host, const unsigned char* val_host);
int seld_debug_set_relu_gates(seld_ctx* ctx, int block, int which, int64_t n, const int64_t* idx_host, const unsigned char* val_host);
/* ---- module operators (module_ops.hip): what seld_amd/modules.py composes the reference's configurable blocks from — modules.mother_block
 * / mother_stage (modules.py:15-43, 184-298: Conv2D(k, 'same', strides) + BatchNormalization + skip / projection / concatenation +
 * squeeze-and-excitation), and, around them, bidirectional_GRU_block and simple_dense_block at ANY feature width.  NHWC fp32 device tensors,
 * asynchronous on `stream`, no allocation (scratch is the caller's), any channel count / kernel / stride.  The dense products run on the
 * fp32 MFMA GEMM of gemm.hip; everything else is memory-bound elementwise / reduction work. */
int seld_m_conv_out(int in, int stride);      /* TensorFlow 'SAME': ceil(in / stride) */
/* col[(b,ho,wo)][(ki,kj,c)] of Conv2D(k, 'same', strides) on x [B,H,W,C]; the convolution is col * kernel[kh kw C, filters] + bias */
int seld_m_im2col(const float* x, float* col, int B, int H, int W, int C, int kh, int kw, int sh, int sw, void* stream);
int seld_m_col2im(const float* dcol, float* dx, int B, int H, int W, int C, int kh, int kw, int sh, int sw, int accumulate, void* stream);
int seld_m_gemm(const float* A, const float* Bm, const float* bias, float* Cm, int M, int N, int K, int transb, int accumulate, void* stream);
int64_t seld_m_gemm_tn_scratch(int K1, int N);
int seld_m_gemm_tn(const float* A, const float* Bm, float* Cm, float* colsum, float* slab, int M, int K1, int N, int seq, int shift, void* stream);
/* tf.keras.layers.BatchNormalization, training mode (layers.py:33, modules.py:232-268): batch mean / biased variance per channel ... */
int64_t seld_m_bn_scratch(int C);   /* floats of caller scratch seld_m_bn_stats / seld_m_bn_bwd take (two-stage sums that fill the card; NULL: one workgroup per channel) */
int seld_m_bn_stats(const float* z, int64_t npix, int C, float* mean, float* var, float* scratch, void* stream);
/* ... out (+)= (z - mean) rsqrt(var + eps) gamma + beta (inference: mean / var = the moving statistics) ... */
int seld_m_bn_apply(const float* z, const float* mean, const float* var, const float* gamma, const float* beta, float eps, float* out,
                    int64_t npix, int C, int accumulate, void* stream);
/* ... moving = moving * momentum + batch * (1 - momentum), the variance Bessel-corrected by count / (count - 1) ... */
int seld_m_bn_moving(const float* mean, const float* var, float* mov_mean, float* mov_var, int C, float momentum, int64_t count, void* stream);
/* ... and its gradient: dgamma, dbeta, dz */
int seld_m_bn_bwd(const float* z, const float* dy, const float* mean, const float* var, const float* gamma, float eps, float* dz, float* dgamma,
                  float* dbeta, int64_t npix, int C, float* scratch, void* stream);
#define SELD_ACT_SWISH 4
/* y = act(x); dx (+)= dy act'(x) from the pre-activation x.  kind: SELD_ACT_NONE / _SIGMOID / _TANH / _RELU / _SWISH */
int seld_m_act(const float* x, float* y, int64_t n, int kind, void* stream);
int seld_m_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int kind, int accumulate, void* stream);
int seld_m_axpy(float* dst, const float* src, int64_t n, float alpha, void* stream);      /* dst += alpha src */
/* tf.concat(axis=-1) piece: mode 0 dst[r][off + c] = src[r][c]; mode 1 (its gradient) src[r][c] += dst[r][off + c] */
int seld_m_copy_channels(float* src, float* dst, int64_t rows, int Cs, int Cd, int off, int mode, void* stream);
/* squeeze-and-excitation (modules.py:287-294): reduce_mean over (H, W); out = se * out; and their gradients */
int seld_m_mean_hw(const float* x, float* out, int B, int HW, int C, void* stream);
int seld_m_scale_hw(const float* x, const float* s, float* y, int B, int HW, int C, void* stream);
int seld_m_scale_hw_bwd_ds(const float* x, const float* dy, float* ds, int B, int HW, int C, void* stream);
int seld_m_scale_hw_bwd_dx(const float* dy, const float* s, const float* dmean, float* dx, int B, int HW, int C, int accumulate, void* stream);
/* The recurrent block (modules.py:302-319), the losses (train.py:26-34, losses.py:4-13) and Adam (train.py:311) as module operators: the same
 * kernels as seld_k_gru_fwd / _bwd / seld_k_losses / seld_k_adam, enqueued on the caller's stream with no host synchronisation (round 5: a
 * composed train step runs without one).  seld_m_losses takes its scratch from the caller: seld_m_losses_scratch(B * S) floats. */
int seld_m_gru_fwd(const float* gx_f, const float* gx_b, const float* U_f, const float* U_b, const float* brec_f, const float* brec_b, float* h_f,
                   float* h_b, float* saved_f, float* saved_b, float* out, int B, int S, int units, void* stream);
int seld_m_gru_bwd(const float* dout, const float* h_f, const float* h_b, const float* saved_f, const float* saved_b, const float* U_f,
                   const float* U_b, float* dgx_f, float* dgx_b, float* dgh_f, float* dgh_b, int B, int S, int units, void* stream);
int64_t seld_m_losses_scratch(int rows);
int seld_m_losses(const float* sed, const float* doa, const float* y_sed, const float* y_doa, const seld_loss_cfg* cfg, float* sloss, float* dloss,
                  float* dsed_pre, float* ddoa_pre, float* scratch, int B, int S, int nc, void* stream);
int seld_m_adam(float* theta, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, int64_t step, void* stream);
/* Measurement aid (bench.py, SURVEY.md §8(d) "state the step-latency floor"): the shader clock the card holds while `blocks`
 * workgroups of 512 threads run a VALU-only loop (the load shape of the GRU recurrence: 2B workgroups, no MFMA), from
 * s_memtime / s_memrealtime (100 MHz) inside the kernel.  No reference counterpart. */
int seld_k_valu_clock_mhz(int blocks, double* mhz);
/* resnet50_block pieces (spec/RESNET50_BLOCK.md; model_config/resnet50_gru.json:2-11), as the train / predict entry points compose them.
 *   seld_k_rn_conv      z [B,H,W/stride_f,Cout] = Conv2D(Cout, ksize in {1,3}, 'same', strides (1,stride_f), use_bias=False)(x [B,H,W,Cin]);
 *                       w HWIO.  ksize 1: a product on rows of stride Cin*stride_f; ksize 3: a product on im2col rows (formed on load by the
 *                       split-bf16 kernels when Cin and Cout are powers of two >= 128, else materialised); split-bf16 kernels wherever the
 *                       shape allows unless seld_k_set_option("rn_split_bf16", 0), else the fp32 MFMA GEMM.
 *   seld_k_rn_conv_bwd  dw (HWIO) and dx from dz
 *   seld_k_rn_bn        out = [relu](BatchNormalization(training)(z) [+ res]) over npix x C (C % 32 == 0); mean / invstd optional outputs
 *   seld_k_rn_bn_bwd    dz, dgamma, dbeta from dy gated by (mask > 0) (mask may be NULL) */
int seld_k_rn_conv(const float* x, const float* w, float* z, int B, int H, int W, int Cin, int Cout, int ksize, int stride_f);
int seld_k_rn_conv_bwd(const float* x, const float* w, const float* dz, float* dw, float* dx, int B, int H, int W, int Cin, int Cout,
                       int ksize, int stride_f);
int seld_k_rn_bn(const float* z, const float* gamma, const float* beta, const float* res, float* out, float* mean, float* invstd,
                 int64_t npix, int C, int relu);
int seld_k_rn_bn_bwd(const float* z, const float* dy, const float* mask, const float* gamma, float* dz, float* dgamma, float* dbeta,
                     int64_t npix, int C);
/* Diagnostic builds only (make CXXFLAGS+=-DGRU_TIMING; tools/tune_gru.py): shader-cycle sums per phase of the recurrence kernels'
 * last launch, wave 0 of every workgroup: cycles[blocks][4] = forward {h read + mat-vec, gate tail, barrier, chunk commit},
 * BPTT {gate gradients, barrier, coefficients + mat-vec + fold, -}.  SELD_ERR_UNSUPPORTED in the normal build. */
int seld_k_gru_timing(int which, unsigned long long* cycles, int blocks);
/* hipGetDeviceProperties of `device`: compute units, engine clock (kHz), memory clock (kHz), memory bus width (bits) —
 * bench.py prints the peaks they imply next to the constants its roofline fractions use. */
int seld_device_clocks(int device, int* compute_units, int* clock_khz, int* mem_clock_khz, int* mem_bus_bits);

#ifdef __cplusplus
}
#endif
#endif
