// loss_adam.hip — fused losses + optimizer of train.trainstep (train.py:22-36).
//   losses_kernel   BinaryCrossentropy (train.py:312-313) + tf.keras.losses.MSE / MAE / MSLE function form
//                   (train.py:317-320; [B,S] rows that tape.gradient SUMS) or
//                   losses.MMSE (losses.py:4-13): loss values and the gradients w.r.t. the
//                   PRE-activation head outputs (sigmoid / tanh derivatives folded in).
//   adam_kernel     Keras Adam, ResourceApplyAdam form, epsilon 1e-7 (train.py:311,34)
//   agc_kernel      utils.adaptive_clip_grad (utils.py:86-96), unit-wise norms (utils.py:70-83)
#include "common.h"

#define BCE_EPS 1e-7f

// deterministic single-block column reduction: out[c] = sum_r in[r*ncol + c], ncol <= 4
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ in, int rows, int ncol,
                                                           double* __restrict__ out) {
    __shared__ double red[1024];
    for (int c = 0; c < ncol; ++c) {
        double s = 0.0;
        for (int r = threadIdx.x; r < rows; r += 1024) s += (double)in[(size_t)r * ncol + c];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int w = 512; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[c] = red[0];
        __syncthreads();
    }
}

// MMSE mask of one row: m[c] = round(sum_k y[k*nc+c]^2)   (tf.round = half-to-even = rintf)
__global__ __launch_bounds__(256) void mmse_mask_rows_kernel(const float* __restrict__ y_doa, float* __restrict__ rowsum,
                                                             int rows, int nc) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* y = y_doa + (size_t)r * 3 * nc;
    float s = 0.f;
    for (int c = 0; c < nc; ++c) {
        const float a = y[c], b = y[nc + c], d = y[2 * nc + c];
        s += 3.f * rintf(a * a + b * b + d * d);
    }
    rowsum[r] = s;
}

__global__ void den_store_kernel(const double* __restrict__ sums, float* __restrict__ den) { den[0] = (float)sums[0]; }

int loss_scratch_floats(int rows) { return rows * 2 + 64; }

// scratch layout: [rows*2 floats row partials][16 doubles reduced sums]
int launch_mmse_den(hipStream_t st, const float* y_doa, float* den, float* scratch, int rows, int nc) {
    double* sums = reinterpret_cast<double*>(scratch + (size_t)rows * 2);
    hipLaunchKernelGGL(mmse_mask_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, y_doa, scratch, rows, nc);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(1024), 0, st, scratch, rows, 1, sums);
    hipLaunchKernelGGL(den_store_kernel, dim3(1), dim3(1), 0, st, sums, den);
    return 0;
}

// sum over the 4 lanes of a quad (DPP quad_perm moves; every lane gets the sum)
__device__ __forceinline__ float loss_quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, true));
    return v;
}

// Four threads per row (q = tid & 3 takes the columns q, q + 4, ...: a quad reads 16 contiguous bytes), 64 rows per block.
// Per block the two loss sums are reduced in a fixed order (quad -> row order in LDS, double) into blockpart[block][2];
// losses_finalize adds those in block order: two runs give the same bits.
__global__ __launch_bounds__(256) void losses_kernel(const float* __restrict__ sed, const float* __restrict__ doa,
                                                     const float* __restrict__ y_sed, const float* __restrict__ y_doa,
                                                     int doa_loss, float coef_sed, float w_doa,
                                                     const float* __restrict__ den_dev, float* __restrict__ dloss_rows,
                                                     float* __restrict__ dsed_pre, float* __restrict__ ddoa_pre,
                                                     double* __restrict__ blockpart, int rows, int nc, int ld_sed, int ld_doa) {
    __shared__ float rs[64][2];
    const int q = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int r = min(blockIdx.x * 64 + rl, rows - 1);       // rows past the end recompute the last row and store nothing
    const bool live = blockIdx.x * 64 + rl < rows;
    const float* p = sed + (size_t)r * nc;
    const float* ys = y_sed + (size_t)r * nc;
    float bsum = 0.f;
    for (int c = q; c < nc; c += 4) {
        const float pv = p[c], y = ys[c];
        const float pc = fminf(fmaxf(pv, BCE_EPS), 1.f - BCE_EPS);
        bsum += -(y * logf(pc + BCE_EPS) + (1.f - y) * logf(1.f - pc + BCE_EPS));
        if (dsed_pre && live) {
            const bool pass = (pv >= BCE_EPS) && (pv <= 1.f - BCE_EPS);
            const float dldp = pass ? -(y / (pc + BCE_EPS) - (1.f - y) / (1.f - pc + BCE_EPS)) : 0.f;
            dsed_pre[(size_t)r * ld_sed + c] = coef_sed * dldp * pv * (1.f - pv);
        }
    }
    const float* d = doa + (size_t)r * 3 * nc;
    const float* yd = y_doa + (size_t)r * 3 * nc;
    float dsum = 0.f;
    if (doa_loss != 1) {
        // the Keras loss FUNCTIONS (mean over the last axis -> one value per row): MSE, MAE (mean |y - p|), MSLE (mean of the squared
        // difference of log(max(., 1e-7) + 1); keras/losses.py mean_squared_logarithmic_error).  Gradients as tape.gradient forms
        // them: d|e| = sign(e) (0 at e = 0), maximum() passes the gradient to the larger argument (half each on a tie)
        const float inv = 1.f / (float)(3 * nc);
        for (int k = q; k < 3 * nc; k += 4) {
            const float pk = d[k], yk = yd[k];
            float term, dldp;
            if (doa_loss == 0) {
                const float e = yk - pk;
                term = e * e;
                dldp = -2.f * e;
            } else if (doa_loss == 2) {
                const float e = pk - yk;
                term = fabsf(e);
                dldp = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
            } else {
                const float lp = logf(fmaxf(pk, BCE_EPS) + 1.f), ly = logf(fmaxf(yk, BCE_EPS) + 1.f);
                const float e = lp - ly;
                term = e * e;
                const float pass = pk > BCE_EPS ? 1.f : (pk == BCE_EPS ? 0.5f : 0.f);
                dldp = 2.f * e * pass / (fmaxf(pk, BCE_EPS) + 1.f);
            }
            dsum += term;
            if (ddoa_pre && live) ddoa_pre[(size_t)r * ld_doa + k] = w_doa * (dldp * inv) * (1.f - pk * pk);
        }
        dsum = loss_quad_sum(dsum) * inv;
        if (dloss_rows && live && q == 0) dloss_rows[r] = dsum;
    } else {
        const float den = den_dev[0];
        for (int c = q; c < nc; c += 4) {
            const float a = yd[c], b = yd[nc + c], e3 = yd[2 * nc + c];
            const float m = rintf(a * a + b * b + e3 * e3);
            for (int k = 0; k < 3; ++k) {
                const int i = k * nc + c;
                const float e = yd[i] - d[i];
                dsum += e * e * m;
                if (ddoa_pre && live) ddoa_pre[(size_t)r * ld_doa + i] = w_doa * (-2.f * e * m / den) * (1.f - d[i] * d[i]);
            }
        }
        dsum = loss_quad_sum(dsum);
    }
    bsum = loss_quad_sum(bsum);
    if (q == 0) { rs[rl][0] = live ? bsum : 0.f; rs[rl][1] = live ? dsum : 0.f; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double s = 0.0;
        for (int i = 0; i < 64; ++i) s += (double)rs[i][threadIdx.x];
        blockpart[(size_t)blockIdx.x * 2 + threadIdx.x] = s;
    }
}

// one wave: lane l adds the partials l, l + 64, ... in order, then a fixed xor tree over the lanes (same bits every run)
__global__ __launch_bounds__(64) void losses_finalize_kernel(const double* __restrict__ blockpart, int nblocks, int doa_loss, double n_sed,
                                                             const float* __restrict__ den_dev, float* __restrict__ sloss,
                                                             float* __restrict__ dloss) {
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 64) { s0 += blockpart[2 * i]; s1 += blockpart[2 * i + 1]; }
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) { s0 += __shfl_xor(s0, w); s1 += __shfl_xor(s1, w); }
    if (threadIdx.x) return;
    if (sloss) sloss[0] = (float)(s0 / n_sed);
    if (doa_loss == 1 && dloss) dloss[0] = (float)(s1 / (double)den_dev[0]);
}

int launch_losses_finalize(hipStream_t st, int doa_loss, const float* den_dev, float* sloss, float* dloss, float* scratch, int B, int S,
                           int nc);
int launch_losses(hipStream_t st, const float* sed, const float* doa, const float* y_sed, const float* y_doa,
                  int doa_loss, float w_sed, float w_doa, float sed_grad_scale, const float* den_dev,
                  float* sloss, float* dloss, float* dsed_pre, float* ddoa_pre, float* scratch, int B, int S, int nc, int ld_sed,
                  int ld_doa, int defer_finalize) {
    if (ld_sed <= 0) ld_sed = nc;          // row strides of the two gradient outputs (a shared [rows][4 nc] buffer: 4 nc each)
    if (ld_doa <= 0) ld_doa = 3 * nc;
    const int rows = B * S;
    // BCE is a mean over rows*nc elements.  With the Keras MSE *function* the per-row loss tensor
    // sloss*w0 + mse[b,s]*w1 is summed by tape.gradient, which multiplies the BCE term by rows.
    const float coef_sed = w_sed * sed_grad_scale * (doa_loss != 1 ? (float)rows : 1.f) / ((float)rows * (float)nc);
    double* blockpart = reinterpret_cast<double*>(scratch);      // [nblocks][2]: rows / 32 doubles of the 2 * rows floats
    const int nblocks = (rows + 63) / 64;
    hipLaunchKernelGGL(losses_kernel, dim3(nblocks), dim3(256), 0, st, sed, doa, y_sed, y_doa, doa_loss, coef_sed, w_doa, den_dev,
                       doa_loss != 1 ? dloss : nullptr, dsed_pre, ddoa_pre, blockpart, rows, nc, ld_sed, ld_doa);
    if (!defer_finalize) launch_losses_finalize(st, doa_loss, den_dev, sloss, dloss, scratch, B, S, nc);
    return 0;
}

// the scalar loss values from the per-block partials launch_losses left in `scratch` (which must be untouched in between): a
// training step runs this after the backward pass — nothing in the backward waits for the scalars
int launch_losses_finalize(hipStream_t st, int doa_loss, const float* den_dev, float* sloss, float* dloss, float* scratch, int B, int S,
                           int nc) {
    const int rows = B * S, nblocks = (rows + 63) / 64;
    hipLaunchKernelGGL(losses_finalize_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<const double*>(scratch), nblocks, doa_loss,
                       (double)rows * nc, den_dev, sloss, dloss);
    return 0;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ theta, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_t,
                                                   float b1, float b2, float eps) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    theta[i] = theta[i] - lr_t * mi / (sqrtf(vi) + eps);
}

int launch_adam(hipStream_t st, float* theta, const float* g, float* m, float* v, int64_t n, float lr_t,
                float beta1, float beta2, float eps) {
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, theta, g, m, v, n, lr_t, beta1,
                       beta2, eps);
    return 0;
}

// unit-wise clip: one thread per "unit" (column); a unit's elements are r*cols + c, r < rows
__global__ __launch_bounds__(256) void agc_kernel(const float* __restrict__ theta, float* __restrict__ g, int rows, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    float pn = 0.f, gn = 0.f;
    for (int r = 0; r < rows; ++r) {
        const float p = theta[(size_t)r * cols + c], gg = g[(size_t)r * cols + c];
        pn += p * p;
        gn += gg * gg;
    }
    pn = sqrtf(pn);
    gn = sqrtf(gn);
    const float max_norm = fmaxf(pn, 1e-3f) * 0.01f;
    if (!(gn < max_norm)) {
        const float sc = max_norm / fmaxf(gn, 1e-6f);
        for (int r = 0; r < rows; ++r) g[(size_t)r * cols + c] *= sc;
    }
}

int launch_agc(hipStream_t st, const float* theta, float* g, int64_t off, int rank, const int64_t* shape, float* scratch) {
    (void)scratch;
    int rows = 1, cols = 1;
    if (rank <= 1) { rows = (int)shape[0]; cols = 1; }
    else if (rank == 2) { rows = (int)shape[0]; cols = (int)shape[1]; }
    else if (rank == 3) { rows = (int)shape[0]; cols = (int)(shape[1] * shape[2]); }
    else { rows = (int)(shape[0] * shape[1] * shape[2]); cols = (int)shape[3]; }
    hipLaunchKernelGGL(agc_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, theta + off, g + off, rows, cols);
    return 0;
}

// dy <- dy * act'(.) from the layer's stored OUTPUT y = act(.): simple_dense_block's dense_activation on the heads' hidden layers
// (modules.py:356, 368-371).  act: 1 sigmoid y (1 - y), 2 tanh 1 - y^2, 3 relu [y > 0].  n % 4 == 0 (dense units are multiples of 4).
__global__ __launch_bounds__(256) void act_bwd_kernel(const float4* __restrict__ y, float4* __restrict__ dy, int64_t n4, int act) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = y[i];
    float4 d = dy[i];
    if (act == 1) { d.x *= v.x * (1.f - v.x); d.y *= v.y * (1.f - v.y); d.z *= v.z * (1.f - v.z); d.w *= v.w * (1.f - v.w); }
    else if (act == 2) { d.x *= 1.f - v.x * v.x; d.y *= 1.f - v.y * v.y; d.z *= 1.f - v.z * v.z; d.w *= 1.f - v.w * v.w; }
    else if (act == 3) { d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f; d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f; }
    dy[i] = d;
}
int launch_act_bwd(hipStream_t st, const float* y, float* dy, int64_t n, int act) {
    if (n & 3) return -1;
    const int64_t n4 = n >> 2;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(y),
                       reinterpret_cast<float4*>(dy), n4, act);
    return 0;
}

// ---- models.seldnet_v1 (models.py:36-52): doa_out = tanh(doa * Concatenate([sed] * 3)) with sed = sigmoid(.), doa = tanh(.) of the two heads.
// Forward: one thread per (row, class): the three DOA components of the class share its sed value; writes the library's copy (what the
// losses read) and the caller's.
__global__ __launch_bounds__(256) void v1_couple_fwd_kernel(const float* __restrict__ sed, const float* __restrict__ doa1, float* __restrict__ out,
                                                            float* __restrict__ out2, int64_t n, int nc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over [rows][nc]
    if (i >= n) return;
    const int64_t r = i / nc;
    const int j = (int)(i - r * nc);
    const float s = sed[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int64_t o = r * 3 * nc + k * nc + j;
        const float v = tanhf(doa1[o] * s);
        out[o] = v;
        if (out2) out2[o] = v;
    }
}
int launch_v1_couple_fwd(hipStream_t st, const float* sed, const float* doa1, float* out, float* out2, int rows, int nc) {
    const int64_t n = (int64_t)rows * nc;
    hipLaunchKernelGGL(v1_couple_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sed, doa1, out, out2, n, nc);
    return 0;
}
// Backward: losses_kernel left du = dL/d(doa sed) (its tanh' factor used the coupled output) in the DOA slot and the BCE term's gradient
// w.r.t. the SED pre-activation in the SED slot.  d doa_pre = du * sed * (1 - doa^2); d sed_pre += (sum_k du_k doa_k) * sed (1 - sed).
__global__ __launch_bounds__(256) void v1_couple_bwd_kernel(const float* __restrict__ sed, const float* __restrict__ doa1, float* __restrict__ dsed_pre,
                                                            int ld_sed, float* __restrict__ ddoa_pre, int ld_doa, int64_t n, int nc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t r = i / nc;
    const int j = (int)(i - r * nc);
    const float s = sed[i];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d1 = doa1[r * 3 * nc + k * nc + j];
        float* g = ddoa_pre + r * ld_doa + k * nc + j;
        const float du = *g;
        acc += du * d1;
        *g = du * s * (1.f - d1 * d1);
    }
    dsed_pre[r * ld_sed + j] += acc * s * (1.f - s);
}
int launch_v1_couple_bwd(hipStream_t st, const float* sed, const float* doa1, float* dsed_pre, int ld_sed, float* ddoa_pre, int ld_doa, int rows, int nc) {
    const int64_t n = (int64_t)rows * nc;
    hipLaunchKernelGGL(v1_couple_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sed, doa1, dsed_pre, ld_sed, ddoa_pre, ld_doa, n, nc);
    return 0;
}

// ---- simple_dense_block with kernel_size > 1 (modules.py:355, 370-372: Conv1D(units, kernel_size, padding='same') over the frames of a clip)
// The layer runs as the dense product it is once its input rows are laid side by side: xe[b, t, j * C + c] = x[b, t + j - (ks - 1) / 2, c]
// (zero outside the clip; TensorFlow's 'same' puts the extra pad of an even kernel at the end), K = ks * C, and the Keras kernel
// [ks, C, units] is that product's [K, units] matrix as stored.  C % 4 == 0.
__global__ __launch_bounds__(256) void time_expand_kernel(const float4* __restrict__ x, float4* __restrict__ xe, int S, int C4, int ks, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over [B * S][ks][C4]
    if (i >= n4) return;
    const int c = (int)(i % C4), j = (int)((i / C4) % ks);
    const int64_t row = i / ((int64_t)C4 * ks);
    const int t = (int)(row % S), ts = t + j - (ks - 1) / 2;
    xe[i] = (ts >= 0 && ts < S) ? x[(row + (ts - t)) * C4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
}
int launch_time_expand(hipStream_t st, const float* x, float* xe, int B, int S, int C, int ks) {
    if ((C & 3) || ks < 1) return -1;
    const int64_t n4 = (int64_t)B * S * ks * (C / 4);
    hipLaunchKernelGGL(time_expand_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(xe), S, C / 4, ks, n4);
    return 0;
}
// its transpose: dx[b, t, c] (+)= sum_j dxe[b, t - j + (ks - 1) / 2, j * C + c], taps in order (fixed summation order)
__global__ __launch_bounds__(256) void time_fold_kernel(const float4* __restrict__ dxe, float4* __restrict__ dx, int S, int C4, int ks, int64_t n4, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over [B * S][C4]
    if (i >= n4) return;
    const int c = (int)(i % C4);
    const int64_t row = i / C4;
    const int t = (int)(row % S);
    float4 a = accumulate ? dx[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < ks; ++j) {
        const int tr = t - j + (ks - 1) / 2;                        // the output row whose tap j read this input row
        if (tr < 0 || tr >= S) continue;
        const float4 v = dxe[((row + (tr - t)) * ks + j) * C4 + c];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    dx[i] = a;
}
int launch_time_fold(hipStream_t st, const float* dxe, float* dx, int B, int S, int C, int ks, int accumulate) {
    if ((C & 3) || ks < 1) return -1;
    const int64_t n4 = (int64_t)B * S * (C / 4);
    hipLaunchKernelGGL(time_fold_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(dxe),
                       reinterpret_cast<float4*>(dx), S, C / 4, ks, n4, accumulate);
    return 0;
}

// ---- Dropout (modules.py:373-374; Keras: kept where uniform >= rate, scaled by 1 / (1 - rate), training only).  The reference draws from
// TensorFlow's generator, which no other program reproduces; here the uniforms are Philox4x32-10 words of the counter
// (element / 4, layer, step, 0) under the key (seed lo, seed hi), u = (word >> 8) * 2^-24 — a pure function of (seed, step, layer, element)
// that the oracle restates (oracle/seldnet_oracle.py::philox_uniform), so the backward pass recomputes the mask instead of storing it.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}
// out = in * mask / (1 - rate)  (forward: in = the layer's activations; backward: in = out = the gradient, in place)
__global__ __launch_bounds__(256) void dropout_kernel(const float4* __restrict__ in, float4* __restrict__ out, int64_t n4, float rate, float scale,
                                                      unsigned seed_lo, unsigned seed_hi, unsigned layer, unsigned step) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const uint4 r = philox4x32_10(make_uint4((unsigned)i, layer, step, (unsigned)(i >> 32)), make_uint2(seed_lo, seed_hi));
    const float4 v = in[i];
    const float k = 1.f / 16777216.f;
    out[i] = make_float4((float)(r.x >> 8) * k >= rate ? v.x * scale : 0.f, (float)(r.y >> 8) * k >= rate ? v.y * scale : 0.f,
                         (float)(r.z >> 8) * k >= rate ? v.z * scale : 0.f, (float)(r.w >> 8) * k >= rate ? v.w * scale : 0.f);
}
int launch_dropout(hipStream_t st, const float* in, float* out, int64_t n, float rate, uint64_t seed, unsigned layer, unsigned step) {
    if ((n & 3) || !(rate >= 0.f && rate < 1.f)) return -1;
    const int64_t n4 = n >> 2;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(in),
                       reinterpret_cast<float4*>(out), n4, rate, 1.f / (1.f - rate), (unsigned)seed, (unsigned)(seed >> 32), layer, step);
    return 0;
}

// ---- Keras GRU dropout masks (modules.py:312-314): one mask row per clip, the same for every step of the sequence
__global__ __launch_bounds__(256) void mask_rows_kernel(const float4* __restrict__ in, const float4* __restrict__ mask, float4* __restrict__ out, int64_t n4, int S,
                                                        int F4, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int64_t r = i / F4;
    const int f = (int)(i - r * F4);
    const float4 v = in[i], m = mask[(r / S) * F4 + f];
    float4 o = make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w);
    if (accumulate) { const float4 p = out[i]; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
    out[i] = o;
}
int launch_mask_rows(hipStream_t st, const float* in, const float* mask, float* out, int64_t rows, int S, int F, int accumulate) {
    if ((F & 3) || rows <= 0 || S <= 0) return -1;
    const int64_t n4 = rows * (F / 4);
    hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(in),
                       reinterpret_cast<const float4*>(mask), reinterpret_cast<float4*>(out), n4, S, F / 4, accumulate);
    return 0;
}
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ out, int64_t n, float v) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v;
}
int launch_fill(hipStream_t st, float* out, int64_t n, float v) {
    if (n <= 0) return -1;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, v);
    return 0;
}
