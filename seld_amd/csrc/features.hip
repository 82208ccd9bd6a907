// features.hip — on-device feature stage of feature_extractor.extract_features (feature_extractor.py:53-88):
// 4-channel STFT (torchaudio spectrogram = torch.stft, center/reflect, periodic hann zero-padded to n_fft,
// :153-173) -> |X|^2 -> HTK mel filterbank -> 10*log10 with top_db 80 (:59-71), plus FOA intensity vectors
// through the same filterbank without dB (:176-193, :75-77) or GCC-PHAT (:196-214), written as [T, n_mels, C].
//
// One workgroup per frame.  The four real channels are packed into TWO complex FFTs (ch0 + i*ch1,
// ch2 + i*ch3), transformed by a radix-2 Stockham autosort FFT in LDS (ping-pong buffers, twiddles in
// LDS), separated by conjugate symmetry, reduced to power / intensity-vector / phase spectra in LDS and
// projected through a SPARSE mel filterbank (each triangular filter is a contiguous bin range).
// HBM traffic is the algorithmic minimum: every sample is read ~n_fft/hop times from L2, every output
// element written once (+ one in-place pass for the top_db clamp, whose max needs the whole clip).
#include "common.h"
#include "../../include/seld_hip.h"

#include <math.h>
#include <string>
#include <vector>

struct seld_feat {
    int sample_rate, n_fft, win_length, hop, n_mels, mode, device, logn;
    int n_bins;                 // n_fft/2 + 1
    float* win = nullptr;       // [n_fft] hann, zero padded, optionally normalised
    float2* tw = nullptr;       // [n_fft/2] exp(-2 pi i t / n_fft)
    int* mel_start = nullptr;   // [n_mels] first bin of the filter
    int* mel_count = nullptr;   // [n_mels]
    int* mel_off = nullptr;     // [n_mels] offset into mel_w
    float* mel_w = nullptr;     // packed weights
    float* gmax = nullptr;      // 1 float: max dB of the clip (ordered-int atomics)
    std::string err;
};

namespace {

std::string g_feat_err;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// MODE 0 = foa (4 mel-dB + 3 mel-IV), 1 = mic (4 mel-dB + 6 GCC)
template <int MODE>
__global__ __launch_bounds__(256) void feat_frame_kernel(const float* __restrict__ wav, int64_t n_samples, int n_fft, int logn,
                                                         int hop, int n_mels, const float* __restrict__ win,
                                                         const float2* __restrict__ tw_g, const int* __restrict__ mel_start,
                                                         const int* __restrict__ mel_count, const int* __restrict__ mel_off,
                                                         const float* __restrict__ mel_w, float* __restrict__ out,
                                                         float* __restrict__ gmax) {
    constexpr int C_OUT = MODE == 0 ? 7 : 10;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = n_fft, NB = N / 2 + 1, NBP = NB + 3;   // padded row of per-bin values
    float2* bufA = reinterpret_cast<float2*>(smem);       // [2][N]
    float2* bufB = bufA + 2 * N;                          // [2][N]
    float2* tw = bufB + 2 * N;                            // [N/2]
    float* val = reinterpret_cast<float*>(tw + N / 2);    // foa: [7][NBP]; mic: [4][NBP] + phase [6][NBP] float2
    float* red = val + (MODE == 0 ? 7 * NBP : 4 * NBP + 12 * NBP);
    const int tid = threadIdx.x;
    const int64_t t = blockIdx.x;
    for (int i = tid; i < N / 2; i += 256) tw[i] = tw_g[i];
    // ---- windowed, reflect-padded frame (torch.stft center=True): sample index t*hop + n - N/2
    const float* x0 = wav;
    const float* x1 = wav + n_samples;
    const float* x2 = wav + 2 * n_samples;
    const float* x3 = wav + 3 * n_samples;
    for (int n = tid; n < N; n += 256) {
        int64_t i = t * hop + n - N / 2;
        if (i < 0) i = -i;
        if (i >= n_samples) i = 2 * (n_samples - 1) - i;
        const float w = win[n];
        bufA[n] = make_float2(w * x0[i], w * x1[i]);
        bufA[N + n] = make_float2(w * x2[i], w * x3[i]);
    }
    __syncthreads();
    // ---- radix-2 Stockham autosort FFT, both packed transforms at once (oracle: features_oracle.stockham_fft)
    float2* src = bufA;
    float2* dst = bufB;
    for (int s = 0; s < logn; ++s) {
        const int m = 1 << s;
        const int tstride = N >> (s + 1);            // N / (2m)
        for (int bi = tid; bi < N; bi += 256) {      // N/2 butterflies x 2 transforms
            const int f = bi >> (logn - 1), j = bi & (N / 2 - 1);
            const int k = j & (m - 1);
            const float2 u = src[f * N + j];
            const float2 v = cmul(tw[k * tstride], src[f * N + j + N / 2]);
            dst[f * N + 2 * j - k] = make_float2(u.x + v.x, u.y + v.y);
            dst[f * N + 2 * j - k + m] = make_float2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        float2* tmp = src; src = dst; dst = tmp;
    }
    // ---- separate the packed real channels; per-bin power / IV / phase spectra
    float2* ph = reinterpret_cast<float2*>(val + 4 * NBP);   // mic only
    for (int k = tid; k < NB; k += 256) {
        const int kn = (N - k) & (N - 1);
        float2 X[4];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const float2 a = src[f * N + k], b = src[f * N + kn];
            // Xa = (Z[k] + conj(Z[N-k]))/2 ; Xb = (Z[k] - conj(Z[N-k]))/(2i)
            X[2 * f] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
            X[2 * f + 1] = make_float2(0.5f * (a.y + b.y), 0.5f * (b.x - a.x));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) val[c * NBP + k] = X[c].x * X[c].x + X[c].y * X[c].y;
        if (MODE == 0) {
            // Re(conj(W) * X_i): IVx <- ch3, IVy <- ch1, IVz <- ch2
            float ivx = X[0].x * X[3].x + X[0].y * X[3].y;
            float ivy = X[0].x * X[1].x + X[0].y * X[1].y;
            float ivz = X[0].x * X[2].x + X[0].y * X[2].y;
            const float nrm = fmaxf(sqrtf(ivx * ivx + ivy * ivy + ivz * ivz), 1e-8f);
            val[4 * NBP + k] = ivx / nrm;
            val[5 * NBP + k] = ivy / nrm;
            val[6 * NBP + k] = ivz / nrm;
        } else {
            int pidx = 0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = a + 1; b < 4; ++b) {
                    // R = conj(Xa) * Xb ; exp(i*angle(R)) = R/|R|, angle(0) = 0 -> 1
                    const float rr = X[a].x * X[b].x + X[a].y * X[b].y;
                    const float ri = X[a].x * X[b].y - X[a].y * X[b].x;
                    const float mag = sqrtf(rr * rr + ri * ri);
                    ph[pidx * NBP + k] = mag > 0.f ? make_float2(rr / mag, ri / mag) : make_float2(1.f, 0.f);
                    ++pidx;
                }
        }
    }
    __syncthreads();
    // ---- sparse mel projection; thread (m = tid % n_mels.., g) handles channels g, g+4
    float lmax = -INFINITY;
    const int nm_ch = MODE == 0 ? 7 : 4;
    for (int idx = tid; idx < n_mels * nm_ch; idx += 256) {
        const int c = idx / n_mels, m = idx - c * n_mels;
        const int st = mel_start[m], cnt = mel_count[m];
        const float* wv = mel_w + mel_off[m];
        const float* vv = val + c * NBP + st;
        float acc = 0.f;
        for (int i = 0; i < cnt; ++i) acc = fmaf(wv[i], vv[i], acc);
        float o = acc;
        if (c < 4) {
            o = 10.f * log10f(fmaxf(acc, 1e-10f));
            lmax = fmaxf(lmax, o);
        }
        out[((size_t)t * n_mels + m) * C_OUT + c] = o;
    }
    if (MODE == 1) {
        // GCC-PHAT: cc[lag] = irfft(phase)[lag], lag = j - n_mels/2, j in [0, n_mels)
        const float invn = 1.f / (float)N;
        for (int idx = tid; idx < 6 * n_mels; idx += 256) {
            const int p = idx / n_mels, j = idx - p * n_mels;
            const int lag = j - n_mels / 2;
            const float2* pp = ph + p * NBP;
            float acc = 0.f;
            for (int k = 1; k < N / 2; ++k) {
                const int ti = (int)(((long long)k * lag) & (N - 1));   // (k*lag) mod N, N power of two
                // e^{+2 pi i k lag / N} = conj(tw_full[ti]); tw_full[ti] = ti < N/2 ? tw[ti] : -tw[ti - N/2]
                float2 w = tw[ti & (N / 2 - 1)];
                if (ti >= N / 2) { w.x = -w.x; w.y = -w.y; }
                acc += pp[k].x * w.x + pp[k].y * w.y;   // Re(P * conj(w)) with w = e^{-i..}: Pr*cos - Pi*sin(+) -> Pr*w.x + Pi*w.y
            }
            const float nyq = (lag & 1) ? -pp[N / 2].x : pp[N / 2].x;
            out[((size_t)t * n_mels + j) * C_OUT + 4 + p] = (pp[0].x + nyq + 2.f * acc) * invn;
        }
    }
    // ---- clip-wide max of the dB channels (top_db clamp needs it)
    red[tid] = lmax;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] = fmaxf(red[tid], red[tid + w]);
        __syncthreads();
    }
    if (tid == 0) atomic_max_float(gmax, red[0]);
}

__global__ __launch_bounds__(256) void feat_topdb_kernel(float* __restrict__ out, const float* __restrict__ gmax, int64_t n_tm,
                                                         int c_out, float top_db) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_tm * 4) return;
    const int64_t tm = i >> 2;
    const int c = (int)(i & 3);
    const float floor_db = gmax[0] - top_db;
    float* p = out + tm * c_out + c;
    *p = fmaxf(*p, floor_db);
}

__global__ void feat_init_max_kernel(float* gmax) { gmax[0] = -INFINITY; }

// features[T_in, FC] -> (x - mean)/max(std, eps), trimmed / zero-padded to T_out rows
// (preprocess_features_labels :117-149 followed by apply_normalizer :226-234; the reference pads BEFORE
// normalising, so padded rows become (0 - mean)/std, reproduced here)
__global__ __launch_bounds__(256) void feat_normalize_kernel(const float* __restrict__ f, const float* __restrict__ mean,
                                                             const float* __restrict__ stdv, float* __restrict__ out,
                                                             int64_t T_in, int64_t T_out, int FC, float eps) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= T_out * FC) return;
    const int64_t t = i / FC;
    const int fc = (int)(i - t * FC);
    const float v = t < T_in ? f[i] : 0.f;
    out[i] = (v - mean[fc]) / fmaxf(stdv[fc], eps);
}

int ffail(seld_feat* f, int code, const std::string& msg) {
    if (f) f->err = msg; else g_feat_err = msg;
    return code;
}

}  // namespace

extern "C" {

const char* seld_feat_last_error(const seld_feat* f) { return f ? f->err.c_str() : g_feat_err.c_str(); }

int seld_feat_create(int sample_rate, int n_fft, int win_length, int hop_length, int n_mels, int mode, int normalized,
                     int device, seld_feat** out) {
    if (!out) return ffail(nullptr, SELD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (n_fft < 64 || n_fft > 4096 || (n_fft & (n_fft - 1))) return ffail(nullptr, SELD_ERR_UNSUPPORTED, "n_fft must be a power of two in [64, 4096]");
    if (win_length <= 0) win_length = n_fft;
    if (hop_length <= 0) hop_length = win_length / 2;
    if (win_length > n_fft) return ffail(nullptr, SELD_ERR_INVALID, "win_length > n_fft");
    if (n_mels <= 0 || n_mels > 256 || (n_mels & 1)) return ffail(nullptr, SELD_ERR_INVALID, "n_mels must be even and in (0, 256]");
    if (mode != 0 && mode != 1) return ffail(nullptr, SELD_ERR_INVALID, "mode must be 0 (foa) or 1 (mic)");
    if (sample_rate <= 0) return ffail(nullptr, SELD_ERR_INVALID, "sample_rate");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return ffail(nullptr, SELD_ERR_HIP, "no such HIP device");
    hipSetDevice(device);
    seld_feat* f = new seld_feat();
    f->sample_rate = sample_rate; f->n_fft = n_fft; f->win_length = win_length; f->hop = hop_length; f->n_mels = n_mels;
    f->mode = mode; f->device = device; f->n_bins = n_fft / 2 + 1;
    f->logn = 0;
    while ((1 << f->logn) < n_fft) ++f->logn;
    const double PI = 3.14159265358979323846;
    // periodic hann of win_length, centred in n_fft (torch.stft pads the window on both sides)
    std::vector<float> win(n_fft, 0.f);
    const int left = (n_fft - win_length) / 2;
    double wsq = 0.0;
    for (int n = 0; n < win_length; ++n) {
        const double w = 0.5 - 0.5 * cos(2.0 * PI * n / win_length);
        win[left + n] = (float)w;
        wsq += w * w;
    }
    if (normalized) for (auto& w : win) w = (float)(w / sqrt(wsq));
    std::vector<float2> tw(n_fft / 2);
    for (int t = 0; t < n_fft / 2; ++t) tw[t] = make_float2((float)cos(2.0 * PI * t / n_fft), (float)-sin(2.0 * PI * t / n_fft));
    // torchaudio create_fb_matrix (HTK, f_min 0, f_max sr//2, norm None), computed in double, stored sparse
    const int nb = f->n_bins;
    const double f_max = (double)(sample_rate / 2);
    const double m_max = 2595.0 * log10(1.0 + f_max / 700.0);
    std::vector<double> f_pts(n_mels + 2);
    for (int i = 0; i < n_mels + 2; ++i) f_pts[i] = 700.0 * (pow(10.0, (m_max * i / (n_mels + 1)) / 2595.0) - 1.0);
    std::vector<int> mstart(n_mels), mcount(n_mels), moff(n_mels);
    std::vector<float> mw;
    for (int m = 0; m < n_mels; ++m) {
        int first = -1, last = -1;
        std::vector<float> wts(nb, 0.f);
        for (int b = 0; b < nb; ++b) {
            const double fr = (nb > 1) ? f_max * b / (nb - 1) : 0.0;
            const double down = (fr - f_pts[m]) / (f_pts[m + 1] - f_pts[m]);
            const double up = (f_pts[m + 2] - fr) / (f_pts[m + 2] - f_pts[m + 1]);
            const double v = fmax(0.0, fmin(down, up));
            wts[b] = (float)v;
            if (v > 0.0) { if (first < 0) first = b; last = b; }
        }
        mstart[m] = first < 0 ? 0 : first;
        mcount[m] = first < 0 ? 0 : last - first + 1;
        moff[m] = (int)mw.size();
        for (int b = 0; b < mcount[m]; ++b) mw.push_back(wts[mstart[m] + b]);
    }
    if (mw.empty()) mw.push_back(0.f);
    bool ok = true;
    ok &= hipMalloc(&f->win, n_fft * sizeof(float)) == hipSuccess;
    ok &= hipMalloc(&f->tw, (n_fft / 2) * sizeof(float2)) == hipSuccess;
    ok &= hipMalloc(&f->mel_start, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_count, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_off, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_w, mw.size() * sizeof(float)) == hipSuccess;
    ok &= hipMalloc(&f->gmax, 16) == hipSuccess;
    if (!ok) { seld_feat_destroy(f); return ffail(nullptr, SELD_ERR_NOMEM, "hipMalloc failed"); }
    hipMemcpy(f->win, win.data(), n_fft * sizeof(float), hipMemcpyHostToDevice);
    hipMemcpy(f->tw, tw.data(), (n_fft / 2) * sizeof(float2), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_start, mstart.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_count, mcount.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_off, moff.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_w, mw.data(), mw.size() * sizeof(float), hipMemcpyHostToDevice);
    *out = f;
    return SELD_OK;
}

void seld_feat_destroy(seld_feat* f) {
    if (!f) return;
    hipSetDevice(f->device);
    hipDeviceSynchronize();
    hipFree(f->win); hipFree(f->tw); hipFree(f->mel_start); hipFree(f->mel_count); hipFree(f->mel_off); hipFree(f->mel_w);
    hipFree(f->gmax);
    delete f;
}

int64_t seld_feat_frames(const seld_feat* f, int64_t n_samples) { return f ? 1 + n_samples / f->hop : -1; }
int seld_feat_channels(const seld_feat* f) { return f ? (f->mode == 0 ? 7 : 10) : -1; }

int seld_feat_extract(seld_feat* f, const float* wav, int n_ch, int64_t n_samples, float* out, void* stream) {
    if (!f || !wav || !out) return SELD_ERR_INVALID;
    if (n_ch != 4) return ffail(f, SELD_ERR_UNSUPPORTED, "feature stage is built for 4-channel (FOA / MIC) audio");
    if (n_samples <= f->n_fft / 2) return ffail(f, SELD_ERR_INVALID, "clip shorter than the reflect padding (n_fft/2)");
    hipStream_t st = (hipStream_t)stream;
    const int N = f->n_fft, NBP = N / 2 + 1 + 3;
    const int64_t T = 1 + n_samples / f->hop;
    const size_t vals = f->mode == 0 ? (size_t)7 * NBP : (size_t)4 * NBP + (size_t)12 * NBP;
    const size_t smem = (size_t)(4 * N) * sizeof(float2) + (size_t)(N / 2) * sizeof(float2) + (vals + 256) * sizeof(float);
    hipLaunchKernelGGL(feat_init_max_kernel, dim3(1), dim3(1), 0, st, f->gmax);
    if (f->mode == 0) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(feat_frame_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(feat_frame_kernel<0>, dim3((unsigned)T), dim3(256), smem, st, wav, n_samples, N, f->logn, f->hop, f->n_mels,
                           f->win, f->tw, f->mel_start, f->mel_count, f->mel_off, f->mel_w, out, f->gmax);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void*>(feat_frame_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(feat_frame_kernel<1>, dim3((unsigned)T), dim3(256), smem, st, wav, n_samples, N, f->logn, f->hop, f->n_mels,
                           f->win, f->tw, f->mel_start, f->mel_count, f->mel_off, f->mel_w, out, f->gmax);
    }
    const int64_t n_tm = T * f->n_mels;
    hipLaunchKernelGGL(feat_topdb_kernel, dim3((unsigned)((n_tm * 4 + 255) / 256)), dim3(256), 0, st, out, f->gmax, n_tm,
                       f->mode == 0 ? 7 : 10, 80.f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ffail(f, SELD_ERR_HIP, std::string("feat_extract: ") + hipGetErrorString(e));
    return SELD_OK;
}

int seld_feat_normalize(const float* feat, const float* mean, const float* stdv, float* out, int64_t T_in, int64_t T_out, int FC,
                        float eps, void* stream) {
    if (!feat || !mean || !stdv || !out || T_in < 0 || T_out <= 0 || FC <= 0) return SELD_ERR_INVALID;
    const int64_t n = T_out * FC;
    hipLaunchKernelGGL(feat_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat, mean,
                       stdv, out, T_in, T_out, FC, eps);
    return hipGetLastError() == hipSuccess ? SELD_OK : SELD_ERR_HIP;
}

}  // extern "C"
