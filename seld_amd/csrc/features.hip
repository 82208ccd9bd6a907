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

#include <algorithm>
#include <math.h>
#include <string.h>
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
    int n_melw = 0;             // number of packed weights
    // the same filters for 16-byte reads (wave kernel): start rounded down to a multiple of 4 bins, weights zero-padded
    int *mel_start4 = nullptr, *mel_cnt4 = nullptr, *mel_off4 = nullptr;
    float* mel_w4 = nullptr;
    int n_melw4 = 0, maxc4 = 0;
    int* trips = nullptr;       // [16] per-slot loop bounds of the wave kernel
    int dbg = 0;                // timing ablations of the wave kernel (tools/tune_features.py): 1 no loads, 2 no FFT passes, 4 no mel
    int use_wave_kernel = 1;    // 0: the workgroup-per-frame radix-2 kernel for every size (A/B and parity of the fallback)
    uint4* dft_tab = nullptr;   // foa, n_fft 1024: constant MFMA fragments + twiddles of feat_dft_kernel (DFT_TAB_BYTES)
    uint4* mel_mm = nullptr;    // feat_dft_kernel's mel projection: [sequences][64] per-lane {first bin, weights, output} | fp32 weights, 16 per slot and mel
    int n_mel_mm = 0, mel_mm_seq = 0, mel_mm_pmax = 1;
    bool use_dft = true;
    int gmax_clips = 1;         // clips the maxima buffer holds
    float* gmax = nullptr;      // [clips][FEAT_MAX_PARTS]: per-workgroup maxima of the dB channels of the running clip (no atomics: thousands
                                // of atomic maxima on ONE word serialise at ~88 per microsecond and were 40 % of the kernel)
    std::string err;
};

#define FEAT_MAX_PARTS 4096

namespace {

std::string g_feat_err;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// MODE 0 = foa (4 mel-dB + 3 mel-IV), 1 = mic (4 mel-dB + 6 GCC)
template <int MODE>
__global__ __launch_bounds__(256) void feat_frame_kernel(const float* __restrict__ wav, int64_t n_samples, int64_t n_frames, int n_fft, int logn,
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
    for (int i = tid; i < N / 2; i += 256) tw[i] = tw_g[i];
    float lmax = -INFINITY;
    wav += (size_t)blockIdx.y * 4 * n_samples;                         // blockIdx.y = clip of a batch
    out += (size_t)blockIdx.y * n_frames * n_mels * C_OUT;
    gmax += (size_t)blockIdx.y * gridDim.x;
    for (int64_t t = blockIdx.x; t < n_frames; t += gridDim.x) {      // grid <= FEAT_MAX_PARTS: one maximum per workgroup
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
    __syncthreads();         // the next frame overwrites the buffers this one still reads
    }
    // ---- clip-wide max of the dB channels (top_db clamp needs it)
    red[tid] = lmax;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] = fmaxf(red[tid], red[tid + w]);
        __syncthreads();
    }
    if (tid == 0) gmax[blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------------------------------
// Wave-per-frame kernel (n_fft 256 .. 1024; the default path).  One WAVE owns a frame end to end, so nothing in the frame's
// chain — FFT passes, per-bin spectra, mel projection — needs a workgroup barrier: a wave's LDS instructions execute in
// order, which makes every exchange between its lanes a plain write followed by a read.  The FFT is a radix-4 Stockham
// autosort (oracle/features_oracle.py::stockham_fft_radix4 is the index model): log4(N) passes instead of log2(N), each lane
// holds N/256 butterflies of both packed transforms in registers, a pass reads ALL its inputs and then writes IN PLACE (one
// N-point buffer per transform), the first pass takes its inputs straight from global memory (windowed, reflect-padded).
// The block's four waves share the twiddle / window / mel tables in LDS and walk the frames persistently.
template <int LOGN>
struct WF {
    static constexpr int N = 1 << LOGN;
    static constexpr int NB4 = N / 256;           // radix-4 butterflies per lane and pass
    static constexpr int NB2 = N / 128;           // radix-2 butterflies per lane (final pass when LOGN is odd)
    static constexpr int P4 = LOGN / 2;
    static constexpr bool HAS2 = (LOGN & 1) != 0;
    static constexpr int NB = N / 2 + 1;
    static constexpr int NBP = NB + 3;            // row stride of the per-bin value planes
    static constexpr int NBI = (NB + 63) / 64;    // bins per lane
};

__device__ __forceinline__ float2 tw_at(const float2* tw, int t, int halfN) {   // exp(-2 pi i t / N), t in [0, N)
    float2 w = tw[t & (halfN - 1)];
    if (t & halfN) { w.x = -w.x; w.y = -w.y; }
    return w;
}
// lanes of one wave exchange data through LDS with no barrier: LDS executes a wave's instructions in order; this only stops
// the COMPILER from moving a lane's later read above its earlier write (to it they are unrelated addresses)
#define WAVE_LDS_FENCE() asm volatile("" ::: "memory")

// radix-4 passes 1 .. P4-1 and the optional final radix-2 pass of NF transforms, in place on buf[f]; `v` holds pass 0's
// inputs (element lane + 64 b + q N/4) when FROM_REGS, otherwise pass 0 reads them from buf[f] too
template <int LOGN, int NF, bool FROM_REGS>
__device__ __forceinline__ void wave_fft(float2 (&v)[NF][WF<LOGN>::NB4][4], float2* const (&buf)[NF], const float2* tw, int lane) {
    using G = WF<LOGN>;
    constexpr int N = G::N, NB4 = G::NB4, Q = N / 4;
#pragma unroll
    for (int s = 0; s < G::P4; ++s) {
        const int L = 1 << (2 * s);
        if (s > 0 || !FROM_REGS) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int b = 0; b < NB4; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[f][b][q] = buf[f][lane + 64 * b + q * Q];
        }
#pragma unroll
        for (int b = 0; b < NB4; ++b) {
            const int j = lane + 64 * b, k = j & (L - 1);
            float2 w1 = make_float2(1.f, 0.f), w2 = w1, w3 = w1;
            if (s > 0) {
                const int ts = k * (N / (4 * L));
                w1 = tw_at(tw, ts, N / 2); w2 = tw_at(tw, 2 * ts, N / 2); w3 = tw_at(tw, 3 * ts, N / 2);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float2 a0 = v[f][b][0];
                const float2 a1 = s > 0 ? cmul(w1, v[f][b][1]) : v[f][b][1];
                const float2 a2 = s > 0 ? cmul(w2, v[f][b][2]) : v[f][b][2];
                const float2 a3 = s > 0 ? cmul(w3, v[f][b][3]) : v[f][b][3];
                const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y), t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
                const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y);
                const float2 t3 = make_float2(a1.y - a3.y, a3.x - a1.x);          // -i (a1 - a3)
                v[f][b][0] = make_float2(t0.x + t2.x, t0.y + t2.y);
                v[f][b][1] = make_float2(t1.x + t3.x, t1.y + t3.y);
                v[f][b][2] = make_float2(t0.x - t2.x, t0.y - t2.y);
                v[f][b][3] = make_float2(t1.x - t3.x, t1.y - t3.y);
            }
        }
        WAVE_LDS_FENCE();        // every input of the pass is in registers before the first in-place write
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int b = 0; b < NB4; ++b) {
                const int j = lane + 64 * b, k = j & (L - 1), o = 4 * (j - k) + k;
#pragma unroll
                for (int p = 0; p < 4; ++p) buf[f][o + p * L] = v[f][b][p];
            }
        WAVE_LDS_FENCE();
    }
    if (G::HAS2) {               // L = N/2: k = j, twiddle tw[j]
        constexpr int NB2 = G::NB2;
        float2 u[NF][NB2], x[NF][NB2];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int b = 0; b < NB2; ++b) {
                const int j = lane + 64 * b;
                u[f][b] = buf[f][j];
                x[f][b] = cmul(tw[j], buf[f][j + N / 2]);
            }
        WAVE_LDS_FENCE();
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int b = 0; b < NB2; ++b) {
                const int j = lane + 64 * b;
                buf[f][j] = make_float2(u[f][b].x + x[f][b].x, u[f][b].y + x[f][b].y);
                buf[f][j + N / 2] = make_float2(u[f][b].x - x[f][b].x, u[f][b].y - x[f][b].y);
            }
        WAVE_LDS_FENCE();
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// windowed samples of channels (xa, xa + n_samples) of the frame that starts at sample s0 -> the first pass's registers
// (element lane + 64 b + q N/4 of the packed complex input); reflect padding only where the frame leaves the clip
template <int LOGN>
__device__ __forceinline__ void feat_load_pair(float2 (&v)[1][WF<LOGN>::NB4][4], const float* xa, int, int64_t n_samples, int64_t s0,
                                               bool interior, const float* win, int lane) {
    constexpr int NB4 = WF<LOGN>::NB4, Q = WF<LOGN>::N / 4;
    const float* xb = xa + n_samples;
    if (interior) {
        const float* pa = xa + s0;
        const float* pb = xb + s0;
#pragma unroll
        for (int b = 0; b < NB4; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = lane + 64 * b + q * Q;
                const float w = win[n];
                v[0][b][q] = make_float2(w * pa[n], w * pb[n]);
            }
    } else {
#pragma unroll
        for (int b = 0; b < NB4; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = lane + 64 * b + q * Q;
                int64_t i = s0 + n;
                if (i < 0) i = -i;
                if (i >= n_samples) i = 2 * (n_samples - 1) - i;
                const float w = win[n];
                v[0][b][q] = make_float2(w * xa[i], w * xb[i]);
            }
    }
}

#define FEAT_WAVES_MAX 16
// waves per workgroup of feat_wave_kernel: 16 (four per SIMD, 128 registers) except the 1 024-point mic kernel, whose four spectra
// spill at that cap: 12 (0.080 against 0.090 ms per clip; foa at n_fft 512: 0.034 with 16 against 0.038 with 12)
#define FEAT_WAVES_OF(MODE_, LOGN_) (((MODE_) == 1 && (LOGN_) == 10) ? 12 : 16)
// LDS of one wave: ONE transform buffer; the same bytes later hold four per-bin value planes [4][NB + 3] and then the staged frame
static size_t feat_wave_bytes(int n_fft, int n_mels, int c_out) {
    const size_t fft = (size_t)n_fft * sizeof(float2), planes = (size_t)4 * (n_fft / 2 + 4) * sizeof(float);
    const size_t stage = (size_t)n_mels * c_out * sizeof(float);
    size_t m = fft > planes ? fft : planes;
    if (stage > m) m = stage;
    return (m + 15) & ~(size_t)15;
}

// Sequence per frame (foa): channels 0,1 -> FFT -> this lane's bins to registers (powers, Re(conj(W) Y)); channels 2,3 -> FFT in
// the SAME buffer -> bins (powers, the other two intensity components, normalisation); the four power planes -> sparse mel + dB;
// the three intensity planes -> sparse mel; the frame staged as [m][7] and stored as one contiguous run.  mic: the four spectra
// stay in registers and three packed inverse transforms give the six GCC-PHAT cross-correlations.
// The mel filters are read 16 bytes at a time: filter m covers cnt4[m] aligned float4 chunks of a plane from bin start4[m]
// (a multiple of 4), its weights zero-padded to match (seld_feat_create).
template <int MODE, int LOGN>
__global__ __launch_bounds__(64 * FEAT_WAVES_OF(MODE, LOGN), 1) void feat_wave_kernel(
    const float* __restrict__ wav, int64_t n_samples, int64_t T, int hop, int n_mels, int n_melw4, int maxc4,
    const float* __restrict__ win_g, const float2* __restrict__ tw_g, const int* __restrict__ mel_start4,
    const int* __restrict__ mel_cnt4, const int* __restrict__ mel_off4, const float* __restrict__ mel_w4, float* __restrict__ out,
    float* __restrict__ gmax, int wave_bytes, int dbg, const int* __restrict__ trips_g) {
    using G = WF<LOGN>;
    constexpr int FEAT_WAVES = FEAT_WAVES_OF(MODE, LOGN);
    constexpr int N = G::N, NB = G::NB, NBP = G::NBP, NBI = G::NBI, NB4 = G::NB4;
    constexpr int C_OUT = MODE == 0 ? 7 : 10;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* tw = reinterpret_cast<float2*>(smem);                  // [N/2]
    float* win = smem + N;                                         // [N]
    float* melw = win + N;                                         // [n_melw4] (a multiple of 4)
    int* mst = reinterpret_cast<int*>(melw + n_melw4);             // [n_mels] start4 | cnt4 | offset
    int* mct = mst + n_mels;
    int* mof = mct + n_mels;
    int* trips = mof + ((n_mels + 3) & ~3);                       // [16]
    const int tid = threadIdx.x, lane_id = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* wbase = reinterpret_cast<char*>(trips + 16) + (size_t)wave * wave_bytes;
    float2* buf = reinterpret_cast<float2*>(wbase);
    float* val = reinterpret_cast<float*>(wbase);
    for (int i = tid; i < N / 2; i += 64 * FEAT_WAVES) tw[i] = tw_g[i];
    for (int i = tid; i < N; i += 64 * FEAT_WAVES) win[i] = win_g[i];
    for (int i = tid; i < n_melw4; i += 64 * FEAT_WAVES) melw[i] = mel_w4[i];
    for (int i = tid; i < n_mels; i += 64 * FEAT_WAVES) { mst[i] = mel_start4[i]; mct[i] = mel_cnt4[i]; mof[i] = mel_off4[i]; }
    __syncthreads();
    // trips[q]: wave-uniform trip count of output slot q's filter loop = the longest filter among the slot's 64 outputs (slot q of
    // the dB pass holds m = (lane + 64 q) >> 2, slot 8 + q of the intensity pass m = (lane + 64 q) / 3); built by seld_feat_create
    if (tid < 16) trips[tid] = trips_g[tid];
    __syncthreads();
    float lmax = -INFINITY;
    float2* const b1[1] = {buf};
    // a batch of clips of one length: blockIdx.y = clip ([clips][4][n_samples] in, [clips][T][n_mels][C_OUT] out, maxima per clip)
    wav += (size_t)blockIdx.y * 4 * n_samples;
    out += (size_t)blockIdx.y * T * n_mels * C_OUT;
    gmax += (size_t)blockIdx.y * gridDim.x;
    for (int64_t t = (int64_t)blockIdx.x * FEAT_WAVES + wave; t < T; t += (int64_t)gridDim.x * FEAT_WAVES) {
        // `lane` made opaque per frame: everything indexed by it (twiddles of every pass, window, table offsets) is otherwise
        // hoisted out of this loop and held in ~100 registers for the one or two frames a wave sees
        int lane = lane_id;
        asm volatile("" : "+v"(lane));
        const int64_t s0 = t * hop - N / 2;                        // first sample of the frame (torch.stft center=True)
        const bool interior = s0 >= 0 && s0 + N <= n_samples;      // wave-uniform: no reflection needed
        float2 v[1][NB4][4];
        // Xa = (Z[k] + conj(Z[N-k]))/2 ; Xb = (Z[k] - conj(Z[N-k]))/(2i): this lane's bins k = lane + 64 i.  foa keeps only what the
        // planes need (four powers, X0 until the second transform is done, three intensity components); mic keeps the four spectra.
        float2 X0[NBI], X1[MODE == 1 ? NBI : 1], X2[MODE == 1 ? NBI : 1], X3[MODE == 1 ? NBI : 1];
        float P[4][NBI], IV[MODE == 0 ? 3 : 1][NBI];
        if (!(dbg & 1)) feat_load_pair<LOGN>(v, wav, 0, n_samples, s0, interior, win, lane);
        if (!(dbg & 2)) wave_fft<LOGN, 1, true>(v, b1, tw, lane);
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int k = lane + 64 * i, kk = k < NB ? k : 0, kn = (N - kk) & (N - 1);
            const float2 a = buf[kk], b = buf[kn];
            X0[i] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
            const float2 x1 = make_float2(0.5f * (a.y + b.y), 0.5f * (b.x - a.x));
            P[0][i] = X0[i].x * X0[i].x + X0[i].y * X0[i].y;
            P[1][i] = x1.x * x1.x + x1.y * x1.y;
            if (MODE == 0) IV[1][i] = X0[i].x * x1.x + X0[i].y * x1.y;      // IVy <- ch1: Re(conj(W) X1)
            else X1[i] = x1;
        }
        WAVE_LDS_FENCE();
        if (!(dbg & 1)) feat_load_pair<LOGN>(v, wav + 2 * n_samples, 0, n_samples, s0, interior, win, lane);
        if (!(dbg & 2)) wave_fft<LOGN, 1, true>(v, b1, tw, lane);
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int k = lane + 64 * i, kk = k < NB ? k : 0, kn = (N - kk) & (N - 1);
            const float2 a = buf[kk], b = buf[kn];
            const float2 x2 = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
            const float2 x3 = make_float2(0.5f * (a.y + b.y), 0.5f * (b.x - a.x));
            P[2][i] = x2.x * x2.x + x2.y * x2.y;
            P[3][i] = x3.x * x3.x + x3.y * x3.y;
            if (MODE == 0) {
                // intensity vector Re(conj(W) X_i) / |.|: IVx <- ch3, IVy <- ch1, IVz <- ch2
                const float ivx = X0[i].x * x3.x + X0[i].y * x3.y, ivy = IV[1][i], ivz = X0[i].x * x2.x + X0[i].y * x2.y;
                // one reciprocal instead of three divisions (v_rcp_f32: 1 ulp; the parity bar is 1e-4 of the channel's maximum)
                const float inv = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(ivx * ivx + ivy * ivy + ivz * ivz), 1e-8f));
                IV[0][i] = ivx * inv; IV[1][i] = ivy * inv; IV[2][i] = ivz * inv;
            } else {
                X2[i] = x2; X3[i] = x3;
            }
        }
        WAVE_LDS_FENCE();
        // ---- power planes [4][NBP] (pad entries zeroed: the filters read whole float4 chunks)
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int k = lane + 64 * i;
            if (k < NB) {
#pragma unroll
                for (int c = 0; c < 4; ++c) val[c * NBP + k] = P[c][i];
            }
        }
        if (lane < 4 * (NBP - NB)) val[(lane / (NBP - NB)) * NBP + NB + lane % (NBP - NB)] = 0.f;
        WAVE_LDS_FENCE();
        auto mel_dot = [&](int m, int c, int trips) -> float {
            const int c4n = mct[m];
            const float4* wv = reinterpret_cast<const float4*>(melw + mof[m]);
            const float4* vv = reinterpret_cast<const float4*>(val + c * NBP + mst[m]);
            float acc0 = 0.f, acc1 = 0.f;             // two chunks per trip: both pairs of LDS reads in flight before the first FMA
            for (int i = 0; i < trips; i += 2) {
                const bool p0 = i < c4n, p1 = i + 1 < c4n;
                const int i0 = p0 ? i : 0, i1 = p1 ? i + 1 : 0;
                const float4 w0 = (dbg & 8) ? make_float4(1.f, 2.f, 3.f, (float)i) : wv[i0], x0 = (dbg & 8) ? make_float4(1.f, 1.f, 1.f, 1.f) : vv[i0];
                const float4 w1 = (dbg & 8) ? make_float4(1.f, 2.f, 3.f, (float)i) : wv[i1], x1 = (dbg & 8) ? make_float4(1.f, 1.f, 1.f, 1.f) : vv[i1];
                if (p0) { acc0 = fmaf(w0.x, x0.x, acc0); acc0 = fmaf(w0.y, x0.y, acc0); acc0 = fmaf(w0.z, x0.z, acc0); acc0 = fmaf(w0.w, x0.w, acc0); }
                if (p1) { acc1 = fmaf(w1.x, x1.x, acc1); acc1 = fmaf(w1.y, x1.y, acc1); acc1 = fmaf(w1.z, x1.z, acc1); acc1 = fmaf(w1.w, x1.w, acc1); }
            }
            return acc0 + acc1;
        };
        // log-mel: element (m = idx >> 2, c = idx & 3) of the frame, idx = lane + 64 q: 16 contiguous bytes per mel band
        float* frame_out = out + (size_t)t * n_mels * C_OUT;
        if (!(dbg & 4))
        for (int q = 0; 64 * q < 4 * n_mels; ++q) {
            const int idx = lane + 64 * q;
            const int tr = __builtin_amdgcn_readfirstlane(trips[q]);
            if (idx < 4 * n_mels) {
                const float md = mel_dot(idx >> 2, idx & 3, tr);
                const float o = (dbg & 32) ? md : 3.0102999566f * __builtin_amdgcn_logf(fmaxf(md, 1e-10f));   // 10 log10 x = 10 log10(2) log2 x (v_log_f32)
                lmax = fmaxf(lmax, o);
                if (!(dbg & 16)) frame_out[(idx >> 2) * C_OUT + (idx & 3)] = o;
            }
        }
        WAVE_LDS_FENCE();
        if (MODE == 0) {
            // ---- the three intensity planes through the same filters, no dB
#pragma unroll
            for (int i = 0; i < NBI; ++i) {
                const int k = lane + 64 * i;
                if (k < NB) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) val[c * NBP + k] = IV[c][i];
                }
            }
            WAVE_LDS_FENCE();                                      // (the pad entries are still zero)
            if (!(dbg & 4))
            for (int q = 0; 64 * q < 3 * n_mels; ++q) {
                const int idx = lane + 64 * q;
                const int tr = __builtin_amdgcn_readfirstlane(trips[8 + q]);
                if (idx < 3 * n_mels) {
                    const float md = mel_dot(idx / 3, idx % 3, tr);
                    if (!(dbg & 16)) frame_out[(idx / 3) * C_OUT + 4 + idx % 3] = md;
                    else lmax = fmaxf(lmax, md);
                }
            }
        } else {
            // ---- GCC-PHAT: cc = irfft(exp(i angle(conj(Xa) Xb))) for the 6 pairs, two pairs per complex transform:
            // Z = P1 + i P2 on the Hermitian-extended spectrum, ifft(Z) = conj(fft(conj(Z))) / N = cc1 + i cc2
            const float invn = 1.f / (float)N;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int i = 0; i < NBI; ++i) {
                    const int k = lane + 64 * i;
                    float2 Pp[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int pr = 2 * g + h;                  // pair order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
                        const float2 xa = pr < 3 ? X0[i] : (pr < 5 ? X1[i] : X2[i]);
                        const float2 xb = pr == 0 ? X1[i] : ((pr == 1 || pr == 3) ? X2[i] : X3[i]);
                        const float rr = xa.x * xb.x + xa.y * xb.y, ri = xa.x * xb.y - xa.y * xb.x;
                        const float m2 = rr * rr + ri * ri, inv = __builtin_amdgcn_rsqf(m2);
                        Pp[h] = m2 > 0.f ? make_float2(rr * inv, ri * inv) : make_float2(1.f, 0.f);   // angle(0) = 0
                    }
                    if (k < NB) {
                        const int kn = (N - k) & (N - 1);
                        buf[kn] = make_float2(Pp[0].x + Pp[1].y, Pp[0].y - Pp[1].x);      // conj(Z[N-k]), Z[N-k] = conj(P1) + i conj(P2)
                        buf[k] = make_float2(Pp[0].x - Pp[1].y, -(Pp[0].y + Pp[1].x));   // conj(Z[k]) (k = 0, N/2: the same value twice)
                    }
                }
                WAVE_LDS_FENCE();
                wave_fft<LOGN, 1, false>(v, b1, tw, lane);
                for (int j = lane; j < n_mels; j += 64) {
                    const float2 r = buf[(j - n_mels / 2) & (N - 1)];
                    frame_out[j * C_OUT + 4 + 2 * g] = r.x * invn;
                    frame_out[j * C_OUT + 5 + 2 * g] = -r.y * invn;
                }
                WAVE_LDS_FENCE();
            }
        }
        WAVE_LDS_FENCE();
    }
    // ---- clip-wide max of the dB channels (top_db clamp needs it)
    lmax = wave_max(lmax);
    __syncthreads();                       // every wave is done with the tables: trips[] is reused for the waves' maxima
    if (lane_id == 0) reinterpret_cast<float*>(trips)[wave] = lmax;
    __syncthreads();
    if (tid == 0) {
        float m = -INFINITY;
        for (int w = 0; w < FEAT_WAVES; ++w) m = fmaxf(m, reinterpret_cast<float*>(trips)[w]);
        gmax[blockIdx.x] = m;
    }
}


// ------------------------------------------------------------------------------------------------
// feat_dft_kernel — foa, n_fft = 1024: the transform on the MATRIX cores.  1024 = 32 x 32: with n = 32 n1 + n2, k = k1 + 32 k2,
//     X[k1 + 32 k2] = sum_n2 W32^(n2 k2) . W1024^(n2 k1) . [ sum_n1 x[32 n1 + n2] W32^(n1 k1) ]
// is two 32 x 32 x 32 products with a pointwise twiddle between them, and a frame's data never leaves the registers of ITS wave:
//   step 1   Y^T[n2][k1] = x^T F32          A operand = the frame as loaded (lane n2 holds x[32 n1 + n2], 8 consecutive n1 per k-group)
//   twiddle  Z = Y . W1024^(n2 k1)          in the accumulator layout (row n2 = mfma_row(r, g), column k1 = lane & 31)
//   step 3   [Fr; Fi] Z                      B operand = Z exactly where it lies: the K slots are assigned to the rows a lane holds (the
//            constant A operand is built in the same order); rows 0..15 of the stacked constant matrix are Re W32^(n2 k2), rows 16..31
//            Im, so X_re[k2] = P[r] - Q[r + 8], X_im[k2] = Q[r] + P[r + 8] are lane-local (P = [Fr; Fi] Zr, Q = [Fr; Fi] Zi).
// Operands: fp32 values split into TWO f16 terms (hi + lo, 22 bits) and three products (hi hi, lo hi, hi lo) accumulated in fp32;
// the frame is scaled by a power of two so that its largest sample sits in [2^13, 2^14), the constant matrices by 2^10, the twiddle
// table carries 2^-15: every f16 operand stays in the normal range and every scaling is exact.  No LDS passes, no barriers: LDS holds
// the constant fragments, the twiddles and — as in feat_wave_kernel, whose second half this kernel shares — the per-bin planes of the
// sparse mel projection.  A lane ends with bins k = (lane & 31) + 32 k2, k2 = mfma_row(i, lane >> 5), i < 8 (+ bin 512 on lane 0).
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define DFT_WAVES 12                                           // foa; the mic form keeps all four spectra in registers: 8 waves (256 VGPRs)
#define DFT_WAVES_MIC 8
#define DFT_WAVES_OF(MODE_) ((MODE_) == 0 ? DFT_WAVES : DFT_WAVES_MIC)
// mic: the constants of the GCC-PHAT inverse transform (see the kernel): F [re | im | -im][hi | lo][64] x 16 B, then the outer stage's [16][64] float4
#define GCC_TAB_BYTES (6 * 1024 + 16 * 1024)
#define DFT_TAB_BYTES (8 * 1024 + 4 * 1024 + 8 * 1024)        // f1 [2][re|im][hi|lo][64] x 16 B | a3 [2][hi|lo][64] x 16 B | tw2 [16][64] float2

// wave-wide maximum without LDS traffic: two quad permutes and two row rotations leave every lane of a 16-lane row with the row's
// maximum, four v_readlane fetch the rows' values (__shfl_xor is ds_bpermute: six dependent LDS round trips per reduction)
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));     // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));     // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true)));    // row_ror:4
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true)));    // row_ror:8
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// two fp32 values -> their f16 hi terms (truncation: the fp32 bits masked to 10 mantissa bits IS the round-toward-zero f16 in the
// normal range; below it the mask keeps more than f16 does, an error of at most one f16 subnormal step, 2^-38 of the frame's largest
// sample) and lo terms (the exact remainders, truncated), packed as elements j, j + 1 of the two fragments
__device__ __forceinline__ void dft_split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {      // one packed word each
    const float h0 = __uint_as_float(__float_as_uint(v0) & 0xffffe000u), h1 = __uint_as_float(__float_as_uint(v1) & 0xffffe000u);
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v0 - h0, v1 - h1));
}
typedef unsigned dft_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h16x8 dft_frag(const unsigned (&w)[4]) { return __builtin_bit_cast(h16x8, dft_u32x4{w[0], w[1], w[2], w[3]}); }

__device__ __forceinline__ int dft_bin(int i, int lane) {      // NB (= out of range) for the slots a lane does not own
    if (i < 8) return (lane & 31) + 32 * ((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5));
    return lane == 0 ? 512 : 513;
}

// -DFEAT_TRACE (diagnostic build, tools/trace_feat.py): s_memtime stamps of every wave of workgroup (0, 0) over its frames 2 and 3:
// frame start | after channels 0..3 | after the intensity normalisation | log-mel rows | intensity rows | frame stored.
#ifdef FEAT_TRACE
__device__ unsigned long long g_feat_trace[DFT_WAVES][2][10];      // (foa form: 12 waves)
#define FEAT_TR(k_)                                                                                      \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (trace_it == 2 || trace_it == 3)) {                        \
        unsigned long long t_;                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if ((threadIdx.x & 63) == 0) g_feat_trace[threadIdx.x >> 6][trace_it - 2][k_] = t_;             \
    }
#else
#define FEAT_TR(k_)
#endif

template <int QSEQ, int MODE>
__global__ __launch_bounds__(64 * DFT_WAVES_OF(MODE), 1) void feat_dft_kernel(
    const float* __restrict__ wav, int64_t n_samples, int64_t T, int hop, int n_mels, const float* __restrict__ win_g,
    const uint4* __restrict__ dft_g, const uint4* __restrict__ mm_g, int n_mm, int mm_pmax, float* __restrict__ out, float* __restrict__ gmax,
    float* __restrict__ gmin, int wave_bytes) {
    constexpr int N = 1024, NB = 513, NBP = 516, NBI = 9, C_OUT = MODE == 0 ? 7 : 10, WAVES = DFT_WAVES_OF(MODE);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* win = smem;                                             // [N]
    float* red = win + N;                                          // [2][16]: the waves' maxima / minima at the very end
    uint4* mm = reinterpret_cast<uint4*>(red + 32);                // [QSEQ][64] per-lane slot descriptions | [64] per-mel slot ranges | weights (see mel_round)
    char* dtab = reinterpret_cast<char*>(mm + n_mm);
    const h16x8* f1 = reinterpret_cast<const h16x8*>(dtab);                       // [(s * 2 + c) * 2 + t][lane]
    const h16x8* a3 = reinterpret_cast<const h16x8*>(dtab + 8 * 1024);            // [s * 2 + t][lane]
    const float2* tw2 = reinterpret_cast<const float2*>(dtab + 12 * 1024);        // [r][lane]
    const int tid = threadIdx.x, lane_id = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* gtab = dtab + DFT_TAB_BYTES;                                                       // mic: GCC_TAB_BYTES of inverse-transform constants
    float* val = reinterpret_cast<float*>(dtab + DFT_TAB_BYTES + (MODE == 1 ? GCC_TAB_BYTES : 0) + (size_t)wave * wave_bytes);      // the wave's four per-bin planes [4][NBP]
    float* stage = val + 4 * NBP;                                  // [n_mels][7] (+ 8 floats the idle blocks add into): the frame's outputs, stored as ONE contiguous run
    for (int i = tid; i < N; i += 64 * WAVES) win[i] = win_g[i];
    for (int i = tid; i < n_mm; i += 64 * WAVES) mm[i] = mm_g[i];
    for (int i = tid; i < (DFT_TAB_BYTES + (MODE == 1 ? GCC_TAB_BYTES : 0)) / 16; i += 64 * WAVES) reinterpret_cast<uint4*>(dtab)[i] = dft_g[i];
    __syncthreads();
    float lmax = -INFINITY, lmin = INFINITY;
    wav += (size_t)blockIdx.y * 4 * n_samples;                     // blockIdx.y = clip of a batch
    out += (size_t)blockIdx.y * T * n_mels * C_OUT;
    gmax += (size_t)blockIdx.y * gridDim.x;
    gmin += (size_t)blockIdx.y * gridDim.x;
    float xr[16];                                                  // the samples the next channel iteration works on
    bool first_frame = true;
    int trace_it = -1;
    for (int64_t t = (int64_t)blockIdx.x * WAVES + wave; t < T; t += (int64_t)gridDim.x * WAVES) {
        ++trace_it;
        FEAT_TR(0)
        int lane = lane_id;
        asm volatile("" : "+v"(lane));                             // see feat_wave_kernel: keeps lane-indexed values out of LICM
        const int li = lane & 31, g = lane >> 5;
        float2 X0[NBI];
        float IV[3][NBI];
        float2 XS[MODE == 1 ? 3 : 1][NBI];                        // mic: the spectra of channels 1 .. 3 at this lane's bins (channel 0 in X0)
        // the window at this lane's 16 sample slots (slot u -> n1 = 16 (u >> 3) + 8 g + (u & 7), n = 32 n1 + li): the same for the four channels
        const float* wn = win + 256 * g + li;                    // slot u at offset 512 (u >> 3) + 32 (u & 7): re-read per channel (16 registers fewer)
        // raw samples of (frame, channel) in A-operand order; channel c + 1 is requested before channel c is processed, so that a wave
        // waits for global memory once per frame instead of four times (-5 % same box, although 16 more registers spill a little)
        auto load16 = [&](float (&dst)[16], int64_t tt, int cc) {
            const int64_t s0_ = tt * hop - N / 2;
            const bool interior_ = s0_ >= 0 && s0_ + N <= n_samples;
            const float* xc_ = wav + (size_t)cc * n_samples;
            if (interior_) {                                   // wave-uniform: one base address per lane, compile-time offsets
                const float* p_ = xc_ + (s0_ + 256 * g + li);
#pragma unroll
                for (int u = 0; u < 16; ++u) dst[u] = p_[512 * (u >> 3) + 32 * (u & 7)];
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) {           // (sample indices of one clip fit 32 bits: checked by the launcher)
                    int i = (int)s0_ + 512 * (u >> 3) + 256 * g + 32 * (u & 7) + li;
                    if (i < 0) i = -i;
                    if (i >= (int)n_samples) i = 2 * ((int)n_samples - 1) - i;
                    dst[u] = xc_[i];
                }
            }
        };
        const int64_t t_next = t + (int64_t)gridDim.x * WAVES < T ? t + (int64_t)gridDim.x * WAVES : t;
        float xn[16];
        if (first_frame) { load16(xr, t, 0); first_frame = false; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) load16(xn, t, c + 1);                   // the next channel; after the last one: channel 0 of this wave's NEXT frame
            else load16(xn, t_next, 0);
            __builtin_amdgcn_sched_barrier(0);
            float xw[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) xw[u] = wn[512 * (u >> 3) + 32 * (u & 7)] * xr[u];
            float amax = 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) amax = fmaxf(amax, fabsf(xw[u]));
            amax = wave_max_dpp(amax);
            // power-of-two scale per channel and frame: largest sample into [2^13, 2^14)
            const int ex = amax > 0.f ? __builtin_amdgcn_frexp_expf(amax) : 14;       // amax = f 2^ex, f in [0.5, 1)
            const float sc_in = __builtin_amdgcn_ldexpf(1.f, 14 - ex);
            const float sc_out = __builtin_amdgcn_ldexpf(1.f, ex - 19);               // X = acc3 2^-5 / sc_in
            const float sc_ny = __builtin_amdgcn_ldexpf(1.f, ex - 24);                // X[512] = sum(+-acc1) 2^-10 / sc_in
            unsigned ahw[2][4], alw[2][4];
#pragma unroll
            for (int u = 0; u < 16; u += 2) dft_split_pair(xw[u] * sc_in, xw[u + 1] * sc_in, ahw[u >> 3][(u & 7) >> 1], alw[u >> 3][(u & 7) >> 1]);
            const h16x8 ah[2] = {dft_frag(ahw[0]), dft_frag(ahw[1])}, al[2] = {dft_frag(alw[0]), dft_frag(alw[1])};
            f32x16 yr = zero16(), yi = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const h16x8 frh = f1[((s * 2 + 0) * 2 + 0) * 64 + lane], frl = f1[((s * 2 + 0) * 2 + 1) * 64 + lane];
                const h16x8 fih = f1[((s * 2 + 1) * 2 + 0) * 64 + lane], fil = f1[((s * 2 + 1) * 2 + 1) * 64 + lane];
                yr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], frh, yr, 0, 0, 0);
                yi = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], fih, yi, 0, 0, 0);
                yr = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], frh, yr, 0, 0, 0);
                yi = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], fih, yi, 0, 0, 0);
                yr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], frl, yr, 0, 0, 0);
                yi = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], fil, yi, 0, 0, 0);
            }
            // bin 512 = sum_n2 (-1)^n2 Y[n2][k1 = 0] (real): the rows of this lane have parity r & 1
            float ny = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ny += (r & 1) ? -yr[r] : yr[r];
            ny = (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(ny), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ny), 32))) * sc_ny;   // lanes 0 / 32 hold k1 = 0
            unsigned zrhw[2][4], zrlw[2][4], zihw[2][4], zilw[2][4];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float2 w0 = tw2[r * 64 + lane], w1 = tw2[(r + 1) * 64 + lane];
                const float zr0 = fmaf(yr[r], w0.x, -yi[r] * w0.y), zi0 = fmaf(yr[r], w0.y, yi[r] * w0.x);
                const float zr1 = fmaf(yr[r + 1], w1.x, -yi[r + 1] * w1.y), zi1 = fmaf(yr[r + 1], w1.y, yi[r + 1] * w1.x);
                dft_split_pair(zr0, zr1, zrhw[r >> 3][(r & 7) >> 1], zrlw[r >> 3][(r & 7) >> 1]);
                dft_split_pair(zi0, zi1, zihw[r >> 3][(r & 7) >> 1], zilw[r >> 3][(r & 7) >> 1]);
            }
            const h16x8 zrh[2] = {dft_frag(zrhw[0]), dft_frag(zrhw[1])}, zrl[2] = {dft_frag(zrlw[0]), dft_frag(zrlw[1])};
            const h16x8 zih[2] = {dft_frag(zihw[0]), dft_frag(zihw[1])}, zil[2] = {dft_frag(zilw[0]), dft_frag(zilw[1])};
            f32x16 Pa = zero16(), Qa = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const h16x8 ch = a3[(s * 2 + 0) * 64 + lane], cl = a3[(s * 2 + 1) * 64 + lane];
                Pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, zrh[s], Pa, 0, 0, 0);
                Qa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, zih[s], Qa, 0, 0, 0);
                Pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, zrl[s], Pa, 0, 0, 0);
                Qa = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, zil[s], Qa, 0, 0, 0);
                Pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, zrh[s], Pa, 0, 0, 0);
                Qa = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, zih[s], Qa, 0, 0, 0);
            }
            // this lane's bins: power straight into the channel's plane, intensity components kept
#pragma unroll
            for (int i = 0; i < NBI; ++i) {
                const int i8 = i < 8 ? i : 0;
                const float2 xc = i < 8 ? make_float2((Pa[i8] - Qa[i8 + 8]) * sc_out, (Qa[i8] + Pa[i8 + 8]) * sc_out) : make_float2(ny, 0.f);
                const int k = dft_bin(i, lane);
                if (i < 8 || k < NB) val[c * NBP + k] = xc.x * xc.x + xc.y * xc.y;
                if (c == 0) X0[i] = xc;
                else if (MODE == 1) XS[c - 1][i] = xc;
                else IV[c == 3 ? 0 : c][i] = X0[i].x * xc.x + X0[i].y * xc.y;      // IVx <- ch3, IVy <- ch1, IVz <- ch2: Re(conj(W) X_c)
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) xr[u] = xn[u];
            FEAT_TR(1 + c)
        }
#pragma unroll
        for (int i = 0; MODE == 0 && i < NBI; ++i) {
            const float ivx = IV[0][i], ivy = IV[1][i], ivz = IV[2][i];
            const float inv = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(ivx * ivx + ivy * ivy + ivz * ivz), 1e-8f));
            IV[0][i] = ivx * inv; IV[1][i] = ivy * inv; IV[2][i] = ivz * inv;
        }
        FEAT_TR(5)
        // The mel projection on the matrix cores, in fp32: v_mfma_f32_4x4x1_16B_f32 is SIXTEEN independent 4 x 4 x 1 products, one per
        // group of four lanes.  A "slot" = four mels x sixteen consecutive bins (sixteen steps of K = 1): mel group g (mels 4 g .. 4 g + 3,
        // whose filters overlap: a run of ~2.5 filter lengths) is cut into ceil(run / 16) slots, and the slots of all groups are dealt to
        // the 16 blocks of QSEQ successive sequences (64-mel, 513-bin bank: 41 slots, 3 sequences, 48 instructions per round; one block
        // per group would take 132: the top group alone runs 132 bins).  A = the four planes at the slot's bins (row i = lane & 3 =
        // channel), B = the slot's weights (column j = lane & 3 = mel 4 g + j; 64 bytes per slot and mel), D[channel r][mel j] in VGPR r.
        // The slots' sums then meet through LDS, added per mel in slot order (see below).
        // Measured, same box (8 x 60 s per launch): 41.0 -> 44.6 k clips/s with the top_db early exit.  The per-lane sparse dot products
        // this replaces were ~2 000 of a frame's ~5 500 vector instructions; the f16 form of the same blocks (4 x 4 x 4, hi / lo terms,
        // 36 instructions per round) was slower: the hi / lo planes cost 16-bit LDS writes that took longer than the products saved.
        auto mel_round = [&]() -> f32x4 {
            if (lane < 4 * (NBP - NB)) val[(lane / (NBP - NB)) * NBP + NB + lane % (NBP - NB)] = 0.f;      // bins 513 .. 515 (a slot reads sixteen bins)
            WAVE_LDS_FENCE();
            f32x4 part[QSEQ];
#pragma unroll
            for (int q = 0; q < QSEQ; ++q) {
                const uint4 e = mm[q * 64 + lane];                 // {the slot's first bin, its weights (16-byte units)}
                const float4* pa = reinterpret_cast<const float4*>(val + (lane & 3) * NBP + e.x);
                const float4* pw = reinterpret_cast<const float4*>(mm + e.y);
                float4 a[4], w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[u] = pa[u]; w[u] = pw[u]; }
                f32x4 d[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].x, w[u].x, z, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].y, w[u].y, d[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].z, w[u].z, d[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) d[u] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u].w, w[u].w, d[u], 0, 0, 0);
                part[q] = (d[0] + d[1]) + (d[2] + d[3]);
            }
            // the slots' sums -> LDS, slot by slot (the planes are done with); every lane (= mel) then adds up the slots of its group, in
            // slot order.  (LDS float atomics into the output row instead: 12 instructions, and the kernel took 1.8 x as long.)
            WAVE_LDS_FENCE();
            f32x4* ps = reinterpret_cast<f32x4*>(val);
#pragma unroll
            for (int q = 0; q < QSEQ; ++q) ps[q * 64 + lane] = part[q];
            WAVE_LDS_FENCE();
            const uint4 ge = mm[QSEQ * 64 + lane];                 // {4 x the first slot of the lane's mel group + (lane & 3), the number of its slots}
            const f32x4* pg = ps + ge.x;
            f32x4 acc = pg[0];
#pragma unroll
            for (int j = 1; j < 12; ++j) {
                if (j >= mm_pmax) break;
                if (j < (int)ge.y) acc += pg[4 * j];
            }
            WAVE_LDS_FENCE();
            return acc;
        };
        float* frame_out = out + (size_t)t * n_mels * C_OUT;
        {
            const f32x4 d = mel_round();
            if (lane < n_mels) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float o = 3.0102999566f * __builtin_amdgcn_logf(fmaxf(d[r], 1e-10f));
                    lmax = fmaxf(lmax, o);
                    lmin = fminf(lmin, o);
                    stage[lane * C_OUT + r] = o;
                }
            }
        }
        FEAT_TR(6)
        if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const int k = dft_bin(i, lane);
            if (k < NB) {
#pragma unroll
                for (int c = 0; c < 3; ++c) val[c * NBP + k] = IV[c][i];
            }
        }
        WAVE_LDS_FENCE();
        {
            const f32x4 d = mel_round();
            if (lane < n_mels) {
#pragma unroll
                for (int r = 0; r < 3; ++r) stage[lane * C_OUT + 4 + r] = d[r];
            }
        }
        WAVE_LDS_FENCE();
        } else {
        // ---- GCC-PHAT (feature_extractor.py:196-214): cc = irfft(exp(i angle(conj(Xa) Xb))), lags -n_mels/2 .. n_mels/2 - 1, for the six pairs.
        // The inverse transform as 32 x 32 products too, PRUNED to the 64 lags that are kept.  With k = k1 + 32 k2 and n = n2 + 32 n1:
        //     cc[n] = (1/N) [ R[0] + (-1)^n R[512] + 2 Re sum_{k1 < 32} e^(2 pi i k1 n2 / 1024) e^(2 pi i k1 n1 / 32) T[k1][n2] ],
        //     T[k1][n2] = sum_{k2 < 16} R[k1 + 32 k2] e^(2 pi i k2 n2 / 32)      (bins 1 .. 511 and, at half weight, bin 0)
        // T is ONE matrix-core step per real product (K = 16 = the k2 of the kept half-spectrum): the A operand is R exactly where the
        // forward transform left the spectra — lane (k1, g) holds k2 = mfma_row(j, g), j < 8, and the constant B operand is built in that
        // order —, hi / lo f16 terms as everywhere (R has unit modulus: scale 2^14).  The lags kept are n1 = 0 (lag n2) and n1 = 31 (lag
        // n2 - 32): the outer sum is 16 complex multiply-adds per lane on the accumulators it holds (k1 = mfma_row(r, g)), the two halves
        // of the wave meet in one v_permlane32_swap, and lane l ends with the lag of output row (l + n_mels / 2) mod 64.
        // (Before: three packed 1 024-point inverse FFTs through LDS per frame: 0.077 ms per clip against the foa form's 0.022.)
        const h16x8* gf = reinterpret_cast<const h16x8*>(gtab);                       // [(re | im | -im) * 2 + (hi | lo)][lane]
        const float4* go = reinterpret_cast<const float4*>(gtab + 6 * 1024);         // [r][lane]: {Re c0, -Im c0, Re c31, -Im c31} x 2 / N x 2^-24
        const h16x8 frh = gf[0 * 64 + lane], frl = gf[1 * 64 + lane], fih = gf[2 * 64 + lane], fil = gf[3 * 64 + lane];
        const h16x8 fnh = gf[4 * 64 + lane], fnl = gf[5 * 64 + lane];
        const int half = n_mels >> 1;
        // lower lanes: lag n2 -> row n2 + half (valid for n2 < half); upper lanes: lag n2 - 32 -> row n2 - (32 - half) (valid for n2 >= 32 - half)
        const int orow = lane < 32 ? lane + half : (lane - 32) - (32 - half);
        const bool ook = lane < 32 ? lane < half : orow >= 0;
        const float sgn = (lane & 1) ? -1.f / (float)N : 1.f / (float)N;             // (-1)^n2 / N
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {                           // pair order (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
            const int ca = pr < 3 ? 0 : (pr < 5 ? 1 : 2), cb = pr == 0 ? 1 : ((pr == 1 || pr == 3) ? 2 : 3);
            unsigned rrh[4], rrl[4], rih[4], ril[4];
            float r512 = 1.f;
#pragma unroll
            for (int i = 0; i < NBI; i += 2) {
                float pr_[2], pi_[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int ii = i + h < NBI ? i + h : 0;
                    const float2 xa = ca == 0 ? X0[ii] : XS[ca - 1][ii], xb = XS[cb - 1][ii];
                    const float rr = xa.x * xb.x + xa.y * xb.y, ri = xa.x * xb.y - xa.y * xb.x;
                    const float m2 = rr * rr + ri * ri, inv = __builtin_amdgcn_rsqf(m2) * 16384.f;
                    pr_[h] = m2 > 0.f ? rr * inv : 16384.f;        // angle(0) = 0
                    pi_[h] = m2 > 0.f ? ri * inv : 0.f;
                    if (i + h == 0 && lane == 0) { pr_[h] = rr < 0.f ? -8192.f : 8192.f; pi_[h] = 0.f; }      // bin 0: real, half weight
                    if (i + h == 8) r512 = rr < 0.f ? -1.f : 1.f;                                              // bin 512: real, added below
                }
                if (i < 8) {
                    dft_split_pair(pr_[0], pr_[1], rrh[i >> 1], rrl[i >> 1]);
                    dft_split_pair(pi_[0], pi_[1], rih[i >> 1], ril[i >> 1]);
                }
            }
            const h16x8 arh = dft_frag(rrh), arl = dft_frag(rrl), aih = dft_frag(rih), ail = dft_frag(ril);
            f32x16 tr = zero16(), ti = zero16();
            tr = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, frh, tr, 0, 0, 0);
            ti = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, fih, ti, 0, 0, 0);
            tr = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, fnh, tr, 0, 0, 0);
            ti = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, frh, ti, 0, 0, 0);
            tr = __builtin_amdgcn_mfma_f32_32x32x16_f16(arl, frh, tr, 0, 0, 0);
            ti = __builtin_amdgcn_mfma_f32_32x32x16_f16(arl, fih, ti, 0, 0, 0);
            tr = __builtin_amdgcn_mfma_f32_32x32x16_f16(ail, fnh, tr, 0, 0, 0);
            ti = __builtin_amdgcn_mfma_f32_32x32x16_f16(ail, frh, ti, 0, 0, 0);
            tr = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, frl, tr, 0, 0, 0);
            ti = __builtin_amdgcn_mfma_f32_32x32x16_f16(arh, fil, ti, 0, 0, 0);
            tr = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, fnl, tr, 0, 0, 0);
            ti = __builtin_amdgcn_mfma_f32_32x32x16_f16(aih, frl, ti, 0, 0, 0);
            float p0 = 0.f, p31 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 cc = go[r * 64 + lane];
                p0 = fmaf(tr[r], cc.x, fmaf(ti[r], cc.y, p0));
                p31 = fmaf(tr[r], cc.z, fmaf(ti[r], cc.w, p31));
            }
            // lower lanes keep lag n2 (n1 = 0), upper lanes lag n2 - 32 (n1 = 31): [p0 low | p31 low] + [p0 high | p31 high]
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p31), false, false);
            const float cc = __uint_as_float(sw[0]) + __uint_as_float(sw[1]) + sgn * r512;
            if (ook) stage[orow * C_OUT + 4 + pr] = cc;
        }
        WAVE_LDS_FENCE();
        }
        FEAT_TR(7)
        for (int j = lane; j < n_mels * C_OUT / 4; j += 64) reinterpret_cast<float4*>(frame_out)[j] = reinterpret_cast<const float4*>(stage)[j];
        WAVE_LDS_FENCE();
        FEAT_TR(8)
    }
    lmax = wave_max_dpp(lmax);
    lmin = -wave_max_dpp(-lmin);           // the clip's minimum too: when it is above max - top_db the clamp pass has nothing to do
    __syncthreads();
    if (lane_id == 0) { red[wave] = lmax; red[16 + wave] = lmin; }
    __syncthreads();
    if (tid == 0) {
        float m = -INFINITY, n = INFINITY;
        for (int w = 0; w < WAVES; ++w) { m = fmaxf(m, red[w]); n = fminf(n, red[16 + w]); }
        gmax[blockIdx.x] = m;
        gmin[blockIdx.x] = n;
    }
}

// top_db clamp: x_db = max(x_db, max over the clip - top_db) on the four dB channels.  Every workgroup first reduces the
// per-workgroup maxima the extraction kernel left in gmax[0 .. nparts), then walks its share of the [T * n_mels] rows.
// gmin (feat_dft_kernel's per-workgroup minima; nullptr after the other extraction kernels): a clip whose smallest dB value is
// already at or above the floor is left alone without being read again (every clip of the bench's noise input; quiet passages of a
// real recording do reach the floor, and then the pass runs).
__global__ __launch_bounds__(256) void feat_topdb_kernel(float* __restrict__ out, const float* __restrict__ gmax, const float* __restrict__ gmin,
                                                         int nparts, int64_t n_tm, int c_out, float top_db) {
    __shared__ float red[256], redn[256];
    float m = -INFINITY, n = gmin ? INFINITY : -INFINITY;
    gmax += (size_t)blockIdx.y * nparts;          // blockIdx.y = clip of a batch
    out += (size_t)blockIdx.y * n_tm * c_out;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, gmax[i]);
    if (gmin) for (int i = threadIdx.x; i < nparts; i += 256) n = fminf(n, gmin[(size_t)blockIdx.y * nparts + i]);
    red[threadIdx.x] = m;
    redn[threadIdx.x] = n;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) {
            red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + w]);
            redn[threadIdx.x] = fminf(redn[threadIdx.x], redn[threadIdx.x + w]);
        }
        __syncthreads();
    }
    const float floor_db = red[0] - top_db;
    if (redn[0] >= floor_db) return;               // uniform over the workgroup; NaNs compare false and take the pass
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_tm * 4; i += (int64_t)gridDim.x * 256) {
        float* p = out + (i >> 2) * c_out + (int)(i & 3);
        if (*p < floor_db) *p = floor_db;          // a read-only pass unless a value is actually below the floor (-1 us per clip)
    }
}

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


// float -> IEEE half bits, round to nearest even (host side of feat_dft_kernel's constant fragments)
static unsigned short feat_f2h(float x) {
    unsigned u; memcpy(&u, &x, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    int e = (int)((u >> 23) & 0xff) - 127 + 15;
    unsigned m = u & 0x7fffffu;
    if (((u >> 23) & 0xff) == 0xff) return (unsigned short)(sign | 0x7c00u | (m ? 0x200u : 0u));
    if (e >= 31) return (unsigned short)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (unsigned short)sign;
        m |= 0x800000u;
        const int sh = 14 - e;                       // 24-bit significand -> 10 bits, shifted by 1 - e more
        unsigned r = m >> sh;
        const unsigned rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1u))) ++r;
        return (unsigned short)(sign | r);
    }
    unsigned r = ((unsigned)e << 10) | (m >> 13);
    const unsigned rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;       // a carry into the exponent is the right answer
    return (unsigned short)(sign | r);
}
static float feat_h2f(unsigned short h) {
    const unsigned sign = (unsigned)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const unsigned m = h & 0x3ffu;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 0x400u), e - 25);
    unsigned u; memcpy(&u, &v, 4); u |= sign; memcpy(&v, &u, 4);
    return v;
}
// hi + lo f16 split of v: hi = round(v), lo = round(v - hi)
static void feat_split_h(double v, unsigned short& hi, unsigned short& lo) {
    hi = feat_f2h((float)v);
    lo = feat_f2h((float)(v - (double)feat_h2f(hi)));
}

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
    std::vector<int> mstart4(n_mels), mcnt4(n_mels), moff4(n_mels);
    std::vector<float> mw4;
    int maxc4 = 0;
    for (int m = 0; m < n_mels; ++m) {
        const int lead = mstart[m] & 3;
        mstart4[m] = mstart[m] - lead;
        mcnt4[m] = mcount[m] ? (lead + mcount[m] + 3) / 4 : 0;
        moff4[m] = (int)mw4.size();
        for (int b = 0; b < 4 * mcnt4[m]; ++b) mw4.push_back(b >= lead && b < lead + mcount[m] ? mw[moff[m] + b - lead] : 0.f);
        if (mcnt4[m] > maxc4) maxc4 = mcnt4[m];
    }
    if (mw4.empty()) mw4.resize(4, 0.f);
    std::vector<int> trips(16, 0);
    for (int idx = 0; idx < 4 * n_mels && idx < 512; ++idx) trips[idx >> 6] = std::max(trips[idx >> 6], mcnt4[idx >> 2]);
    for (int idx = 0; idx < 3 * n_mels && idx < 384; ++idx) trips[8 + (idx >> 6)] = std::max(trips[8 + (idx >> 6)], mcnt4[idx / 3]);
    bool ok = true;
    ok &= hipMalloc(&f->win, n_fft * sizeof(float)) == hipSuccess;
    ok &= hipMalloc(&f->tw, (n_fft / 2) * sizeof(float2)) == hipSuccess;
    ok &= hipMalloc(&f->mel_start, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_count, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_off, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_w, mw.size() * sizeof(float)) == hipSuccess;
    ok &= hipMalloc(&f->gmax, 2 * FEAT_MAX_PARTS * sizeof(float)) == hipSuccess;      // maxima | minima
    ok &= hipMalloc(&f->mel_start4, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_cnt4, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_off4, n_mels * sizeof(int)) == hipSuccess;
    ok &= hipMalloc(&f->mel_w4, mw4.size() * sizeof(float)) == hipSuccess;
    ok &= hipMalloc(&f->trips, 16 * sizeof(int)) == hipSuccess;
    if (!ok) { seld_feat_destroy(f); return ffail(nullptr, SELD_ERR_NOMEM, "hipMalloc failed"); }
    hipMemcpy(f->win, win.data(), n_fft * sizeof(float), hipMemcpyHostToDevice);
    hipMemcpy(f->tw, tw.data(), (n_fft / 2) * sizeof(float2), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_start, mstart.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_count, mcount.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_off, moff.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_w, mw.data(), mw.size() * sizeof(float), hipMemcpyHostToDevice);
    f->n_melw = (int)mw.size();
    hipMemcpy(f->mel_start4, mstart4.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_cnt4, mcnt4.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_off4, moff4.data(), n_mels * sizeof(int), hipMemcpyHostToDevice);
    hipMemcpy(f->mel_w4, mw4.data(), mw4.size() * sizeof(float), hipMemcpyHostToDevice);
    if (n_fft == 1024) {
        // feat_dft_kernel's constants, in the lane order of the MFMA operands (see the kernel's header comment)
        std::vector<unsigned short> tab((DFT_TAB_BYTES + (mode == 1 ? GCC_TAB_BYTES : 0)) / 2, 0);
        auto mrow = [](int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; };
        for (int sidx = 0; sidx < 2; ++sidx)
            for (int lane = 0; lane < 64; ++lane) {
                const int li = lane & 31, g = lane >> 5;
                for (int j = 0; j < 8; ++j) {
                    // step 1, B operand: F[n1][k1 = li] = exp(-2 pi i n1 k1 / 32) * 2^10, n1 = 16 s + 8 g + j
                    const int n1 = 16 * sidx + 8 * g + j;
                    const double ang = -2.0 * PI * (double)((n1 * li) & 31) / 32.0;
                    unsigned short h, l;
                    feat_split_h(cos(ang) * 1024.0, h, l);
                    tab[((((sidx * 2 + 0) * 2 + 0) * 64 + lane) * 8) + j] = h; tab[((((sidx * 2 + 0) * 2 + 1) * 64 + lane) * 8) + j] = l;
                    feat_split_h(sin(ang) * 1024.0, h, l);
                    tab[((((sidx * 2 + 1) * 2 + 0) * 64 + lane) * 8) + j] = h; tab[((((sidx * 2 + 1) * 2 + 1) * 64 + lane) * 8) + j] = l;
                    // step 3, A operand: row rho = li: rho < 16 -> Re W32^(n2 k2), else Im, k2 = rho & 15, n2 = mfma_row(8 s + j, g)
                    const int n2 = mrow(8 * sidx + j, g), k2 = li & 15;
                    const double a2 = -2.0 * PI * (double)((n2 * k2) & 31) / 32.0;
                    feat_split_h((li < 16 ? cos(a2) : sin(a2)) * 1024.0, h, l);
                    tab[4096 + (((sidx * 2 + 0) * 64 + lane) * 8) + j] = h; tab[4096 + (((sidx * 2 + 1) * 64 + lane) * 8) + j] = l;
                }
            }
        float* tw2 = reinterpret_cast<float*>(tab.data() + 6144);            // byte offset 12 KB
        for (int r = 0; r < 16; ++r)
            for (int lane = 0; lane < 64; ++lane) {
                const int n2 = mrow(r, lane >> 5), k1 = lane & 31;
                const double ang = -2.0 * PI * (double)(n2 * k1) / 1024.0;
                tw2[(r * 64 + lane) * 2] = (float)(cos(ang) / 32768.0);
                tw2[(r * 64 + lane) * 2 + 1] = (float)(sin(ang) / 32768.0);
            }
        if (mode == 1) {
            // mic: the pruned inverse transform of GCC-PHAT (see the kernel).  B operand of T = R F: column n2 = lane & 31, K slot (g, j) = k2 =
            // mfma_row(j, g): F = exp(+2 pi i k2 n2 / 32) x 2^10 as re | im | -im, hi | lo; then the outer stage's constants per accumulator
            // register r and lane: k1 = mfma_row(r, g), c0 = exp(2 pi i k1 n2 / 1024), c31 = c0 exp(-2 pi i k1 / 32), as {Re, -Im} x 2 / N x 2^-24
            unsigned short* gt = tab.data() + DFT_TAB_BYTES / 2;
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int k2 = mrow(j, lane >> 5), n2 = lane & 31;
                    const double ang = 2.0 * PI * (double)((k2 * n2) & 31) / 32.0;
                    const double v[3] = {cos(ang) * 1024.0, sin(ang) * 1024.0, -sin(ang) * 1024.0};
                    for (int q = 0; q < 3; ++q) {
                        unsigned short h, l;
                        feat_split_h(v[q], h, l);
                        gt[((q * 2 + 0) * 64 + lane) * 8 + j] = h; gt[((q * 2 + 1) * 64 + lane) * 8 + j] = l;
                    }
                }
            float* go = reinterpret_cast<float*>(gt + 6 * 512);
            const double sc = 2.0 / 1024.0 / 16777216.0;
            for (int r = 0; r < 16; ++r)
                for (int lane = 0; lane < 64; ++lane) {
                    const int k1 = mrow(r, lane >> 5), n2 = lane & 31;
                    const double a0 = 2.0 * PI * (double)(k1 * n2) / 1024.0, a31 = a0 - 2.0 * PI * (double)k1 / 32.0;
                    float* e = go + (r * 64 + lane) * 4;
                    e[0] = (float)(cos(a0) * sc); e[1] = (float)(-sin(a0) * sc); e[2] = (float)(cos(a31) * sc); e[3] = (float)(-sin(a31) * sc);
                }
        }
        if (hipMalloc(&f->dft_tab, tab.size() * 2) != hipSuccess) { seld_feat_destroy(f); return ffail(nullptr, SELD_ERR_NOMEM, "hipMalloc failed"); }
        hipMemcpy(f->dft_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice);
    }
    if (n_fft == 1024 && n_mels <= 64) {
        // the mel projection as 4 x 4 x 4 blocks (see feat_dft_kernel's mel_round): group g = mels 4 g .. 4 g + 3 runs over bins
        // [gs, ge), gs a multiple of 4, cut into slots of four steps of four bins; slot v goes to block v & 15 of sequence v >> 4
        struct Slot { int g, bin; };
        std::vector<Slot> slots;
        for (int g = 0; g < 16; ++g) {
            int gs = 1 << 30, ge = 0;
            for (int m = 4 * g; m < 4 * g + 4 && m < n_mels; ++m)
                if (mcount[m]) { gs = std::min(gs, mstart[m] & ~3); ge = std::max(ge, mstart[m] + mcount[m]); }
            for (int b0 = gs; b0 < ge; b0 += 16) slots.push_back({g, std::min(b0, 500)});      // sixteen bins from b0; the planes end at bin 515
        }
        const int n_seq = std::max(1, ((int)slots.size() + 15) / 16);
        if (n_seq <= 4 && slots.size() * 64 <= 4 * 516 * 4) {     // (the slots' sums use the planes' space)
            std::vector<unsigned> mmv((size_t)(n_seq + 1) * 64 * 4, 0u);
            const unsigned zero_w = (unsigned)mmv.size() / 4;
            mmv.resize(mmv.size() + 4 * 4, 0u);                      // sixteen zero weights: the idle blocks'
            for (int q = 0; q < n_seq; ++q)
                for (int lane = 0; lane < 64; ++lane) { unsigned* e = &mmv[(q * 64 + lane) * 4]; e[0] = 0; e[1] = zero_w; }
            int prev_g = -1, prev_end = 0;
            for (size_t v = 0; v < slots.size(); ++v) {
                const Slot& sl = slots[v];
                const int from = (sl.g == prev_g) ? std::max(sl.bin, prev_end) : sl.bin;      // (a slot moved down to bin 500 must not count bins twice)
                prev_g = sl.g; prev_end = sl.bin + 16;
                const unsigned base = (unsigned)mmv.size() / 4;
                for (int n = 0; n < 4; ++n)
                    for (int k = 0; k < 16; ++k) {
                        const int m = 4 * sl.g + n, bin = sl.bin + k;
                        const float w = (m < n_mels && bin >= from && bin >= mstart[m] && bin < mstart[m] + mcount[m]) ? mw[moff[m] + bin - mstart[m]] : 0.f;
                        unsigned u; memcpy(&u, &w, 4);
                        mmv.push_back(u);
                    }
                for (int n = 0; n < 4; ++n) {
                    unsigned* e = &mmv[(((v >> 4) * 64) + (v & 15) * 4 + n) * 4];
                    e[0] = (unsigned)sl.bin; e[1] = base + 4u * (unsigned)n;
                }
            }
            int pmax = 1;
            for (int g = 0; g < 16; ++g) {
                int first = -1, cnt = 0;
                for (size_t v = 0; v < slots.size(); ++v) if (slots[v].g == g) { if (first < 0) first = (int)v; ++cnt; }
                pmax = std::max(pmax, cnt);
                for (int n = 0; n < 4; ++n) {
                    unsigned* e = &mmv[((size_t)n_seq * 64 + 4 * g + n) * 4];
                    e[0] = (unsigned)(4 * std::max(first, 0) + n); e[1] = (unsigned)std::max(cnt, 1);      // (a group without mels: any slot, never stored)
                }
            }
            if (pmax <= 12) {                                        // (the gather loop's bound)
                f->n_mel_mm = (int)mmv.size() / 4;
                f->mel_mm_seq = n_seq;
                f->mel_mm_pmax = pmax;
                if (hipMalloc(&f->mel_mm, mmv.size() * sizeof(unsigned)) != hipSuccess) { seld_feat_destroy(f); return ffail(nullptr, SELD_ERR_NOMEM, "hipMalloc failed"); }
                hipMemcpy(f->mel_mm, mmv.data(), mmv.size() * sizeof(unsigned), hipMemcpyHostToDevice);
            }
        }
    }
    f->n_melw4 = (int)mw4.size();
    hipMemcpy(f->trips, trips.data(), 16 * sizeof(int), hipMemcpyHostToDevice);
    f->maxc4 = maxc4;
    *out = f;
    return SELD_OK;
}

void seld_feat_destroy(seld_feat* f) {
    if (!f) return;
    hipSetDevice(f->device);
    hipDeviceSynchronize();
    hipFree(f->win); hipFree(f->tw); hipFree(f->mel_start); hipFree(f->mel_count); hipFree(f->mel_off); hipFree(f->mel_w);
    hipFree(f->gmax); hipFree(f->dft_tab); hipFree(f->mel_mm);
    hipFree(f->mel_start4); hipFree(f->mel_cnt4); hipFree(f->mel_off4); hipFree(f->mel_w4); hipFree(f->trips);
    delete f;
}

int seld_feat_set_option(seld_feat* f, const char* key, int value) {
    if (!f || !key) return SELD_ERR_INVALID;
    if (!strcmp(key, "wave_kernel")) { f->use_wave_kernel = value != 0; return SELD_OK; }
    if (!strcmp(key, "dft")) { f->use_dft = value != 0; return SELD_OK; }
    if (!strcmp(key, "dbg")) { f->dbg = value; return SELD_OK; }
    return ffail(f, SELD_ERR_INVALID, std::string("unknown option: ") + key);
}

int64_t seld_feat_frames(const seld_feat* f, int64_t n_samples) { return f ? 1 + n_samples / f->hop : -1; }
int seld_feat_channels(const seld_feat* f) { return f ? (f->mode == 0 ? 7 : 10) : -1; }

int seld_feat_extract_batch(seld_feat* f, const float* wav, int n_clips, int n_ch, int64_t n_samples, float* out, void* stream) {
    if (!f || !wav || !out || n_clips < 1) return SELD_ERR_INVALID;
    if (n_clips > f->gmax_clips) {          // per-clip maxima: [clips][FEAT_MAX_PARTS]
        hipSetDevice(f->device);
        float* g = nullptr;
        if (hipMalloc(&g, (size_t)2 * n_clips * FEAT_MAX_PARTS * sizeof(float)) != hipSuccess) return ffail(f, SELD_ERR_NOMEM, "feat_extract_batch: maxima buffer");
        hipStreamSynchronize((hipStream_t)stream);
        hipFree(f->gmax);
        f->gmax = g; f->gmax_clips = n_clips;
    }
    if (n_ch != 4) return ffail(f, SELD_ERR_UNSUPPORTED, "feature stage is built for 4-channel (FOA / MIC) audio");
    if (n_samples <= f->n_fft / 2) return ffail(f, SELD_ERR_INVALID, "clip shorter than the reflect padding (n_fft/2)");
    hipStream_t st = (hipStream_t)stream;
    const int N = f->n_fft, NBP = N / 2 + 1 + 3;
    const int64_t T = 1 + n_samples / f->hop;
    int nparts = 0;
    bool launched = false, have_min = false;
    // wave-per-frame kernel: n_fft 256 .. 1024 (2048 spills registers: the workgroup kernel serves it), n_mels <= 128
    // n_fft 1024, n_mels <= 64: the transform(s) AND the mel projection on the matrix cores (feat_dft_kernel; foa 12 waves, mic 8)
    const int dft_waves = DFT_WAVES_OF(f->mode), dft_cout = f->mode == 0 ? 7 : 10;
    const size_t dft_tables = (size_t)N * sizeof(float) + 32 * sizeof(float) + (size_t)f->n_mel_mm * 16 + DFT_TAB_BYTES + (f->mode == 1 ? GCC_TAB_BYTES : 0);
    const size_t dft_wb = (((size_t)4 * (N / 2 + 4) + (size_t)f->n_mels * dft_cout + 8) * sizeof(float) + 15) & ~(size_t)15;
    if (f->dft_tab && f->mel_mm && f->use_dft && f->use_wave_kernel && (f->n_mels & 3) == 0 && n_samples < (int64_t)1 << 30 &&
        dft_tables + dft_waves * dft_wb <= 160 * 1024) {
        const size_t tables = dft_tables;
        const size_t wb = dft_wb;                                   // the four per-bin planes of a wave + its staged output frame
        const size_t smem = tables + dft_waves * wb;
        // persistent workgroups: one per CU over the whole batch (a workgroup loads its tables once; a wave that walks several frames
        // has the next frame's first channel in flight while it finishes the current one)
        int64_t blocks = (T + dft_waves - 1) / dft_waves;
        const int64_t per_clip = std::max<int64_t>(1, (256 + n_clips - 1) / n_clips);
        if (blocks > per_clip) blocks = per_clip;
#define FEAT_DFT_CASE(Q_, M_)                                                                                                   \
        {                                                                                                                       \
            hipFuncSetAttribute(reinterpret_cast<const void*>(feat_dft_kernel<Q_, M_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);   \
            hipLaunchKernelGGL((feat_dft_kernel<Q_, M_>), dim3((unsigned)blocks, (unsigned)n_clips), dim3(64 * DFT_WAVES_OF(M_)), smem, st, wav, n_samples, T, f->hop, \
                               f->n_mels, f->win, f->dft_tab, f->mel_mm, f->n_mel_mm, f->mel_mm_pmax, out, f->gmax, f->gmax + (size_t)f->gmax_clips * FEAT_MAX_PARTS, (int)wb); \
        }
        if (f->mode == 0) {
            if (f->mel_mm_seq == 1) FEAT_DFT_CASE(1, 0) else if (f->mel_mm_seq == 2) FEAT_DFT_CASE(2, 0) else if (f->mel_mm_seq == 3) FEAT_DFT_CASE(3, 0) else FEAT_DFT_CASE(4, 0)
        } else {
            if (f->mel_mm_seq == 1) FEAT_DFT_CASE(1, 1) else if (f->mel_mm_seq == 2) FEAT_DFT_CASE(2, 1) else if (f->mel_mm_seq == 3) FEAT_DFT_CASE(3, 1) else FEAT_DFT_CASE(4, 1)
        }
#undef FEAT_DFT_CASE
        launched = true; nparts = (int)blocks; have_min = true;
    }
    if (!launched && f->logn >= 8 && f->logn <= 10 && f->use_wave_kernel && f->n_mels <= 128 && f->n_mels <= N) {
        const size_t tables = (size_t)(N / 2) * sizeof(float2) + (size_t)N * sizeof(float) + (size_t)f->n_melw4 * sizeof(float) +
                              (size_t)(2 * f->n_mels + ((f->n_mels + 3) & ~3) + 16) * sizeof(int);
        const int fw = FEAT_WAVES_OF(f->mode, f->logn);
        const size_t wb = feat_wave_bytes(N, f->n_mels, f->mode == 0 ? 7 : 10), smem = tables + fw * wb;
        int64_t blocks = (T + fw - 1) / fw;
        if (blocks > 1024) blocks = 1024;
#define FEAT_WAVE_CASE(MODE_, LOGN_)                                                                                            \
        {                                                                                                                       \
            hipFuncSetAttribute(reinterpret_cast<const void*>(feat_wave_kernel<MODE_, LOGN_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            hipLaunchKernelGGL((feat_wave_kernel<MODE_, LOGN_>), dim3((unsigned)blocks, (unsigned)n_clips), dim3(64 * FEAT_WAVES_OF(MODE_, LOGN_)), smem, st, wav, n_samples, T, \
                               f->hop, f->n_mels, f->n_melw4, f->maxc4, f->win, f->tw, f->mel_start4, f->mel_cnt4, f->mel_off4, f->mel_w4, \
                               out, f->gmax, (int)wb, f->dbg, f->trips);                                                                      \
            launched = true; nparts = (int)blocks;                                                                              \
        }
        if (f->mode == 0) {
            if (f->logn == 8) FEAT_WAVE_CASE(0, 8) else if (f->logn == 9) FEAT_WAVE_CASE(0, 9) else FEAT_WAVE_CASE(0, 10)
        } else {
            if (f->logn == 8) FEAT_WAVE_CASE(1, 8) else if (f->logn == 9) FEAT_WAVE_CASE(1, 9) else FEAT_WAVE_CASE(1, 10)
        }
#undef FEAT_WAVE_CASE
    }
    if (!launched) {
    nparts = (int)(T < FEAT_MAX_PARTS ? T : FEAT_MAX_PARTS);
    const size_t vals = f->mode == 0 ? (size_t)7 * NBP : (size_t)4 * NBP + (size_t)12 * NBP;
    const size_t smem = (size_t)(4 * N) * sizeof(float2) + (size_t)(N / 2) * sizeof(float2) + (vals + 256) * sizeof(float);
    if (f->mode == 0) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(feat_frame_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(feat_frame_kernel<0>, dim3((unsigned)nparts, (unsigned)n_clips), dim3(256), smem, st, wav, n_samples, T, N, f->logn, f->hop, f->n_mels,
                           f->win, f->tw, f->mel_start, f->mel_count, f->mel_off, f->mel_w, out, f->gmax);
    } else {
        hipFuncSetAttribute(reinterpret_cast<const void*>(feat_frame_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(feat_frame_kernel<1>, dim3((unsigned)nparts, (unsigned)n_clips), dim3(256), smem, st, wav, n_samples, T, N, f->logn, f->hop, f->n_mels,
                           f->win, f->tw, f->mel_start, f->mel_count, f->mel_off, f->mel_w, out, f->gmax);
    }
    }
    const int64_t n_tm = T * f->n_mels;
    int64_t tb = (n_tm * 4 + 255) / 256;
    if (tb > std::max<int64_t>(32, 2048 / n_clips)) tb = std::max<int64_t>(32, 2048 / n_clips);     // (mostly an early exit: keep the grid small)
    hipLaunchKernelGGL(feat_topdb_kernel, dim3((unsigned)tb, (unsigned)n_clips), dim3(256), 0, st, out, f->gmax, have_min ? f->gmax + (size_t)f->gmax_clips * FEAT_MAX_PARTS : nullptr, nparts, n_tm, f->mode == 0 ? 7 : 10, 80.f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ffail(f, SELD_ERR_HIP, std::string("feat_extract: ") + hipGetErrorString(e));
    return SELD_OK;
}

int seld_feat_extract(seld_feat* f, const float* wav, int n_ch, int64_t n_samples, float* out, void* stream) {
    return seld_feat_extract_batch(f, wav, 1, n_ch, n_samples, out, stream);
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

#ifdef FEAT_TRACE
extern "C" int seld_feat_trace_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_feat_trace), sizeof(unsigned long long) * DFT_WAVES * 2 * 10) == hipSuccess ? 0 : 1;
}
#endif
