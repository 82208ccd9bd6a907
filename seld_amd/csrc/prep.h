// prep.h — the per-step weight pre-passes (device bodies), shared by their stand-alone kernels and by the merged launch of
// prep.hip: pre-split bf16 planes of the GEMM weights (gemm_sb.hip), of the 64 -> 64 conv weights incl. the flipped form
// (conv_sb.hip), and the folded head weights W1 W2 (gemm.hip).
#pragma once
#include "common.h"

// B fp32 ([K,N] if !transb, [N,K] if transb) -> planes [chunk c = k/32][plane][n][piece'][8] bf16, piece' = piece ^ ((n >> 2) & 3).
// transb == 2: src is a 3x3 kernel [9][N = Cin][ldb = Cout] (HWIO) and B the matrix of its INPUT-GRADIENT convolution,
// B[(tap', co)][ci] = src[8 - tap'][ci][co] (taps flipped, channels swapped), K = 9 Cout
__device__ __forceinline__ void gemm_split_b_body(const GemmSplitJobs& jobs, int job, int bx, int nbx) {
    const int K = jobs.K[job], N = jobs.N[job], ldb = jobs.ldb[job], transb = jobs.transb[job];
    const float* __restrict__ src = jobs.src[job];
    unsigned short* __restrict__ dst = jobs.dst[job];
    for (int idx = bx * 256 + threadIdx.x; idx < K * N; idx += nbx * 256) {
        // the fast index follows the contiguous axis of the source
        const int k = transb ? idx % K : idx / N, n = transb ? idx / K : idx % N;
        float x;
        if (transb == 2) {
            const int co_n = K / 9, tap = k / co_n, co = k - tap * co_n;
            x = src[((size_t)(8 - tap) * N + n) * ldb + co];
        } else
            x = transb ? src[(size_t)n * ldb + k] : src[(size_t)k * ldb + n];
        // plane 0: the truncated high part — in bf16 single-product mode the value ROUNDED to nearest bf16, which is all a ONE-form
        // consumer reads.  Planes 1, 2 split the residual x - plane0 exactly either way (|residual| <= an ulp of plane 0: 16
        // significant bits, 8 + 8), so a consumer WITHOUT a ONE form (six products over all three planes) computes with the exact
        // weights in both modes — it never reads an unwritten plane.
        const unsigned u = jobs.one ? bf16_rne_bits(x) << 16 : __float_as_uint(x) & 0xffff0000u;
        const float r = x - __uint_as_float(u);
        // a transposed / flipped B is a backward operand: its mid plane is rounded (common.h split2r_pair) for the four-product form; plane 2 completes
        // the split exactly either way
        const unsigned v = __float_as_uint(r) + (transb ? 0x8000u : 0u);
        const float s = r - __uint_as_float(v & 0xffff0000u);
        const int c = k >> 5, kk = k & 31, piece = (kk >> 3) ^ ((n >> 2) & 3);
        unsigned short* o = dst + ((size_t)c * 3 * N + n) * 32 + piece * 8 + (kk & 7);
        o[0] = (unsigned short)(u >> 16);
        o[(size_t)N * 32] = (unsigned short)(v >> 16);
        o[(size_t)2 * N * 32] = (unsigned short)(__float_as_uint(s) >> 16);
    }
}

// w [9][in 64][out 64] fp32 -> planes [9][3][out][in] bf16; flip: the input-gradient form (tap 8 - tap, in / out swapped)
__device__ __forceinline__ void split_weights_body(const SplitWeightJobs& jobs, int job, int bx) {
    const float* __restrict__ w = jobs.w[job];
    unsigned short* __restrict__ wsp = jobs.dst[job];
    const int idx = bx * 256 + threadIdx.x;
    if (idx >= 9 * 4096) return;
    const int tap = idx >> 12, rem = idx & 4095, out = rem >> 6, in = rem & 63;
    const float x = jobs.flip[job] ? w[(8 - tap) * 4096 + out * 64 + in] : w[tap * 4096 + in * 64 + out];
    const unsigned u = jobs.one ? bf16_rne_bits(x) << 16 : __float_as_uint(x) & 0xffff0000u;      // see gemm_split_b_body
    const float r = x - __uint_as_float(u);
    // the flipped (input-gradient) form rounds its mid plane (common.h split2r_pair: the four-product kernel reads planes 0, 1 only and wants a
    // zero-mean remainder); plane 2 still completes the exact split, so the six-product kernel sees the same weights either way
    const unsigned v = __float_as_uint(r) + (jobs.flip[job] ? 0x8000u : 0u);
    const float s = r - __uint_as_float(v & 0xffff0000u);
    unsigned short* o = wsp + (size_t)tap * 3 * 4096 + out * 64 + in;
    o[0] = (unsigned short)(u >> 16);
    o[4096] = (unsigned short)(v >> 16);
    o[2 * 4096] = (unsigned short)(__float_as_uint(s) >> 16);
}

// row k of Weff = [W1s W2s | W1d W2d] (k == K: the bias row b1 W2 + b2); one thread per column
__device__ __forceinline__ void heads_weff_body(const HeadsLin& h, float* __restrict__ weff, int k, int col) {
    const int NT = h.n[0] + h.n[1];
    if (k > h.K || col >= NT) return;
    const int hd = col >= h.n[0], n = col - (hd ? h.n[0] : 0), N = h.n[hd];
    const float* w1 = h.w1[hd];
    const float* w2 = h.w2[hd];
    // four interleaved partial sums (fixed order): the loads of a row do not wait on one dependent FMA chain
    const float* a = k < h.K ? w1 + (size_t)k * h.Hd : h.b1[hd];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = 0;
#pragma unroll 2
    for (; j + 3 < h.Hd; j += 4) {
        s0 += (double)a[j] * (double)w2[(size_t)j * N + n];
        s1 += (double)a[j + 1] * (double)w2[(size_t)(j + 1) * N + n];
        s2 += (double)a[j + 2] * (double)w2[(size_t)(j + 2) * N + n];
        s3 += (double)a[j + 3] * (double)w2[(size_t)(j + 3) * N + n];
    }
    for (; j < h.Hd; ++j) s0 += (double)a[j] * (double)w2[(size_t)j * N + n];
    double s = (s0 + s1) + (s2 + s3);
    if (k == h.K) s += (double)h.b2[hd][n];
    weff[(size_t)k * NT + col] = (float)s;
}
