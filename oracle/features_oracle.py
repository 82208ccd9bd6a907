"""CPU oracle for the feature stage (STFT -> log-mel + FOA intensity vectors | GCC-PHAT).

TEST INFRASTRUCTURE ONLY (see oracle/seldnet_oracle.py header).  PARITY UNPINNED for numerics: the
reference's arithmetic lives in un-vendored torchaudio 0.8-0.10 (requirements.txt:1-8; `complex_norm`
pins the range), absent from the build container.  This file restates the published torchaudio
algorithms at the reference's call sites:

  feature_extractor.py:153-173  complex_spec -> torchaudio.functional.spectrogram(pad=0, hann(win_length)
                                periodic, n_fft, hop, win_length, power=None, normalized) with torchaudio
                                defaults center=True, pad_mode='reflect', onesided=True  (= torch.stft)
  feature_extractor.py:59-71    complex_norm(power=2) -> MelScale(n_mels, sample_rate) [create_fb_matrix:
                                HTK mel, f_min 0, f_max sr//2, norm None, all_freqs = linspace(0, sr//2, n_freqs)]
                                -> amplitude_to_DB(10, amin 1e-10, db_multiplier 0, top_db 80) [max over the
                                whole [chan,freq,time] tensor in 0.8 and, for a 3-D input, in 0.9/0.10 too]
  feature_extractor.py:176-193  foa_intensity_vectors (eps 1e-8), then melscale WITHOUT dB (:75-77)
  feature_extractor.py:196-214  gcc_features: irfft(exp(j*angle(conj(Xm)*Xn)), dim=freq), keep [-n_mels/2:] + [:n_mels/2]
  feature_extractor.py:84-88    cat -> [chan,freq,time] -> transpose(0,2) -> [time,freq,chan]
  feature_extractor.py:218-234  calculate_statistics / apply_normalizer

Pinned by the reference's own smoke test (feature_extractor_test.py:24-34: zeros[4,32000] @16 kHz ->
ndim 3, 7 | 10 channels) and cross-checked against scipy.signal / a direct DFT in tests/test_features_cpu.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def hann_window(win_length: int, dtype=torch.float64):
    return torch.hann_window(win_length, periodic=True, dtype=dtype)


def complex_spec(wav, pad=0, n_fft=512, win_length=None, hop_length=None, normalized=False):
    """-> complex [chan, n_fft//2+1, 1 + n//hop]"""
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = win_length // 2
    wav = torch.as_tensor(wav)
    if pad > 0:
        wav = torch.nn.functional.pad(wav, (pad, pad))
    win = hann_window(win_length, wav.dtype)
    spec = torch.stft(wav, n_fft, hop_length, win_length, win, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    if normalized:
        spec = spec / win.pow(2.0).sum().sqrt()
    return spec


def mel_filterbank(n_freqs: int, n_mels: int, sample_rate: int, f_min: float = 0.0, dtype=np.float64):
    """torchaudio.functional.create_fb_matrix(n_freqs, f_min, f_max=sr//2, n_mels, sample_rate, norm=None) -> [n_freqs, n_mels]"""
    f_max = float(sample_rate // 2)
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(dtype)


def amplitude_to_db(x, multiplier=10.0, amin=1e-10, db_multiplier=0.0, top_db=80.0):
    x_db = multiplier * torch.log10(torch.clamp(x, min=amin)) - multiplier * db_multiplier
    if top_db is not None:
        x_db = torch.clamp(x_db, min=float(x_db.max()) - top_db)
    return x_db


def foa_intensity_vectors(spec, eps=1e-8):
    w = torch.conj(spec[0])
    iv = torch.stack([torch.real(w * spec[3]), torch.real(w * spec[1]), torch.real(w * spec[2])], 0)
    norm = torch.sqrt((iv ** 2).sum(0))
    return iv / torch.maximum(norm, torch.full_like(norm, eps))


def gcc_features(spec, n_mels: int):
    n_chan = spec.shape[0]
    out = []
    for m in range(n_chan):
        for n in range(m + 1, n_chan):
            R = torch.conj(spec[m]) * spec[n]
            cc = torch.fft.irfft(torch.exp(1j * torch.angle(R)), dim=0)
            out.append(torch.cat([cc[-n_mels // 2:], cc[:(n_mels + 1) // 2]], 0))
    return torch.stack(out, 0)


def gcc_pruned_inverse_model(R: np.ndarray, n_lags: int = 64) -> np.ndarray:
    """Index model of the inverse transform features.hip's mic kernel uses for GCC-PHAT (n_fft 1024): irfft of the half spectrum R[0..512]
    evaluated ONLY at the n_lags kept lags, as two stages of 32-point sums.  With k = k1 + 32 k2 (k2 < 16: bins 0 .. 511) and n = n2 + 32 n1,
        T[k1][n2] = sum_k2 R[k1 + 32 k2] exp(2 pi i k2 n2 / 32)                 (bin 0 at half weight; one matrix-core step per real product)
        cc[n]     = (1/N) [ 2 Re sum_k1 exp(2 pi i k1 n2 / 1024) exp(2 pi i k1 n1 / 32) T[k1][n2] + (-1)^n Re R[512] ]
    and only n1 = 0 (lags 0 .. 31) and n1 = 31 (lags -32 .. -1) are formed.  Returns cc[-n_lags/2 :] || cc[: n_lags/2] like gcc_features.
    tests/test_features_cpu.py holds it against numpy's irfft."""
    N = 1024
    R = np.asarray(R, np.complex128).copy()
    assert R.shape == (N // 2 + 1,)
    r512 = R[512].real
    R[0] = 0.5 * R[0].real                                  # irfft ignores the imaginary parts of bins 0 and N/2
    k1, k2, n2 = np.arange(32), np.arange(16), np.arange(32)
    Rm = R[:512].reshape(16, 32).T                          # [k1][k2] = R[k1 + 32 k2]
    T = Rm @ np.exp(2j * np.pi * np.outer(k2, n2) / 32)     # [k1][n2]
    out = {}
    for n1 in (0, 31):
        c = np.exp(2j * np.pi * np.outer(k1, n2) / N) * np.exp(2j * np.pi * k1 * n1 / 32)[:, None]
        out[n1] = (2.0 * (c * T).sum(0).real + (-1.0) ** n2 * r512) / N
    h = n_lags // 2
    return np.concatenate([out[31][32 - h:], out[0][:h]])


def extract_features(wav, sample_rate, mode="foa", n_mels=64, dtype=torch.float32, **kwargs) -> np.ndarray:
    """reference feature_extractor.extract_features (53-88) -> [time, n_mels, 7|10]"""
    wav = torch.as_tensor(np.asarray(wav)).to(dtype)
    spec = complex_spec(wav, **kwargs)
    fb = torch.as_tensor(mel_filterbank(spec.shape[1], n_mels, sample_rate)).to(dtype)
    mel = lambda s: torch.matmul(s.transpose(1, 2), fb).transpose(1, 2)
    feats = [amplitude_to_db(mel(spec.real ** 2 + spec.imag ** 2))]
    if mode == "foa":
        feats.append(mel(foa_intensity_vectors(spec)))
    elif mode == "mic":
        feats.append(gcc_features(spec, n_mels))
    else:
        raise ValueError("invalid mode")
    return torch.cat(feats, 0).permute(2, 1, 0).contiguous().numpy()


def calculate_statistics(features_list):
    f = np.concatenate(features_list, 0)
    return f.mean(axis=0, keepdims=True), f.std(axis=0, keepdims=True)


def apply_normalizer(feature, mean, std, eps=1e-8):
    return (feature - mean) / np.maximum(std, eps)


# ----------------------------------------------------------------------------------------------
# Reference model of the GPU kernel's FFT (radix-2 Stockham autosort, ping-pong buffers): used by
# the CPU tests to validate the index arithmetic the HIP kernel implements.
def stockham_fft(x: np.ndarray) -> np.ndarray:
    n = x.shape[-1]
    logn = int(math.log2(n))
    assert 1 << logn == n
    tw = np.exp(-2j * np.pi * np.arange(n // 2) / n)
    a = x.astype(np.complex128).copy()
    for s in range(logn):
        m = 1 << s
        b = np.empty_like(a)
        j = np.arange(n // 2)
        k = j & (m - 1)
        w = tw[k * (n // (2 * m))]
        u, v = a[..., j], w * a[..., j + n // 2]
        b[..., 2 * j - k] = u + v
        b[..., 2 * j - k + m] = u - v
        a = b
    return a


def stockham_fft_radix4(x: np.ndarray) -> np.ndarray:
    """Index model of the wave-per-frame FFT of features.hip (feat_wave_kernel): radix-4 Stockham autosort passes,
    L = 1, 4, 16, ... (every pass reads ALL its inputs, then writes: in place), one final radix-2 pass when log2 n is odd.
    Pass with sub-transform length L: butterfly j in [0, n/4), k = j mod L, inputs a_q = src[j + q n/4] times
    W_{4L}^{q k} = tw[q k n/(4L)], outputs dst[4 (j - k) + k + p L] = sum_q a_q (-i)^{p q}."""
    n = x.shape[-1]
    logn = int(math.log2(n))
    assert 1 << logn == n and n >= 4
    tw = np.exp(-2j * np.pi * np.arange(n) / n)
    a = x.astype(np.complex128).copy()
    L = 1
    for _ in range(logn // 2):
        b = np.empty_like(a)
        j = np.arange(n // 4)
        k = j & (L - 1)
        v = [a[..., j + q * (n // 4)] * tw[q * k * (n // (4 * L))] for q in range(4)]
        t0, t1, t2 = v[0] + v[2], v[0] - v[2], v[1] + v[3]
        t3 = -1j * (v[1] - v[3])
        o = 4 * (j - k) + k
        b[..., o], b[..., o + L], b[..., o + 2 * L], b[..., o + 3 * L] = t0 + t2, t1 + t3, t0 - t2, t1 - t3
        a = b
        L *= 4
    if logn & 1:
        b = np.empty_like(a)
        j = np.arange(n // 2)
        k = j & (L - 1)
        u, v = a[..., j], tw[k * (n // (2 * L))] * a[..., j + n // 2]
        b[..., 2 * j - k], b[..., 2 * j - k + L] = u + v, u - v
        a = b
    return a
