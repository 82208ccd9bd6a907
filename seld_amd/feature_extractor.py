"""Host-side mirror of the reference's feature_extractor.py for the on-device feature stage.

`extract_features(wav, sample_rate, mode, n_mels, **kwargs)` keeps the reference's signature and
return type (feature_extractor.py:53-57: np.ndarray [time, freq, chan]); the arithmetic runs in
libseld_hip.so (features.hip).  `extract_features_device` returns the device tensor instead, and
`FeatureExtractor` keeps the plan (window, twiddles, sparse mel filterbank) for repeated calls —
the in-loop variant BASELINE.json's config 5 asks for.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_MODES = {"foa": 0, "mic": 1}


class FeatureExtractor:
    def __init__(self, sample_rate: int, mode: str = "foa", n_mels: int = 64, pad: int = 0, n_fft: int = 512,
                 win_length=None, hop_length=None, normalized: bool = False, device: int | None = None):
        if mode not in _MODES:
            raise ValueError("invalid mode")           # feature_extractor.py:81-82
        if pad != 0:
            raise ValueError("pad != 0 has no kernel (the reference never passes it)")
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SeldLibraryError("no HIP device visible: seld_amd has no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        win_length = n_fft if win_length is None else win_length
        hop_length = win_length // 2 if hop_length is None else hop_length
        h = C.c_void_p()
        rc = self.lib.seld_feat_create(int(sample_rate), int(n_fft), int(win_length), int(hop_length), int(n_mels),
                                       _MODES[mode], int(bool(normalized)), self.device, C.byref(h))
        if rc:
            raise ValueError(f"{_lib.ERR_NAMES.get(rc, rc)}: {self.lib.seld_feat_last_error(None).decode()}")
        self.h = h
        self.n_mels, self.hop = n_mels, hop_length
        self.channels = int(self.lib.seld_feat_channels(h))
        self._dev = torch.device("cuda", self.device)

    def set_option(self, key: str, value: int) -> None:
        """kernel selection (seld_feat_set_option): "wave_kernel" 1 (default) | 0"""
        _lib.check(self.lib.seld_feat_set_option(self.h, key.encode(), int(value)))

    def __call__(self, wav) -> torch.Tensor:
        """wav [4, n] (torch / numpy) -> device tensor [1 + n//hop, n_mels, 7|10]"""
        w = torch.as_tensor(np.asarray(wav) if not isinstance(wav, torch.Tensor) else wav)
        w = w.to(self._dev, torch.float32).contiguous()
        if w.dim() != 2:
            raise ValueError("wav must be [channels, samples]")
        T = int(self.lib.seld_feat_frames(self.h, w.shape[1]))
        out = torch.empty((T, self.n_mels, self.channels), dtype=torch.float32, device=self._dev)
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        rc = self.lib.seld_feat_extract(self.h, w.data_ptr(), int(w.shape[0]), int(w.shape[1]), out.data_ptr(), st)
        if rc:
            raise _lib.SeldError(rc, self.lib.seld_feat_last_error(self.h).decode())
        return out

    def batch(self, wavs) -> torch.Tensor:
        """wavs [n_clips, 4, n] (clips of one length, or a list of such tensors) -> device tensor [n_clips, 1 + n//hop, n_mels, 7|10]:
        the whole batch in one pair of launches (seld_feat_extract_batch); each clip's top_db clamp is its own."""
        if isinstance(wavs, (list, tuple)):
            wavs = torch.stack([torch.as_tensor(np.asarray(w) if not isinstance(w, torch.Tensor) else w) for w in wavs])
        w = torch.as_tensor(np.asarray(wavs) if not isinstance(wavs, torch.Tensor) else wavs).to(self._dev, torch.float32).contiguous()
        if w.dim() != 3:
            raise ValueError("wavs must be [clips, channels, samples]")
        T = int(self.lib.seld_feat_frames(self.h, w.shape[2]))
        out = torch.empty((w.shape[0], T, self.n_mels, self.channels), dtype=torch.float32, device=self._dev)
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        rc = self.lib.seld_feat_extract_batch(self.h, w.data_ptr(), int(w.shape[0]), int(w.shape[1]), int(w.shape[2]), out.data_ptr(), st)
        if rc:
            raise _lib.SeldError(rc, self.lib.seld_feat_last_error(self.h).decode())
        return out

    def normalize(self, feat: torch.Tensor, mean, std, n_frames: int = 3000, eps: float = 1e-8) -> torch.Tensor:
        """preprocess_features_labels (pad/trim to n_frames, :117-149) + apply_normalizer (:226-234)."""
        FC = feat.shape[1] * feat.shape[2]
        as_dev = lambda a: (a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, np.float32))).to(self._dev, torch.float32).reshape(-1).contiguous()
        m, s = as_dev(mean), as_dev(std)
        if m.numel() != FC or s.numel() != FC:
            raise ValueError("mean/std must have freq*chan elements")
        out = torch.empty((n_frames, feat.shape[1], feat.shape[2]), dtype=torch.float32, device=self._dev)
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(self.lib.seld_feat_normalize(feat.data_ptr(), m.data_ptr(), s.data_ptr(), out.data_ptr(), int(feat.shape[0]),
                                                int(n_frames), int(FC), float(eps), st))
        return out

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.seld_feat_destroy(self.h)
                self.h = None
        except Exception:
            pass


class FeatureStatistics:
    """calculate_statistics (feature_extractor.py:218-224) on the device: fold feature tensors [..., freq, chan] (any leading
    shape: one file [T,F,C] or a batch [n,T,F,C]) into per-(freq, chan) sums, then `result()` -> (mean, std) as the reference's
    [1, freq, chan] float arrays (device tensors; population std, numpy's ddof = 0)."""

    def __init__(self, n_freq: int = 64, n_chan: int = 7, device: int | None = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SeldLibraryError("no HIP device visible: seld_amd has no CPU fallback")
        self._dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.F, self.Cn = int(n_freq), int(n_chan)
        self.FC = self.F * self.Cn
        self.acc = torch.zeros(2 * self.FC + 1, dtype=torch.float64, device=self._dev)
        self.scratch = torch.empty(int(self.lib.seld_feat_stats_scratch_doubles(self.FC)), dtype=torch.float64, device=self._dev)

    def update(self, feat) -> "FeatureStatistics":
        f = torch.as_tensor(np.asarray(feat) if not isinstance(feat, torch.Tensor) else feat).to(self._dev, torch.float32).contiguous()
        if f.dim() < 3 or f.shape[-2] != self.F or f.shape[-1] != self.Cn:
            raise ValueError(f"features must be [..., {self.F}, {self.Cn}]")
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(self.lib.seld_feat_stats_accumulate(f.data_ptr(), int(f.numel() // self.FC), self.FC, self.acc.data_ptr(),
                                                       self.scratch.data_ptr(), st))
        return self

    def result(self):
        mean = torch.empty((1, self.F, self.Cn), dtype=torch.float32, device=self._dev)
        std = torch.empty_like(mean)
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(self.lib.seld_feat_stats_finalize(self.acc.data_ptr(), self.FC, mean.data_ptr(), std.data_ptr(), st))
        return mean, std


def calculate_statistics(feature_path: str):
    """reference feature_extractor.calculate_statistics (feature_extractor.py:218-224): mean / std [1, freq, chan] over all frames of
    the sorted *.npy files of a directory — file by file through the device accumulator instead of one host concatenation."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(feature_path, "*.npy")))
    if not files:
        raise ValueError(f"no .npy feature files under {feature_path}")
    stats = None
    for f in files:
        a = np.load(f)
        if stats is None:
            stats = FeatureStatistics(a.shape[-2], a.shape[-1])
        stats.update(a)
    mean, std = stats.result()
    return mean.cpu().numpy(), std.cpu().numpy()


def apply_normalizer_array(feature, mean, std, eps: float = 1e-8, n_frames: int | None = None) -> torch.Tensor:
    """apply_normalizer's arithmetic (feature_extractor.py:226-234) for one feature array [T, F, C] -> device tensor [n_frames or T, F, C]."""
    f = torch.as_tensor(np.asarray(feature) if not isinstance(feature, torch.Tensor) else feature).to("cuda", torch.float32).contiguous()
    T, F_, Cn = f.shape
    lib = _lib.load()
    as_dev = lambda a: (a.detach() if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, np.float32))).to(f.device, torch.float32).reshape(-1).contiguous()
    m, s = as_dev(mean), as_dev(std)
    n_out = T if n_frames is None else int(n_frames)
    out = torch.empty((n_out, F_, Cn), dtype=torch.float32, device=f.device)
    st = C.c_void_p(torch.cuda.current_stream(f.device).cuda_stream)
    _lib.check(lib.seld_feat_normalize(f.data_ptr(), m.data_ptr(), s.data_ptr(), out.data_ptr(), int(T), n_out, F_ * Cn, float(eps), st))
    return out


def extract_features_device(wav, sample_rate, mode="foa", n_mels=64, **kwargs) -> torch.Tensor:
    return FeatureExtractor(sample_rate, mode, n_mels, **kwargs)(wav)


def extract_features(wav, sample_rate, mode="foa", n_mels=64, **kwargs) -> np.ndarray:
    """reference feature_extractor.extract_features (feature_extractor.py:53-88)."""
    return extract_features_device(wav, sample_rate, mode, n_mels, **kwargs).cpu().numpy()
