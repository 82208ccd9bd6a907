"""Host-side mirror of the reference's metrics.SELDMetrics (metrics.py:7-170): the counters live on the
device and are updated by one kernel per call (no per-step host sync); `result()` reads 11 + 4*nc doubles."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def safe_div(x, y, eps=1e-8):
    return x / np.maximum(y, eps)                      # utils.py:23-25


class SELDMetrics:
    def __init__(self, doa_threshold=20, block_size=10, n_classes=14, device=None):
        self.doa_threshold, self.block_size, self.n_classes = doa_threshold, block_size, n_classes
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SeldLibraryError("no HIP device visible: seld_amd has no CPU fallback")
        self._dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self._n = int(self.lib.seld_metrics_state_size(n_classes))
        self.state = torch.zeros(self._n, dtype=torch.float64, device=self._dev)
        self._scratch = None

    def reset_states(self):
        self.state.zero_()

    def update_states(self, y_true, y_pred):
        t = [torch.as_tensor(a, dtype=torch.float32, device=self._dev).contiguous() for a in (*y_true, *y_pred)]
        if t[0].dim() == 2:
            t = [a[None] for a in t]
        B, S, nc = t[0].shape
        if nc != self.n_classes or t[1].shape != (B, S, 3 * nc) or t[2].shape != t[0].shape or t[3].shape != t[1].shape:
            raise ValueError("label / prediction shapes do not match n_classes")
        need = int(self.lib.seld_metrics_scratch_floats(B, S, nc, self.block_size))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float32, device=self._dev)
        st = C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)
        _lib.check(self.lib.seld_metrics_update(t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), B, S, nc,
                                                self.block_size, float(self.doa_threshold), self.state.data_ptr(),
                                                self._scratch.data_ptr(), st))

    def _s(self):
        return self.state.cpu().numpy()

    def result(self):
        """-> ER, F, DE, DE_F (metrics.py:34-53)"""
        TP, FP, TN, FN, S, D, I, Nref, Nsys, total_DE, DE_TP = self._s()[:11]
        ER = safe_div(S + D + I, Nref)
        prec, recall = safe_div(TP, TP + FP), safe_div(TP, TP + FN)
        F = safe_div(2 * prec * recall, prec + recall)
        DE = safe_div(total_DE, DE_TP) if DE_TP > 0 else 180.0
        DE_prec, DE_recall = safe_div(DE_TP, Nsys), safe_div(DE_TP, Nref)
        DE_F = safe_div(2 * DE_prec * DE_recall, DE_prec + DE_recall)
        return float(ER), float(F), float(DE), float(DE_F)

    def class_result(self):
        s, nc = self._s(), self.n_classes
        tp, fp, fn = s[11:11 + nc], s[11 + nc:11 + 2 * nc], s[11 + 3 * nc:11 + 4 * nc]
        return safe_div(tp, tp + fn), safe_div(tp, tp + fp)


def calculate_seld_score(metric_values):
    """metrics.py:157-170"""
    error_rate, f_score, doa_error, recall = metric_values
    return (error_rate + 1 - f_score + doa_error / 180 + 1 - recall) / 4
