"""Synthetic batches of the metric's geometry for bench.py and examples: there is no dataset in the image.

x ~ N(0, 1) of shape [B, T, 64, C]; SED targets ~ Bernoulli(0.1) per (frame, class) with at least one active entry; DOA
targets = a random unit vector per active (frame, class), zeros elsewhere, in the reference's [x | y | z] block layout
(transforms.py:117-119: doa[..., k*n_classes + c]).  Plain numpy: not part of any compute path.
"""
from __future__ import annotations

import numpy as np


def synthetic_batch(B: int, T: int, F_: int = 64, C: int = 7, n_classes: int = 12, seed: int = 1234, pool_t: int = 5):
    rng = np.random.default_rng(seed)
    S = T // pool_t
    x = rng.standard_normal((B, T, F_, C), dtype=np.float32)
    sed = (rng.random((B, S, n_classes)) < 0.1).astype(np.float32)
    sed[0, 0, 0] = 1.0
    vec = rng.standard_normal((B, S, 3, n_classes))
    vec /= np.linalg.norm(vec, axis=2, keepdims=True)
    doa = (vec * sed[:, :, None, :]).reshape(B, S, 3 * n_classes).astype(np.float32)
    return x, sed, doa
