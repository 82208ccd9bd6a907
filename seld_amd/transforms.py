"""Mirror of the reference's transforms.py for the accelerated path: the batch augmentations of train.get_dataset
(train.py:157-165) on DEVICE tensors through the C ABI (seld_aug_mask, seld_aug_gather_sign), plus the pure index
helpers.  Randomness: the reference draws with tf.random.uniform inside tf.data; here a numpy Generator makes the
same draws on the host (same distributions: size ~ U{0..max_mask_size-1}, offset ~ U{0..total-size-1},
flip ~ U{0,1}^3, swap ~ U{0,2}, acs index ~ U{0..7}) and only the small tables go to the device.

  mask(specs, axis, max_mask_size, period, n_mask, rng)   transforms.py:6-44    specs [B,T,F,C] batch (each sample and
                                                                               each period-frame segment its own draw)
  foa_intensity_vec_aug(x, y, rng)                         transforms.py:73-114
  acs_aug(x, y, rng)                                       transforms.py:159-207
  mic_gcc_perm(mic_perm)                                   transforms.py:122-140
  split_total_labels_to_sed_doa(x, y)                      transforms.py:117-119
mcs_aug (CGMM mask estimation in float64, transforms.py:237-292) is outside the accelerated path."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

# transforms.py:146-155
channel_list = [
    [[1, 3, 0, 2], [0, -3, -2, 1]],
    [[3, 1, 2, 0], [0, -3, 2, -1]],
    [[0, 1, 2, 3], [0, 1, 2, 3]],
    [[1, 0, 3, 2], [0, -1, -2, 3]],
    [[2, 0, 3, 1], [0, 3, -2, -1]],
    [[0, 2, 1, 3], [0, 3, 2, 1]],
    [[3, 2, 1, 0], [0, -1, 2, -3]],
    [[2, 3, 0, 1], [0, 1, -2, -3]],
]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_tensor(x, name):
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous float32 CUDA tensor (augmentation runs on the device)")
    return x


def split_total_labels_to_sed_doa(x, y):
    n_classes = y.shape[-1] // 4
    return x, (y[..., :n_classes], y[..., n_classes:])


def draw_mask(rng, n, total, max_mask_size):
    """(size, offset) int32 [n]: tf.random.uniform([], maxval=max_mask_size) and ([], maxval=total - size)."""
    if max_mask_size is None:
        max_mask_size = total
    size = rng.integers(0, max_mask_size, size=n)
    if (total - size <= 0).any():
        raise ValueError("max_mask_size must be smaller than the masked axis")      # tf.random.uniform(maxval=0) fails too
    offset = (rng.random(n) * (total - size)).astype(np.int64)
    return size.astype(np.int32), offset.astype(np.int32)


def mask(specs, axis, max_mask_size=None, period=100, n_mask=1, rng=None, draws=None):
    """In-place on specs [B,T,F,C] (device).  axis: -3 (time) or -2 (frequency), as in the reference's per-sample call
    (there specs is [T,F,C]).  `draws` = (size, offset) int arrays [B * T//period] overrides the generator (tests)."""
    specs = _dev_tensor(specs, "specs")
    if specs.dim() != 4 or axis not in (-3, -2):
        raise ValueError("mask: specs must be [B,T,F,C] and axis -3 or -2")
    B, T, F, Cc = specs.shape
    if T % period != 0:
        raise ValueError("(spec time length / period)' rest must be 0")
    lib = _lib.load()
    nseg = T // period
    total = period if axis == -3 else F
    rng = rng or np.random.default_rng()
    for _ in range(n_mask):
        size, offset = draws if draws is not None else draw_mask(rng, B * nseg, total, max_mask_size)
        sz = torch.as_tensor(np.ascontiguousarray(size, np.int32)).to(specs.device)
        of = torch.as_tensor(np.ascontiguousarray(offset, np.int32)).to(specs.device)
        args = (_ptr(of), _ptr(sz), None, None) if axis == -3 else (None, None, _ptr(of), _ptr(sz))
        rc = lib.seld_aug_mask(_ptr(specs), B, T, F, Cc, period, *args, _stream())
        if rc:
            raise _lib.SeldError(f"seld_aug_mask failed ({rc})")
    return specs


def _gather_sign(t, B, outer, R, inner, src, sgn):
    lib = _lib.load()
    s = torch.as_tensor(np.ascontiguousarray(src, np.int32)).to(t.device)
    g = torch.as_tensor(np.ascontiguousarray(sgn, np.float32)).to(t.device)
    rc = lib.seld_aug_gather_sign(_ptr(t), B, outer, R, inner, _ptr(s), _ptr(g), _stream())
    if rc:
        raise _lib.SeldError(f"seld_aug_gather_sign failed ({rc})")


def foa_tables(flip, p):
    """src/sgn tables of foa_intensity_vec_aug for x (7 channels) and y rows (4): transforms.py:88-111."""
    flip, p = np.asarray(flip), np.asarray(p).reshape(-1)
    B = flip.shape[0]
    s = 1 - 2 * flip.astype(np.int64)                                   # sign per ORIGINAL axis, applied before the gather
    perm = np.stack([p, np.ones_like(p), 2 - p], -1)
    check = (perm != np.array([[0, 1, 2]])).sum(-1, keepdims=True)
    fp = (perm + check) % 3
    b = np.arange(B)[:, None]
    x_src = np.concatenate([np.zeros((B, 1), np.int64), 1 + perm, 4 + fp], -1)
    x_sgn = np.concatenate([np.ones((B, 4)), s[b, fp]], -1)
    y_src = np.concatenate([np.zeros((B, 1), np.int64), 1 + fp], -1)
    y_sgn = np.concatenate([np.ones((B, 1)), s[b, fp]], -1)
    return x_src, x_sgn, y_src, y_sgn


def foa_intensity_vec_aug(x, y, rng=None, draws=None):
    """x [B,T,F,7], y [B,S,4*n_classes] device tensors, modified in place and returned.  `draws` = (flip [B,3], p [B])."""
    x, y = _dev_tensor(x, "x"), _dev_tensor(y, "y")
    B, T, F, Cc = x.shape
    if Cc != 7:
        raise ValueError("foa_intensity_vec_aug expects 7 feature channels")
    rng = rng or np.random.default_rng()
    flip, p = draws if draws is not None else (rng.integers(0, 2, (B, 3)), 2 * rng.integers(0, 2, B))
    x_src, x_sgn, y_src, y_sgn = foa_tables(flip, p)
    _gather_sign(x, B, T * F, 7, 1, x_src, x_sgn)
    nc = y.shape[-1] // 4
    _gather_sign(y, B, int(np.prod(y.shape[1:-1])), 4, nc, y_src, y_sgn)
    return x, y


def mic_gcc_perm(mic_perm):
    pairs = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
    mic_perm = np.asarray(mic_perm)
    order = {(a, b): k for k, (a, b) in enumerate(pairs.tolist())}
    out = np.empty((mic_perm.shape[0], 6), np.int64)
    for k, (a, b) in enumerate(pairs.tolist()):
        for i in range(mic_perm.shape[0]):
            u, v = int(mic_perm[i, a]), int(mic_perm[i, b])
            out[i, k] = order[(min(u, v), max(u, v))]
    return out


def acs_tables(idx):
    """src/sgn tables of acs_aug for x (17 channels) and y rows (4): transforms.py:175-203."""
    cl = np.array(channel_list)[np.asarray(idx)]
    B = cl.shape[0]
    foa_flip = cl[:, 1, 1:]
    sgn = np.sign(foa_flip)
    foa_perm = sgn * foa_flip - 1
    check = (foa_perm != np.array([0, 1, 2])).sum(-1, keepdims=True)
    fp = (foa_perm + check) % 3
    mic = cl[:, 0, :]
    x_src = np.concatenate([np.zeros((B, 1), np.int64), 1 + foa_perm, 4 + fp, 7 + mic, 11 + mic_gcc_perm(mic)], -1)
    x_sgn = np.concatenate([np.ones((B, 4)), sgn, np.ones((B, 10))], -1)           # sign indexed by OUTPUT position
    y_src = np.concatenate([np.zeros((B, 1), np.int64), 1 + fp], -1)
    y_sgn = np.concatenate([np.ones((B, 1)), sgn], -1)
    return x_src, x_sgn, y_src, y_sgn


def acs_aug(x, y, rng=None, draws=None):
    """x [B,T,F,17] (foa 4, intensity 3, mic 4, gcc 6), y [B,S,4*n_classes]; in place.  `draws` = idx [B] in [0,8)."""
    x, y = _dev_tensor(x, "x"), _dev_tensor(y, "y")
    B, T, F, Cc = x.shape
    if Cc != 17:
        raise ValueError("acs_aug expects 17 feature channels")
    rng = rng or np.random.default_rng()
    idx = draws if draws is not None else rng.integers(0, 8, B)
    x_src, x_sgn, y_src, y_sgn = acs_tables(idx)
    _gather_sign(x, B, T * F, 17, 1, x_src, x_sgn)
    nc = y.shape[-1] // 4
    _gather_sign(y, B, int(np.prod(y.shape[1:-1])), 4, nc, y_src, y_sgn)
    return x, y
