"""CPU oracle for sliding-window inference (evaluator.py:16-50 == trainv2.py:158-192).
TEST INFRASTRUCTURE ONLY (see oracle/seldnet_oracle.py header)."""
import numpy as np
import torch

from . import seldnet_oracle as O


def frame(x, win, step):
    """tf.signal.frame(x, win, step, axis=0, pad_end=False)"""
    n = 1 + (x.shape[0] - win) // step
    return np.stack([x[w * step:w * step + win] for w in range(n)], 0)


def overlap_average(y):
    """tf.signal.overlap_and_add(y^T, 1)^T / counts for y [n_win, L, D]"""
    n, L, D = y.shape
    out = np.zeros((n - 1 + L, D), np.float64)
    cnt = np.zeros((n - 1 + L, 1), np.float64)
    for w in range(n):
        out[w:w + L] += y[w]
        cnt[w:w + L] += 1
    return out / cnt


def ensemble_outputs(spec, flat_w, flat_state, xs, win_size=300, step_size=5, dtype=torch.float64):
    res = []
    for x in xs:
        wins = frame(np.asarray(x), win_size, step_size)
        S = win_size // step_size
        z = np.zeros((wins.shape[0], S, spec.n_classes), np.float32)
        r = O.test_step(spec, flat_w, flat_state, wins, z, np.zeros((wins.shape[0], S, 3 * spec.n_classes), np.float32), dtype=dtype)
        res.append((overlap_average(r["sed"]), overlap_average(r["doa"])))
    return res
