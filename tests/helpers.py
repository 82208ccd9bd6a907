"""Shared helpers for the parity tests: error metric and ctypes plumbing."""
import ctypes as C

import numpy as np
import torch

REL_TOL = 1e-4  # north_star: "within 1e-4 rel fp32"


def rel_err(a, b):
    """max|a-b| / max|b| — the tensor-normalised relative error used for every fp32 parity check."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


def check(name, got, ref, tol=REL_TOL):
    e = rel_err(got, ref)
    print(f"[parity] {name:40s} rel_err={e:.3e}  (|ref|max={np.abs(np.asarray(ref)).max():.3e})")
    assert np.isfinite(np.asarray(got)).all(), f"{name}: non-finite output"
    assert e <= tol, f"{name}: rel err {e:.3e} > {tol}"
    return e


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None
