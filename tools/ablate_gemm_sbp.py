"""Ablations of gemm_sbp_kernel (option gsb_dbg bits 8-11: no MFMA steps / no stores / no next-tile loads / no B staging), HIP-event timings."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
M, N = 19200, 384
a = torch.randn(M, 128, device="cuda")
b0, b1 = torch.randn(128, N, device="cuda"), torch.randn(128, N, device="cuda")
bias = torch.randn(N, device="cuda")
c0, c1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
args = (P(a), None, P(b0), P(b1), P(bias), P(bias), P(c0), P(c1), M, N, 128, 0, 0, 1)
for name, dbg in [("tiled", 0), ("stationary", 64), ("no mfma", 64 | 1 << 8), ("no stores", 64 | 2 << 8), ("no next loads", 64 | 4 << 8), ("no B staging", 64 | 8 << 8),
                  ("no mfma, no stores", 64 | 3 << 8), ("nothing but staging", 64 | 7 << 8), ("nothing", 64 | 15 << 8)]:
    lib.seld_k_set_option(b"gsb_dbg", dbg)
    for _ in range(3):
        assert lib.seld_k_gemm_sb(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.seld_k_gemm_sb(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:24s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per call (incl. ~7 us of B split pre-pass)", flush=True)
lib.seld_k_set_option(b"gsb_dbg", 0)
