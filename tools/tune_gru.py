"""GRU recurrence kernels at the headline shape (B=32, S=600) through seld_k_gru_fwd, with and without the
saved-gates output; meant to run under `rocprofv3 --kernel-trace` (durations come from the trace)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
B, S = 32, 600
g = torch.Generator(device="cuda").manual_seed(0)
gx = [torch.randn(B, S, 384, device="cuda", generator=g) for _ in range(2)]
U = [torch.randn(128, 384, device="cuda", generator=g) * 0.1 for _ in range(2)]
br = [torch.randn(384, device="cuda", generator=g) * 0.1 for _ in range(2)]
h = [torch.empty(B, S, 128, device="cuda") for _ in range(2)]
sv = [torch.empty(B, S, 4, 128, device="cuda") for _ in range(2)]
for save in (1, 0, 1, 0):
    for _ in range(5):
        rc = lib.seld_k_gru_fwd(P(gx[0]), P(gx[1]), P(U[0]), P(U[1]), P(br[0]), P(br[1]), P(h[0]), P(h[1]),
                                P(sv[0]) if save else None, P(sv[1]) if save else None, None, B, S, 128)
        assert rc == 0, rc
print("ok")
dout = torch.randn(B, S, 128, device="cuda", generator=g)
dgx = [torch.empty(B, S, 384, device="cuda") for _ in range(2)]
dgh = [torch.empty(B, S, 384, device="cuda") for _ in range(2)]
for _ in range(5):
    rc = lib.seld_k_gru_bwd(P(dout), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), P(U[0]), P(U[1]), P(dgx[0]), P(dgx[1]), P(dgh[0]), P(dgh[1]),
                            B, S, 128)
    assert rc == 0, rc
print("bwd ok")

# -DGRU_TIMING build (SELD_HIP_LIB=... python tools/tune_gru.py): per-phase cycle shares of the last launches
import numpy as np
for which, names in ((0, ("h read + mat-vec", "gate tail", "barrier wait", "chunk commit + barrier")),
                     (1, ("gate gradients -> LDS/global", "barrier wait", "coefficients + mat-vec + fold", "-"))):
    buf = np.zeros((2 * B, 4), np.uint64)
    rc = lib.seld_k_gru_timing(which, C.c_void_p(buf.ctypes.data), 2 * B)
    if rc != 0:
        print("(normal build: no phase counters)")
        break
    per_step = buf.astype(np.float64).mean(0) / S
    tot = per_step.sum()
    print(("gru_fwd" if which == 0 else "gru_bwd") + f": {tot:.0f} stamped cycles per recurrence step (wave 0, mean over {2 * B} workgroups): " +
          ", ".join(f"{n} {v:.0f} ({100 * v / tot:.0f} %)" for n, v in zip(names, per_step) if n != "-"))
