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
dout = torch.randn(B, S, 128, device="cuda", generator=g)
dgx = [torch.empty(B, S, 384, device="cuda") for _ in range(2)]
dgh = [torch.empty(B, S, 384, device="cuda") for _ in range(2)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()      # seld_k_* launch on the null stream, which torch's default stream is
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fwd(save):
    rc = lib.seld_k_gru_fwd(P(gx[0]), P(gx[1]), P(U[0]), P(U[1]), P(br[0]), P(br[1]), P(h[0]), P(h[1]),
                            P(sv[0]) if save else None, P(sv[1]) if save else None, None, B, S, 128)
    assert rc == 0, rc


def bwd():
    rc = lib.seld_k_gru_bwd(P(dout), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), P(U[0]), P(U[1]), P(dgx[0]), P(dgx[1]), P(dgh[0]), P(dgh[1]),
                            B, S, 128)
    assert rc == 0, rc


import numpy as np  # noqa: E402

# step-body variants (option "gru_var": bit 0 forward VAR 1, bit 1 backward VAR 1, bit 3 falling priority; 0 = round 3; gru.hip): outputs against variant 0's, then same-box timings
lib.seld_k_set_option(b"gru_var", 0)
fwd(1)
bwd()
torch.cuda.synchronize()
ref = [t.clone() for t in (h[0], h[1], sv[0], sv[1], dgx[0], dgx[1], dgh[0], dgh[1])]
names = ("h_f", "h_b", "sv_f", "sv_b", "dgx_f", "dgx_b", "dgh_f", "dgh_b")
variants = [int(v) for v in sys.argv[1:]] or [3, 11]
for var in variants:
    lib.seld_k_set_option(b"gru_var", var)
    for t in (h[0], h[1], sv[0], sv[1], dgx[0], dgx[1], dgh[0], dgh[1]):
        t.zero_()
    fwd(1)
    bwd()
    torch.cuda.synchronize()
    for name, a, b in zip(names, (h[0], h[1], sv[0], sv[1], dgx[0], dgx[1], dgh[0], dgh[1]), ref):
        e = float((a - b).abs().max() / b.abs().max())
        print(f"gru_var={var} vs 0: {name:6s} max rel diff {e:.2e}")
        assert e < 2e-5, (name, e)
    # inference form (no saved gates) must give the same h
    hk = h[0].clone()
    h[0].zero_()
    fwd(0)
    torch.cuda.synchronize()
    assert torch.equal(hk, h[0]), "h differs between the saving and the inference form"
for rep in range(3):
    for var in [0] + variants:
        lib.seld_k_set_option(b"gru_var", var)
        print(f"gru_var={var}: gru_fwd {timed(lambda: fwd(1)):.4f} ms (saving gates), {timed(lambda: fwd(0)):.4f} ms (inference), gru_bwd {timed(bwd):.4f} ms   [B={B}, S={S}]")
# odd sequence lengths / short chunks through the same kernels
for S2 in (1, 7, 17, 33):
    for var in [0] + variants:
        lib.seld_k_set_option(b"gru_var", var)
        hh = [torch.zeros(B, S2, 128, device="cuda") for _ in range(2)]
        ss = [torch.zeros(B, S2, 4, 128, device="cuda") for _ in range(2)]
        gg = [gx[k][:, :S2].contiguous() for k in range(2)]
        assert lib.seld_k_gru_fwd(P(gg[0]), P(gg[1]), P(U[0]), P(U[1]), P(br[0]), P(br[1]), P(hh[0]), P(hh[1]), P(ss[0]), P(ss[1]), None, B, S2, 128) == 0
        torch.cuda.synchronize()
        if var == 0:
            r2 = [t.clone() for t in hh + ss]
        else:
            for a, b in zip(hh + ss, r2):
                e = float((a - b).abs().max() / b.abs().max())
                assert e < 2e-5, (S2, var, e)
print("short sequences ok")
lib.seld_k_set_option(b"gru_var", 11)
