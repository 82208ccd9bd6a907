"""Does a CU finish two GRU recurrences faster side by side than one after the other?  Times seld_k_gru_fwd / _bwd at B = 32, 128 (one
workgroup per CU) and 256 (two per CU IF the kernel's registers allow 4 waves per SIMD: a build with amdgpu_waves_per_eu(4, 4)).
    SELD_HIP_LIB=.../libseld_hip_occ.so python tools/gru_coresidency.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
S = 600
g = torch.Generator(device="cuda").manual_seed(0)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (32, 128, 256):
    gx = [torch.randn(B, S, 384, device="cuda", generator=g) for _ in range(2)]
    U = [torch.randn(128, 384, device="cuda", generator=g) * 0.1 for _ in range(2)]
    br = [torch.randn(384, device="cuda", generator=g) * 0.1 for _ in range(2)]
    h = [torch.empty(B, S, 128, device="cuda") for _ in range(2)]
    sv = [torch.empty(B, S, 4, 128, device="cuda") for _ in range(2)]
    dout = torch.randn(B, S, 128, device="cuda", generator=g)
    dgx = [torch.empty(B, S, 384, device="cuda") for _ in range(2)]
    dgh = [torch.empty(B, S, 384, device="cuda") for _ in range(2)]
    fwd = lambda: lib.seld_k_gru_fwd(P(gx[0]), P(gx[1]), P(U[0]), P(U[1]), P(br[0]), P(br[1]), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), None, B, S, 128)
    bwd = lambda: lib.seld_k_gru_bwd(P(dout), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), P(U[0]), P(U[1]), P(dgx[0]), P(dgx[1]), P(dgh[0]), P(dgh[1]), B, S, 128)
    print(f"B={B:4d} ({2 * B} workgroups): gru_fwd {timed(fwd):.4f} ms, gru_bwd {timed(bwd):.4f} ms")
