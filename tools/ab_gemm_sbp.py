"""A/B of the K = 128 products: gemm_sbp_kernel (B stationary, persistent workgroups; option gsb_dbg bit 6) against gemm_sb_kernel<4> (the default),
same inputs: bit equality of the results and HIP-event timings.  Shapes: the GRU input projections (mode 1, N = 2 x 384), resnet50_block's
128 -> 512 product (mode 0), ragged row counts."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def run(M, N, mode, dbg, reps):
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randn(M, 128, device="cuda", generator=g)
    b0 = torch.randn(128, N, device="cuda", generator=g) * 0.1
    b1 = torch.randn(128, N, device="cuda", generator=g) * 0.1
    bias0, bias1 = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    c0 = torch.full((M, N), float("nan"), device="cuda")
    c1 = torch.full((M, N), float("nan"), device="cuda")
    lib.seld_k_set_option(b"gsb_dbg", dbg)
    args = (P(a), None, P(b0), P(b1) if mode else None, P(bias0), P(bias1) if mode else None, P(c0), P(c1) if mode else None, M, N, 128, 0, 0, mode)
    for _ in range(3):
        assert lib.seld_k_gemm_sb(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.seld_k_gemm_sb(*args)
    e1.record()
    torch.cuda.synchronize()
    lib.seld_k_set_option(b"gsb_dbg", 0)
    return c0, c1, e0.elapsed_time(e1) / reps * 1e3, a, b0, bias0


for M, N, mode in [(19200, 384, 1), (38400, 512, 0), (19200, 256, 0), (19200 - 13, 384, 1), (100, 256, 0), (33, 384, 1)]:
    n0, n1, t_new, a, b0, bias0 = run(M, N, mode, 64, 20)
    o0, o1, t_old, _, _, _ = run(M, N, mode, 0, 20)
    same = torch.equal(n0, o0) and (not mode or torch.equal(n1, o1))
    ref = a.double() @ b0.double() + bias0.double()
    err = float((n0.double() - ref).abs().max() / ref.abs().max())
    print(f"M={M} N={N} mode={mode}: stationary {t_new:.1f} us (incl. the B split pre-pass), tiled {t_old:.1f} us, bit-identical {same}, max err vs fp64 {err:.2e}", flush=True)
    assert same and err < 1e-6
