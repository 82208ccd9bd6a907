"""Run the model's GEMM shapes through seld_k_gemm a few times each; meant to sit under
`rocprofv3 --kernel-trace --stats` (kernel durations per shape come from the trace)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
rows = 32 * 600
shapes = [  # (M, N, K, transb, tag)
    (rows, 384, 128, 0, "inproj"), (rows, 128, 128, 0, "head1"), (rows, 36, 128, 0, "head2"),
    (rows, 128, 384, 1, "gru_dx"), (rows, 128, 36, 1, "head2_bwd"), (rows, 128, 128, 1, "head1_bwd"),
]
for M, N, K, tb, tag in shapes:
    a = torch.randn(M, K, device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    bias = torch.randn(N, device="cuda")
    c = torch.empty(M, N, device="cuda")
    for _ in range(10):
        rc = lib.seld_k_gemm(P(a), P(b), P(bias), P(c), M, N, K, tb, 0, 0)
        assert rc == 0, rc
    lib.seld_k_set_option(b"gsb_dbg", 0)
    print(tag, "ok")

# split-bf16 path at the model's merged shapes (mode 1: two products sharing A; mode 2: concatenated K)
for M, N, K, tb, mode, tag in [(rows, 384, 128, 0, 1, "sb_inproj"), (rows, 128, 384, 1, 2, "sb_gru_dx"),
                               (rows, 128, 128, 0, 1, "sb_head1"), (rows, 128, 128, 1, 2, "sb_head1_bwd")]:
    a0, a1 = torch.randn(M, K, device="cuda"), torch.randn(M, K, device="cuda")
    b0, b1 = (torch.randn((N, K) if tb else (K, N), device="cuda") for _ in range(2))
    bias = torch.randn(N, device="cuda")
    c0, c1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    for it in range(12):
        lib.seld_k_set_option(b"gsb_dbg", {10: 4, 11: 8}.get(it, 0))   # last two launches: force the 4-wave / the 16-wave form
        rc = lib.seld_k_gemm_sb(P(a0), P(a1), P(b0), P(b1), P(bias), P(bias), P(c0), P(c1), M, N, K, tb, 0, mode)
        assert rc == 0, rc
    lib.seld_k_set_option(b"gsb_dbg", 0)
    print(tag, "ok")
