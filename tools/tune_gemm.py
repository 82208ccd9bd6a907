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
    print(tag, "ok")
