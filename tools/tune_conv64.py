#!/usr/bin/env python3
"""conv2 / conv3 forward (= input-gradient) kernels at the headline shapes through seld_k_conv3x3_fwd, each kernel choice in turn;
run under `rocprofv3 --kernel-trace --stats` (durations come from the trace)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
B, H = 32, 600
g = torch.Generator(device="cuda").manual_seed(0)
for W in (16, 4):
    x = torch.randn(B, H, W, 64, device="cuda", generator=g)
    w = torch.randn(3, 3, 64, 64, device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g)
    z = torch.empty(B, H, W, 64, device="cuda")
    st = torch.empty(2, 64, device="cuda")
    for opts in ({"conv64_dbuf": 1}, {"conv64_dbuf": 0}):
        for k, v in opts.items():
            assert lib.seld_k_set_option(k.encode(), v) == 0
        for _ in range(6):
            rc = lib.seld_k_conv3x3_fwd(P(x), P(w), P(b), P(z), P(st), B, H, W, 64, 64)
            assert rc == 0, rc
lib.seld_k_set_option(b"conv64_dbuf", 1)
print("ok")
