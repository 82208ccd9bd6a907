#!/usr/bin/env python3
"""Exact (six bf16 products) against bf16 single-product mode, kernel by kernel at the headline shapes (B = 32, S = 600)."""
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


def timed(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


a_ = torch.randn(4096, 4096, device="cuda")
for _ in range(30):
    (a_ @ a_).sum().item()          # clock ramp
for W in (16, 4):
    x = torch.randn(B, H, W, 64, device="cuda", generator=g)
    w = torch.randn(3, 3, 64, 64, device="cuda", generator=g) * 0.05
    b = torch.randn(64, device="cuda", generator=g)
    z = torch.empty(B, H, W, 64, device="cuda")
    dw = torch.empty(3, 3, 64, 64, device="cuda")
    db = torch.empty(64, device="cuda")
    st = torch.empty(2, 64, device="cuda")
    for one in (0, 1, 0, 1):
        lib.seld_k_set_option(b"bf16_single", one)
        tf = timed(lambda: lib.seld_k_conv3x3_fwd(P(x), P(w), P(b), P(z), P(st), B, H, W, 64, 64))
        tw = timed(lambda: lib.seld_k_conv3x3_wgrad(P(x), P(z), P(dw), P(db), B, H, W, 64, 64))
        print(f"W={W:2d} bf16_single={one}: conv fwd (incl. weight split) {tf:.4f} ms, wgrad (incl. slab combine) {tw:.4f} ms")
lib.seld_k_set_option(b"bf16_single", 0)
