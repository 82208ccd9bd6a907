"""Near-free timeline of the VAR 1 GRU recurrence kernels inside workgroup 0 (diagnostic build, -DGRU_TRACE2: s_memtime stamps that are
collected by the step's own barrier wait):
    cd seld_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGRU_TRACE2 -c gru.hip -o /tmp/gru_trace2.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../libseld_hip_trace2.so $(ls *.o | grep -v '^gru.o') /tmp/gru_trace2.o
    SELD_HIP_LIB=$PWD/seld_amd/libseld_hip_trace2.so python tools/trace_gru2.py
forward stamps: 0 step start (h reads issued next) | 1 z,r chains done | 2 candidate chain done | 3 h' computed | 4 stores issued (barrier next)
backward stamps: 0 step start | 1 gate gradients written (barrier next) | 2 barrier passed | 3 mat-vec done | 4 carry done"""
import ctypes as C
import os
import sys

import numpy as np
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
lib.seld_k_set_option(b"gru_var", int(sys.argv[1]) if len(sys.argv) > 1 else 3)
for _ in range(3):
    assert lib.seld_k_gru_fwd(P(gx[0]), P(gx[1]), P(U[0]), P(U[1]), P(br[0]), P(br[1]), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), None, B, S, 128) == 0
    assert lib.seld_k_gru_bwd(P(dout), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), P(U[0]), P(U[1]), P(dgx[0]), P(dgx[1]), P(dgh[0]), P(dgh[1]), B, S, 128) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    lib.seld_k_gru_fwd(P(gx[0]), P(gx[1]), P(U[0]), P(U[1]), P(br[0]), P(br[1]), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), None, B, S, 128)
e1.record()
torch.cuda.synchronize()
print(f"gru_fwd (trace build) {e0.elapsed_time(e1) / 10:.4f} ms")
for which, nm, ns in ((4, "gru_fwd", 5), (5, "gru_bwd", 5)):
    buf = np.zeros((8, 8, 6), np.uint64)
    rc = lib.seld_k_gru_timing(which, C.c_void_p(buf.ctypes.data), 64)
    if rc != 0:
        print("(not a -DGRU_TRACE2 build)")
        sys.exit(0)
    t = buf.astype(np.int64)
    print(f"== {nm}: cycles per step (wave 0 stamp 0 to the next step's): {np.diff(t[0, :, 0]).tolist()}")
    for s in range(2, 6):
        t0 = t[:, s, 0].min()
        print(f"  step {100 + s} (next step starts at {int(t[:, s + 1, 0].min() - t0)}):")
        for w in range(8):
            print(f"    w{w}: " + " ".join(f"{int(t[w, s, k] - t0):5d}" for k in range(ns)))
