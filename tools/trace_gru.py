"""Timeline of the GRU recurrence kernels inside one workgroup (diagnostic build):
    cd seld_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DGRU_TRACE -c gru.hip -o /tmp/gru_trace.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../libseld_hip_trace.so $(ls *.o | grep -v '^gru.o') /tmp/gru_trace.o
    SELD_HIP_LIB=$PWD/seld_amd/libseld_hip_trace.so python tools/trace_gru.py
Prints, for recurrence steps 100..107 of workgroup 0 and each of its 8 waves, the s_memtime stamps relative to the step's first
stamp: forward: step start (after the barrier) | mat-vec done | gate tail done; backward: step start | gate gradients written |
barrier passed.  Stamps cost ~50 cycles each and drain the LDS queue: read the STRUCTURE (who waits for whom), not the total."""
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
for rb in (0,):
    for _ in range(3):
        assert lib.seld_k_gru_fwd(P(gx[0]), P(gx[1]), P(U[0]), P(U[1]), P(br[0]), P(br[1]), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), None, B, S, 128) == 0
        assert lib.seld_k_gru_bwd(P(dout), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), P(U[0]), P(U[1]), P(dgx[0]), P(dgx[1]), P(dgh[0]), P(dgh[1]), B, S, 128) == 0
    for which, nm in ((2, "gru_fwd"), (3, "gru_bwd")):
        buf = np.zeros((8, 8, 4), np.uint64)
        rc = lib.seld_k_gru_timing(which, C.c_void_p(buf.ctypes.data), 64)
        if rc != 0:
            print("(not a -DGRU_TRACE build)")
            sys.exit(0)
        t = buf.astype(np.int64)
        print(f"== {nm}: cycles per step (wave 0 start to next start): {np.diff(t[0, :, 0]).tolist()}")
        for s in range(2, 6):
            t0 = t[:, s, 0].min()
            print(f"  step {100 + s}: " + " | ".join(f"w{w}: " + " ".join(f"{int(t[w, s, k] - t0):5d}" for k in range(3)) for w in range(8)))
