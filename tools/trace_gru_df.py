"""Timeline of the barrier-free GRU recurrence (gru_df.hip) inside workgroup 0 (diagnostic build, -DDF_TRACE):
    cd seld_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDF_TRACE -c gru_df.hip -o /tmp/gru_df_trace.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/build_diag/libseld_hip_dftrace.so $(ls *.o | grep -v '^gru_df.o') /tmp/gru_df_trace.o
    SELD_HIP_LIB=$PWD/build_diag/libseld_hip_dftrace.so python tools/trace_gru_df.py [gru_var]
stamps: 0 step start | 1 group X's half landed | 2 phase A FMAs issued | 3 group Y's half landed | 4 h' published | polls A | polls B"""
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
lib.seld_k_set_option(b"gru_var", int(sys.argv[1]) if len(sys.argv) > 1 else 27)
fwd = lambda: lib.seld_k_gru_fwd(P(gx[0]), P(gx[1]), P(U[0]), P(U[1]), P(br[0]), P(br[1]), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), None, B, S, 128)
bwd = lambda: lib.seld_k_gru_bwd(P(dout), P(h[0]), P(h[1]), P(sv[0]), P(sv[1]), P(U[0]), P(U[1]), P(dgx[0]), P(dgx[1]), P(dgh[0]), P(dgh[1]), B, S, 128)
for _ in range(3):
    assert fwd() == 0 and bwd() == 0
for nm, fn in (("gru_fwd", fwd), ("gru_bwd", bwd)):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{nm} (trace build) {e0.elapsed_time(e1) / 10:.4f} ms")
for which, nm in ((6, "gru_fwd"), (7, "gru_bwd")):
    buf = np.zeros((8, 16, 8), np.uint64)
    rc = lib.seld_k_gru_timing(which, C.c_void_p(buf.ctypes.data), 128)
    if rc != 0:
        print("(not a -DDF_TRACE build)")
        sys.exit(0)
    t = buf.astype(np.int64)
    if not t[:, :, 0].any():
        continue
    print(f"== {nm}: cycles per step (wave 0, stamp 4 to the next step's): {np.diff(t[0, :, 4]).tolist()}")
    t0 = t[:, 2, 0].min()
    for s in range(2, 9):
        print(f"  step {96 + s}:")
        for w in range(8):
            print(f"    w{w}: " + " ".join(f"{int(t[w, s, k] - t0):6d}" for k in range(5)) + f"   polls {int(t[w, s, 5])} {int(t[w, s, 6])}")
