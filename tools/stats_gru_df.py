"""Poll statistics of the barrier-free GRU recurrence (gru_df.hip, -DDF_STATS build): per wave of workgroup 0 over one launch —
polls that found group X / group Y not ready, steps in which group Y's half had to be re-read, kernel cycles.
    SELD_HIP_LIB=<stats build> python tools/stats_gru_df.py [gru_var]"""
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
torch.cuda.synchronize()
for which, nm in ((6, "gru_fwd"), (7, "gru_bwd")):
    buf = np.zeros((8, 16, 8), np.uint64)
    rc = lib.seld_k_gru_timing(which, C.c_void_p(buf.ctypes.data), 128)
    if rc != 0:
        print("(not a -DDF_STATS build)")
        sys.exit(0)
    st = buf.reshape(-1)[:32].reshape(8, 4).astype(np.int64)
    if not st.any():
        continue
    print(f"== {nm}: S = {S} steps")
    for w in range(8):
        print(f"  w{w}: polls A {st[w, 0]:6d}  polls B {st[w, 1]:6d}  steps with a B re-read {st[w, 2]:4d}  cycles/step {st[w, 3] / S:8.1f}")
