"""Why do the conv-stack gradients differ from the fp64 oracle by more than 1e-4 at full size?  Not summation error: ROUTING
FLIPS — pooling windows whose two largest elements (or whose maximum and 0) lie within one fp32 rounding of each other send a
whole gradient element to a different pixel (or gate it differently) than the fp64 evaluation does.

Test (tests/test_model_gpu.py::test_parity_given_identical_routing at a size the suite can afford; here at any size): evaluate the
fp64 oracle a second time with the MaxPool/ReLU routing of all three conv blocks replaced by the decisions the library took
(seld_debug_pool_routing).  The HIP gradients must agree with that evaluation to ~1e-6 while they differ from the free-running
fp64 oracle by the flips' worth, and every flipped decision must have an fp64 margin fp32 cannot resolve.

  python tools/diag_routing_flips.py [B=32] [T=3000] [MSE|MMSE]     (GPU box; ~40 GB of host memory and ~15 min at B=32)"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g
from oracle import seldnet_oracle as O
from seld_amd import _lib, losses, models, train


def _heartbeat():      # the fp64 oracle runs for minutes without output: gpurun takes 7 silent minutes for a hang
    t0 = time.time()
    while True:
        time.sleep(60)
        print(f"[heartbeat] {time.time() - t0:.0f} s", flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
mode = sys.argv[3] if len(sys.argv) > 3 else "MSE"
cfg = g.SELDNET_CONFIG
spec = O.Spec.from_config(cfg)
w, st = O.random_weights(spec, 0)
x, ys, yd = O.synthetic_batch(B, T, seed=1234)
model = models.seldnet((B, T, 64, 7), cfg)
model.set_weights(w, st)
train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.get_doa_loss(mode), (1.0, 1000.0), train.Adam(1e-3), False)
g_gpu = model.get_grads().astype(np.float64)
routing = {}
H, W = T, 64
for i, (pt, pf) in enumerate(spec.pools):
    shape = (B, H // pt, W // pf, 64)
    pos = torch.empty(shape, dtype=torch.uint8, device="cuda")
    gate = torch.empty(shape, dtype=torch.uint8, device="cuda")
    _lib.check(model.lib.seld_debug_pool_routing(model.ctx, i, C.c_void_p(pos.data_ptr()), C.c_void_p(gate.data_ptr())), model.ctx)
    routing[i] = (pos.cpu().to(torch.int64), gate.cpu().bool())
    H, W = H // pt, W // pf
del model
torch.cuda.empty_cache()
kw = dict(doa_loss=mode, loss_weight=(1.0, 1000.0), lr=1e-3, step=1, dtype=torch.float64)
free = {}
g_free = O.train_step(spec, w, st, x, ys, yd, record_routing=free, **kw)["grad"]
print(f"B={B} T={T} doa_loss={mode}")
for i in range(len(spec.pools)):
    pos, gate = routing[i]
    f = free[i]
    arg = (pos != f["pos"]) & gate & f["gate"]
    chosen = f["windows"].gather(-1, pos.unsqueeze(-1)).squeeze(-1)
    margin = (f["top"] - chosen)[arg]
    gflip = gate != f["gate"]
    gmargin = f["top"].abs()[gflip]
    print(f"block {i}: {pos.numel()} pooled elements; argmax flips {int(arg.sum())} (fp64 margins {np.sort(margin.numpy())[::-1][:8]}); "
          f"ReLU gate flips {int(gflip.sum())} (|top| {np.sort(gmargin.numpy())[::-1][:8]})", flush=True)
    del f["windows"]
free.clear()
g_rout = O.train_step(spec, w, st, x, ys, yd, routing=routing, **kw)["grad"]
tr, _ = O.variable_specs(spec)
print("%-28s %-22s %-26s %-22s" % ("variable", "HIP vs fp64 oracle", "HIP vs fp64 WITH its routing", "routed fp64 vs free fp64"))
off = 0
for n, shp in tr:
    k = int(np.prod(shp))
    a, f, r = g_gpu[off:off + k], g_free[off:off + k], g_rout[off:off + k]
    off += k
    if n.startswith("conv") and n.endswith("bias"):
        continue
    den = np.abs(f).max()
    print("%-28s %-22.3e %-26.3e %-22.3e" % (n, np.abs(a - f).max() / den, np.abs(a - r).max() / den, np.abs(r - f).max() / den))
