"""Soak run: N training steps of a model, twice from the same initial state, on fresh batches drawn from one seeded stream — the final weights,
BatchNorm statistics and Adam slots of the two runs must be the same BITS (the step uses two or three streams, events, a Gram matrix under the
GRU recurrence, side-stream kernel gradients: a missing dependency shows up as a difference sooner or later), and everything must stay finite.
    python tools/soak.py [seldnet|seldnet_v1|seldnet_heads|xception_gru|resnet50_gru] [steps] [B] [T]"""
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402  (model_config_of)
from seld_amd import losses, models, train  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "seldnet"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
T = int(sys.argv[4]) if len(sys.argv) > 4 else 600
cfg = bench.model_config_of(name if name in ("xception_gru", "resnet50_gru") else "seldnet")
factory = models.seldnet
if name == "seldnet_v1":
    factory = models.seldnet_v1
if name == "seldnet_heads":
    cfg = copy.deepcopy(cfg)
    cfg["SED_ARGS"].update(kernel_size=3, dense_activation="relu", dropout_rate=0.2)
    cfg["DOA_ARGS"].update(kernel_size=2, dropout_rate=0.1)


def run():
    model = factory((B, T, 64, 7), cfg)
    g = torch.Generator(device="cuda").manual_seed(1234)
    opt = train.Adam(1e-3)
    S = model.S
    t0 = time.perf_counter()
    for i in range(steps):
        x = torch.randn((B, T, 64, 7), device="cuda", generator=g)
        ys = (torch.rand((B, S, 12), device="cuda", generator=g) < 0.1).float()
        yd = (torch.rand((B, S, 36), device="cuda", generator=g) * 2 - 1) * torch.cat([ys] * 3, -1)
        fn = losses.MMSE if i % 2 else losses.MSE
        y_p, sl, dl = train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), fn, (1.0, 1000.0), opt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    w, st = model.get_weights()
    out = (np.array(w), np.array(st), float(sl.float().mean().item()), dt)
    model.close()
    return out


a, b = run(), run()
ok = np.isfinite(a[0]).all() and np.isfinite(a[1]).all() and np.isfinite(a[2])
same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
print(f"{name}: {steps} steps of [{B},{T},64,7] twice ({a[3]:.1f} s, {b[3]:.1f} s): finite {bool(ok)}, bitwise equal {bool(same)}, last BCE {a[2]:.5f}, "
      f"|w| max {np.abs(a[0]).max():.3f}")
sys.exit(0 if ok and same else 1)
