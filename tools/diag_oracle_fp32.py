"""How far is an fp32 evaluation of the SAME train step from the fp64 one, per variable?  (CPU oracle in both
precisions.)  Calibrates what "parity at 1e-4" can mean for the cancellation-heavy conv gradients at full size."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g
from oracle import seldnet_oracle as O

B, T = int(sys.argv[1]), int(sys.argv[2])
spec = O.Spec.from_config(g.SELDNET_CONFIG)
w, st = O.random_weights(spec, 0)
x, ys, yd = O.synthetic_batch(B, T, seed=1234)
kw = dict(doa_loss="MSE", loss_weight=(1.0, 1000.0), lr=1e-3, step=1)
r64 = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float64, **kw)["grad"]
r32 = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float32, **kw)["grad"]
off = 0
for n, s in O.variable_specs(spec)[0]:
    k = int(np.prod(s))
    a, b = r32[off:off + k], r64[off:off + k]
    off += k
    if "bias" in n and n.startswith("conv"):
        continue
    print("%-28s fp32 oracle vs fp64: %.3e" % (n, np.abs(a - b).max() / np.abs(b).max()))
