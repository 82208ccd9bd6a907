#!/usr/bin/env python3
"""Per-variable gradient error of the resnet50_gru train step: HIP vs the fp64 oracle, with the fp32 oracle's own error beside it.
    python tools/diag_resnet_grads.py [B T b0,b1,b2,b3 [option=value ...]]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import seldnet_oracle as O
from seld_amd import losses, models, train

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 300)
blocks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 1, 1, 1]
opts = [a.split("=") for a in sys.argv[4:]]
cfg = {"FIRST": "resnet50_block", "FIRST_ARGS": {"filters": 32, "block_num": blocks},
       "SECOND": "bidirectional_GRU_block", "SECOND_ARGS": {"units": [128, 128], "dropout_rate": 0.0},
       "SED": "simple_dense_block", "SED_ARGS": {"units": [128], "n_classes": 14, "activation": "sigmoid", "name": "sed_out"},
       "DOA": "simple_dense_block", "DOA_ARGS": {"units": [128], "n_classes": 42, "activation": "tanh", "name": "doa_out"}, "n_classes": 12}
spec = O.Spec.from_config(cfg)
w, st = O.random_weights(spec, 7)
x, ys, yd = O.synthetic_batch(B, T, seed=19)
model = models.seldnet((B, T, 64, 7), cfg)
for k, v in opts:
    model.set_option(k, int(v))
model.set_weights(w, st)
kw = dict(doa_loss="MSE", loss_weight=(1.0, 1000.0), lr=1e-3, step=1)
r64 = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float64, **kw)["grad"]
r32 = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float32, **kw)["grad"]
train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.get_doa_loss("MSE"), (1.0, 1000.0), train.Adam(1e-3))
g = model.get_grads()
print(f"{'variable':28s} {'|ref|max':>10s} {'hip':>10s} {'fp32 oracle':>12s}")
for n, off, sh in model.variables:
    k = int(np.prod(sh)); r = r64[off:off + k]; den = np.abs(r).max() or 1.0
    print(f"{n:28s} {den:10.3e} {np.abs(g[off:off+k]-r).max()/den:10.3e} {np.abs(r32[off:off+k]-r).max()/den:12.3e}")
