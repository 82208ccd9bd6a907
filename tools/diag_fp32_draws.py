"""How much does the fp32 ORACLE's own distance from the fp64 oracle (both on the SAME decisions) vary between fp32 evaluations that differ only in
summation order?  resnet50_gru.json at the fixture's size (16 clips of [3000,64,7]); the fp32 evaluation is repeated with different thread counts
(PyTorch's convolution / reduction partitioning changes with them) and with the clips of the batch in another order (BatchNorm's sums).
Prints, per draw, the largest and the median per-variable error (max |g32 - g64| / max |g64|).  ~6 min on 8 cores, ~30 GB.
    python tools/diag_fp32_draws.py [resnet50_gru|xception_gru]"""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import seldnet_oracle as O  # noqa: E402

sp = importlib.util.spec_from_file_location("mgb", os.path.join(ROOT, "tests", "golden", "make_golden_blocks.py"))
mg = importlib.util.module_from_spec(sp)
sp.loader.exec_module(mg)
which = sys.argv[1] if len(sys.argv) > 1 else "resnet50_gru"
Bn = mg.MODELS[which][0]
spec = O.Spec.from_config(mg.model_config(which))
tr, _ = O.variable_specs(spec)
w, st = O.random_weights(spec, 0)
x, ys, yd = O.synthetic_batch(Bn, mg.T, seed=1234)
kw = dict(doa_loss="MSE", loss_weight=(1.0, 1000.0), lr=1e-3, step=1)
rec = {}
t0 = time.time()
r = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float64, record_routing=rec, **kw)
print(f"fp64 step {time.time() - t0:.0f} s", flush=True)
routing = {}
for k, v in rec.items():
    routing[k] = (v["pos"].clone(), v["gate"].clone()) if "pos" in v else v["gate"].clone()
del rec
g64 = r["grad"]


def errs(g32):
    out, off = [], 0
    for name, shape in tr:
        k = int(np.prod(shape))
        g = g64[off:off + k]
        if not (name.startswith("conv") and name.endswith("bias")):
            out.append((float(np.abs(g32[off:off + k].astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-300)), name))
        off += k
    return out


def draw(label, perm=None):
    xs, a, b = (x, ys, yd) if perm is None else (x[perm], ys[perm], yd[perm])
    rt = routing if perm is None else {k: ((v[0][perm], v[1][perm]) if isinstance(v, tuple) else v[perm]) for k, v in routing.items()}
    t0 = time.time()
    g = O.train_step(spec, w, st, xs, a, b, dtype=torch.float32, routing=rt, **kw)["grad"]
    e = errs(g)
    worst = max(e)
    print(f"{label:34s} largest {worst[0]:.3e} ({worst[1]}), median {np.median([v for v, _ in e]):.3e}, 90th pct {np.percentile([v for v, _ in e], 90):.3e}   [{time.time() - t0:.0f} s]",
          flush=True)


only_bn = os.environ.get("DRAWS") == "bn"
for nt in (() if only_bn else (8, 4, 2, 1)):
    torch.set_num_threads(nt)
    draw(f"fp32 given decisions, {nt} threads")
torch.set_num_threads(8)
O.BN_FORM = "shifted"      # the forward value of every BatchNormalization as ONE fma on (scale, shift): what the kernels applied up to round 5
draw("fp32 given decisions, BN as z*scale+shift")
O.BN_FORM = "centred"
rng = np.random.default_rng(0)
for i in range(0 if only_bn else 3):
    draw(f"fp32 given decisions, clip order {i + 1}", perm=torch.as_tensor(rng.permutation(Bn)))
