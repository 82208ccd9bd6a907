"""Stress for an intermittent non-finite forward output seen ONCE (test_full_length_clips_train_step inside a full suite run, round 5): many fresh contexts,
each one train step at B = 2, T = 3000 from fixed weights, outputs / gradients checked for finiteness and for bit equality with the first context's.
Freed device memory is poisoned with NaN between contexts (a context that reads memory it never wrote shows up as NaN or as a difference).
    python tools/stress_first_step.py [iterations] [key=value ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from oracle import seldnet_oracle as O  # noqa: E402  (weights / batch generators only: nothing of the oracle is timed or shipped)
from seld_amd import losses, models, train  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
opts = dict(kv.split("=") for kv in sys.argv[2:])
churn = int(opts.pop("churn", 0))
idle = float(opts.pop("idle", 0))
cfg = bench.model_config_of("seldnet")
spec = O.Spec.from_config(cfg)
w, st = O.random_weights(spec, 0)
B, T = 2, 3000
x, ys, yd = O.synthetic_batch(B, T, seed=1234)
ref = None
bad = 0
for it in range(iters):
    # poison: a few hundred MB of NaN handed back to the driver before the context allocates
    junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    del junk
    torch.cuda.empty_cache()
    if churn:      # another model's context lives and dies in between (its streams, events and a few GB of buffers)
        other = bench.model_config_of("xception_gru" if it % 2 else "resnet50_gru")
        mo = models.seldnet((2, 600, 64, 7), other)
        xo, yso, ydo = O.synthetic_batch(2, 600, seed=it)
        train.trainstep(mo, xo, (yso, ydo), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), train.Adam(1e-3), False)
        mo.close()
    if idle:
        torch.cuda.synchronize()
        time.sleep(idle)
    model = models.seldnet((B, T, 64, 7), cfg)
    for k, v in opts.items():
        model.set_option(k, int(v))
    model.set_weights(w, st)
    y_p, sl, dl = train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), train.Adam(1e-3), False)
    out = [y_p[0].cpu().numpy(), y_p[1].cpu().numpy(), model.get_grads().copy(), model.get_weights()[0].copy()]
    model.close()
    fin = all(np.isfinite(a).all() for a in out)
    same = ref is None or all(np.array_equal(a, b) for a, b in zip(out, ref))
    if ref is None and fin:
        ref = out
    if not fin or not same:
        bad += 1
        which = [n for n, a in zip(("sed", "doa", "grads", "weights"), out) if not np.isfinite(a).all()]
        diff = [n for n, a, b in zip(("sed", "doa", "grads", "weights"), out, ref or out) if not np.array_equal(a, b)]
        rows = np.argwhere(~np.isfinite(out[0]).all(axis=2)) if not np.isfinite(out[0]).all() else []
        print(f"iteration {it}: non-finite in {which}, differs in {diff}, sed rows {rows[:3].tolist()} .. {rows[-3:].tolist() if len(rows) else []}", flush=True)
print(f"{iters} fresh contexts, options {opts}: {bad} bad")
