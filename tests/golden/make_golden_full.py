"""Generates tests/golden/seldnet_full_b32_t3000_{mse,mmse}.npz: the HEADLINE configuration (BASELINE.json configs[1]:
seldnet.json, 32 clips of [3000,64,7]) evaluated once by the CPU oracle in fp64, in the build container.

Run from the repo root (needs ~30 GB of host memory, ~3 min on 8 cores):  python tests/golden/make_golden_full.py

Stored per case (KB-sized; inputs and weights are regenerated from seeds by the test):
  * train outputs: strided samples of sed / doa, sloss, dloss (the [B,S] rows in MSE mode: a strided sample);
  * per trainable variable: an evenly strided gradient sample (<= 512 elements; conv0.kernel whole), its indices are
    recomputed by `sample_index`, the gradient's l2 norm and max |.|;
  * BN moving statistics after the step (all 384), a strided sample of the post-Adam weights;
  * `bar_fp32`: per variable, max|g32 - g64| / max|g64| of the SAME oracle evaluated in fp32 -- what two evaluations of
    the reference's own arithmetic differ by at this size.  tests/test_model_gpu.py::test_full_batch_vs_golden holds the HIP
    path to max(1e-4, bar) per variable.
  * DECISIONS (since round 3, as make_golden_blocks.py stores them): per conv block the MaxPool(ReLU(.)) routing of the fp64 forward —
    `dec.pool{i}.near` = flat indices of the pooled elements whose fp64 margin (top1 - top2, or |top1| for the ReLU gate) is below
    `dec.pool{i}.eps` (round 5: make_golden_blocks.margin_rule — 8 x the fp32 oracle's own error on the window maxima `dec.pool{i}.err32`,
    clamped to [1e-5, 2e-3]; a fixed 1e-5 before), `dec.pool{i}.near_val` = the fp64 decision at each of them (round 4: tests inject these with seld_debug_set_routing and then
    hold every variable's gradient to 1e-4 of THIS free-running fp64 evaluation, whose own decisions they are),
    `dec.pool{i}.digest` = (count, position-weighted checksum) of value = 0 | 1 + argmax position over all OTHER elements.  The
    library's routing (seld_debug_pool_routing), digested with the same indices excluded, must give the same pair: every decision it
    takes differently from fp64 then has an fp64 margin below 1e-5 — asserted at B = 32 without an oracle on the GPU box.
The reference cannot be imported here (TensorFlow absent), so these vectors pin the oracle restatement,
not TensorFlow: parity stays "unpinned" in the sense of DESIGN.md §0."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import seldnet_oracle as O  # noqa: E402
from __graft_entry__ import SELDNET_CONFIG  # noqa: E402

B, T = 32, 3000
MAX_SAMPLE = 512


def sample_index(name: str, n: int) -> np.ndarray:
    """Indices (into the variable's flat gradient) that the fixture stores."""
    if name == "conv0.kernel" or n <= MAX_SAMPLE:
        return np.arange(n)
    return np.linspace(0, n - 1, MAX_SAMPLE).astype(np.int64)


def out_sample_index(n: int, k: int = 4096) -> np.ndarray:
    return np.linspace(0, n - 1, min(n, k)).astype(np.int64)


def near_ties(spec, taps, wd):
    """[n_conv][4]: per conv block, the number of pooling windows (window = one pooled element) of the fp64 forward whose
    routing an fp32 evaluation can flip: columns = (top1 > 0 and top1 - top2 < d) for d = 1e-6, 1e-5, then |top1| < d for the
    same d (the ReLU gate).  Each flip moves one whole routed gradient element: the mechanism behind `bar_fp32`."""
    rows = []
    for i, pool in enumerate(spec.pools):
        z = taps.pop(f"conv{i}.z")
        Bz, H, Wd, C = z.shape
        m, v = z.mean(axis=(0, 1, 2)), z.var(axis=(0, 1, 2))
        sc = wd[f"bn{i}.gamma"].numpy().astype(np.float64) / np.sqrt(v + O.BN_EPS)
        sh = wd[f"bn{i}.beta"].numpy().astype(np.float64) - m * sc
        row = np.zeros(4, np.int64)
        for b in range(Bz):          # clip by clip: bounded memory
            y = z[b] * sc + sh
            y = y.reshape(H // pool[0], pool[0], Wd // pool[1], pool[1], C).transpose(0, 2, 4, 1, 3).reshape(-1, pool[0] * pool[1])
            y.partition(y.shape[1] - 2, axis=1) if y.shape[1] > 1 else None
            t1 = y[:, -1]
            t2 = y[:, -2] if y.shape[1] > 1 else np.full_like(t1, -np.inf)
            for j, d in enumerate((1e-6, 1e-5)):
                row[j] += int(((t1 > 0) & (t1 - t2 < d)).sum())
                row[2 + j] += int((np.abs(t1) < d).sum())
        rows.append(row)
        del z
    return np.array(rows)


def main():
    spec = O.Spec.from_config(SELDNET_CONFIG)
    tr, _ = O.variable_specs(spec)
    for dl in ("MSE", "MMSE"):
        w, st = O.random_weights(spec, 0)
        x, ys, yd = O.synthetic_batch(B, T, seed=1234)
        kw = dict(doa_loss=dl, loss_weight=(1.0, 1000.0), lr=1e-3, step=1)
        rec = {}
        r = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float64, want_taps=(dl == "MSE"), record_routing=rec, **kw)
        out = {"meta": np.array([B, T, {"MSE": 0, "MMSE": 1}[dl]])}
        import importlib.util
        sp_ = importlib.util.spec_from_file_location("make_golden_blocks", os.path.join(ROOT, "tests", "golden", "make_golden_blocks.py"))
        mb = importlib.util.module_from_spec(sp_)
        sp_.loader.exec_module(mb)
        if dl == "MSE":
            out["near_ties"] = near_ties(spec, r.pop("taps"), O.unflatten(torch.as_tensor(w), tr))
        rec32 = {}
        g32 = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float32, record_routing=rec32, **kw)["grad"]
        # round 5: the fp32 oracle ON the fp64 decisions — fp32 arithmetic alone, no decision taken differently (`bar_fp32_given`)
        routing64 = {i: (rec[i]["pos"].clone(), rec[i]["gate"].clone()) for i in range(len(spec.pools))}
        r32g = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float32, routing=routing64, **kw)
        g32g = r32g["grad"]
        out["out_err_fp32_given"] = np.array([np.abs(r32g["sed"] - r["sed"]).max() / np.abs(r["sed"]).max(),
                                              np.abs(r32g["doa"] - r["doa"]).max() / np.abs(r["doa"]).max()])
        del routing64
        for i in range(len(spec.pools)):
            rr, r32 = rec.pop(i), rec32.pop(i)
            rr.pop("windows", None)
            r32.pop("windows", None)
            val, margin = mb.pool_decisions(rr)
            # round 5: the margin follows make_golden_blocks.margin_rule (8 x the fp32 oracle's own error on the window maxima, clamped to
            # [1e-5, 2e-3]) instead of a fixed 1e-5: the deeper blocks' maxima carry the summation error of everything in front of them
            err = float(np.abs(r32["top"].numpy().astype(np.float64) - rr["top"].numpy()).max())
            eps = mb.margin_rule(err)
            v32, _ = mb.pool_decisions(r32)
            diff = v32.reshape(-1) != val.reshape(-1)
            near = np.flatnonzero(margin.reshape(-1) < eps).astype(np.int64)
            out[f"dec.pool{i}.near"] = near.astype(np.uint32)
            out[f"dec.pool{i}.near_val"] = val.reshape(-1)[near].astype(np.uint8)      # the fp64 decisions AT the near-ties (round 4: injected by the test)
            out[f"dec.pool{i}.digest"] = mb.decision_digest(val, near)
            out[f"dec.pool{i}.eps"], out[f"dec.pool{i}.err32"] = np.float64(eps), np.float64(err)
            print(f"  block {i}: {val.size} routing decisions, fp32 error on the window maxima {err:.2e} -> eps {eps:.1e}: {near.size} near; the fp32 "
                  f"oracle flips {int(diff.sum())} (largest fp64 margin {float(margin.reshape(-1)[diff].max()) if diff.any() else 0.0:.2e}, "
                  f"{int((diff & (margin.reshape(-1) >= eps)).sum())} outside eps)", flush=True)
            del rr, r32, val, margin, v32
        off = 0
        bars, norms, maxes, bars_g, nbars_g = [], [], [], [], []
        for name, shape in tr:
            k = int(np.prod(shape))
            g = r["grad"][off:off + k]
            out["g." + name] = g[sample_index(name, k)]
            norms.append(np.linalg.norm(g))
            maxes.append(np.abs(g).max())
            bars.append(np.abs(g32[off:off + k].astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-300))
            bars_g.append(np.abs(g32g[off:off + k].astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-300))
            nbars_g.append(abs(np.linalg.norm(g32g[off:off + k].astype(np.float64)) - norms[-1]) / max(norms[-1], 1e-300))
            off += k
        out["grad_norms"], out["grad_max"], out["bar_fp32"] = np.array(norms), np.array(maxes), np.array(bars)
        out["bar_fp32_given"], out["norm_bar_fp32_given"] = np.array(bars_g), np.array(nbars_g)
        out["new_w_err_fp32_given"] = np.abs(r32g["new_w"].astype(np.float64) - r["new_w"]).max()
        out["sed"] = r["sed"].reshape(-1)[out_sample_index(r["sed"].size)]
        out["doa"] = r["doa"].reshape(-1)[out_sample_index(r["doa"].size)]
        out["sloss"] = r["sloss"]
        dlv = np.asarray(r["dloss"]).reshape(-1)
        out["dloss"] = dlv[out_sample_index(dlv.size)]
        out["dloss_sum"] = dlv.sum()
        out["new_state"] = r["new_state"]
        out["new_w"] = r["new_w"][out_sample_index(r["new_w"].size)]
        # samples as float32 (6e-8 relative: far below the 1e-4 bar), scalars / norms / bars as float64: ~100 KB per case
        f64 = ("grad_norms", "grad_max", "bar_fp32", "sloss", "dloss_sum", "bar_fp32_given", "norm_bar_fp32_given", "out_err_fp32_given", "new_w_err_fp32_given")
        out = {k: (v if k in ("meta", "near_ties") or k.startswith("dec.") else np.asarray(v, np.float64 if k in f64 else np.float32)) for k, v in out.items()}
        path = os.path.join(ROOT, "tests", "golden", f"seldnet_full_b32_t3000_{dl.lower()}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes")
        if "near_ties" in out:
            print("  near-tie windows per conv block [gap<1e-6, gap<1e-5, |top|<1e-6, |top|<1e-5]:", out["near_ties"].tolist())
        for (name, _), b, bg in zip(tr, bars, bars_g):
            print("  %-28s fp32-oracle bar %.3e   given the fp64 decisions %.3e" % (name, b, bg))


if __name__ == "__main__":
    main()
