"""Generates tests/golden/{xception_gru_full_b32,resnet50_gru_full_b16}_t3000_mse.npz: BASELINE.json configs[3] and configs[4]
(model_config/xception_gru.json:2-11 at 32 clips, resnet50_gru.json:2-11 at the 16 clips one GPU of the DP-8 job holds; both
[3000,64,7]) evaluated once by the CPU oracle in fp64, in the build container.  The blocks themselves are this repository's
specs (spec/XCEPTION_BLOCK.md, spec/RESNET50_BLOCK.md): the reference names them and does not define them, so these vectors pin
the oracle restatement of OUR spec — "parity unpinned" in the sense of DESIGN.md section 0.

Run from the repo root (xception: ~30 GB / 8 min, resnet50: ~25 GB / 6 min on 8 cores):
    python tests/golden/make_golden_blocks.py [xception_gru|resnet50_gru]

Stored per model (inputs and weights are regenerated from seeds by the test):
  * outputs / losses / BN state / post-Adam weights: strided samples as make_golden_full.py stores them;
  * per trainable variable: a strided gradient sample (<= 512 elements), l2 norm, max |.|, and `bar_fp32` = the error of the SAME
    oracle evaluated in fp32 (what two evaluations of the reference's own arithmetic differ by at this size);
  * DECISIONS, so that a test can assert "every decision the library takes differently from fp64 is one fp32 cannot resolve"
    without an oracle on the GPU box.  A decision tensor is the first block's MaxPool(ReLU(.)) routing (value = 0 if the window
    passes 0, else 1 + argmax position), one of resnet50_block's 48 ReLU gates (value = gate), or — round 4 — one of xception_block's 24
    unit-input ReLU gates / its exit MaxPool(ReLU(.)) routing.  For each: `eps` (the margin
    below which a decision counts as unresolvable — `margin_rule`: 8 x the fp32 oracle's own error on the value the decision is taken on
    (`err32`, stored beside it), clamped to [1e-5, 2e-3]; since round 5 for the pooling routings too, which used a fixed 1e-5 and so
    made the fixtures sensitive to the summation order of the first convolution), `near` = the flat indices whose fp64 margin is below eps, `near_val` = the fp64
    decision at each of them (round 4: injected by the test through seld_debug_set_routing / _set_relu_gates), and `digest` = (count,
    position-weighted checksum mod 2^64) of the fp64 decisions over all OTHER indices (`decision_digest`).  The library's
    decisions, digested with the same `near` indices excluded, must give the same pair: then every differing decision lies in
    `near`, i.e. has an fp64 margin below eps;
  * `fp32_flips`: how many decisions the fp32 oracle takes differently, and the largest fp64 margin among them (for the record)."""
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import seldnet_oracle as O  # noqa: E402
from __graft_entry__ import SELDNET_CONFIG  # noqa: E402

T = int(os.environ.get("GOLDEN_T", 3000))
MAX_SAMPLE = 512
MODELS = {
    "xception_gru": (32, "xception_block", {"filters": 32, "block_num": 8, "kernel_regularizer": {"l1": 0, "l2": 1e-3}}),
    "resnet50_gru": (16, "resnet50_block", {"filters": 32, "block_num": [3, 4, 6, 3], "kernel_regularizer": {"l1": 0, "l2": 1e-3}}),
}
_K = np.uint64(0x9E3779B97F4A7C15)


def model_config(which: str) -> dict:
    cfg = copy.deepcopy(SELDNET_CONFIG)
    cfg["FIRST"], cfg["FIRST_ARGS"] = MODELS[which][1], copy.deepcopy(MODELS[which][2])
    return cfg


def fixture_path(which: str) -> str:
    return os.path.join(ROOT, "tests", "golden", f"{which}_full_b{MODELS[which][0]}_t{T}_mse.npz")


def sample_index(n: int) -> np.ndarray:
    return np.arange(n) if n <= MAX_SAMPLE else np.linspace(0, n - 1, MAX_SAMPLE).astype(np.int64)


def out_sample_index(n: int, k: int = 4096) -> np.ndarray:
    return np.linspace(0, n - 1, min(n, k)).astype(np.int64)


def decision_digest(values: np.ndarray, near: np.ndarray) -> np.ndarray:
    """(count of non-zero decisions, sum of value[i] * (i + 1) * K mod 2^64) over all flat indices i NOT in `near`.
    `values`: small non-negative integers (0 / 1 gates, or 0 = window passes 0, 1 + argmax position otherwise)."""
    v = np.ascontiguousarray(values).reshape(-1).astype(np.uint64)
    if near.size:
        v[near] = 0
    nz = np.flatnonzero(v)
    with np.errstate(over="ignore"):
        chk = ((nz.astype(np.uint64) + np.uint64(1)) * _K * v[nz]).sum(dtype=np.uint64)
    return np.array([nz.size, chk], dtype=np.uint64)


EPS_MIN, EPS_MAX, EPS_FACTOR = 1e-5, 2e-3, 8.0


def margin_rule(err32: float) -> float:
    """The near-tie margin of ONE decision tensor (round 5: the same rule for MaxPool(ReLU) routings and for ReLU gates): a decision whose
    fp64 margin is below EPS_FACTOR x the fp32 oracle's own error on the value it is taken on (`err32`: max |pre-activation_fp32 -
    pre-activation_fp64| for a gate; max |window maximum_fp32 - window maximum_fp64| for a pooling window) is one that ANY fp32 evaluation —
    any summation order of the products in front of it — may take either way; clamped to [1e-5, 2e-3].  The fixtures store `err32` beside
    `eps`: tests/test_oracle_pins.py::test_fixture_margins_follow_the_rule holds every stored eps to this function."""
    return float(min(max(EPS_FACTOR * err32, EPS_MIN), EPS_MAX))


def pool_decisions(rec):
    """record_routing[0] of the oracle -> (value tensor, margin tensor): margin = top1 - top2 where the window passes,
    and |top1| everywhere (whichever is smaller decides)."""
    gate = rec["gate"].numpy()
    val = np.where(gate, rec["pos"].numpy() + 1, 0)
    top = rec["top"].numpy()
    margin = np.minimum(np.abs(top), np.where(gate, rec["gap"].numpy(), np.inf))
    return val, margin


def main(which: str):
    Bn, first, _ = MODELS[which]
    cfg = model_config(which)
    spec = O.Spec.from_config(cfg)
    tr, _ = O.variable_specs(spec)
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(Bn, T, seed=1234)
    kw = dict(doa_loss="MSE", loss_weight=(1.0, 1000.0), lr=1e-3, step=1)
    t0 = time.time()
    rec64 = {}
    r = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float64, record_routing=rec64, **kw)
    print(f"{which}: fp64 oracle step {time.time() - t0:.0f} s", flush=True)
    for v in rec64.values():
        v.pop("windows", None)
    out = {"meta": np.array([Bn, T, 0])}
    # ---- decisions of the fp64 evaluation
    dec64, top64 = {}, {}
    routing64 = {}      # every fp64 decision, in the form the oracle takes GIVEN decisions in (round 5: `bar_fp32_given`)
    v0 = rec64.pop(0)
    val, margin = pool_decisions(v0)
    dec64["pool0"], top64["pool0"] = (val, margin, None), v0["top"].numpy().astype(np.float64)
    routing64[0] = (v0["pos"].clone(), v0["gate"].clone())
    for key in list(rec64.keys()):
        v = rec64.pop(key)
        routing64[key] = (v["pos"].clone(), v["gate"].clone()) if "pos" in v else v["gate"].clone()
        if "pos" in v:      # a MaxPool(ReLU(.)) routing (xception_block's exit)
            val, margin = pool_decisions(v)
            dec64[key], top64[key] = (val, margin, None), v["top"].numpy().astype(np.float64)
        else:               # a ReLU gate (resnet50_block's 48; xception_block's 24 unit inputs)
            dec64[key] = (v["gate"].numpy(), np.abs(v["pre"].numpy()), v["pre"].numpy())
    t0 = time.time()
    rec32 = {}
    r32 = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float32, record_routing=rec32, **kw)
    print(f"{which}: fp32 oracle step {time.time() - t0:.0f} s", flush=True)
    g32 = r32["grad"]
    names, flips = [], []
    for key, (val, margin, pre64) in dec64.items():
        if pre64 is None:
            rr = rec32.pop(0 if key == "pool0" else key)
            rr.pop("windows", None)
            v32, _ = pool_decisions(rr)
            err = float(np.abs(rr["top"].numpy().astype(np.float64) - top64[key]).max())
        else:
            rr = rec32.pop(key)
            v32 = rr["gate"].numpy()
            err = float(np.abs(rr["pre"].numpy().astype(np.float64) - pre64).max())
        eps = margin_rule(err)
        near = np.flatnonzero(margin.reshape(-1) < eps).astype(np.int64)
        diff = v32.reshape(-1) != val.reshape(-1)
        worst = float(margin.reshape(-1)[diff].max()) if diff.any() else 0.0
        k = key.replace(".", "_")
        out[f"dec.{k}.near"] = near.astype(np.uint32)
        out[f"dec.{k}.near_val"] = val.reshape(-1)[near].astype(np.uint8)      # the fp64 decisions AT the near-ties (round 4: injected by the test)
        out[f"dec.{k}.digest"] = decision_digest(val, near)
        out[f"dec.{k}.eps"] = np.float64(eps)
        out[f"dec.{k}.err32"] = np.float64(err)
        names.append(k)
        flips.append((int(diff.sum()), worst, int((diff & (margin.reshape(-1) >= eps)).sum())))
        print(f"  {key:14s} {val.size:10d} decisions, eps {eps:.1e}: {near.size:7d} near, fp32 oracle flips {int(diff.sum()):5d} "
              f"(largest fp64 margin {worst:.2e}, {flips[-1][2]} outside eps)", flush=True)
        del rr, v32
    out["dec_names"] = np.array(names)
    out["fp32_flips"] = np.array(flips, np.float64)
    # ---- the fp32 oracle evaluated ON the fp64 decisions (round 5): what fp32 ARITHMETIC alone — no decision taken differently — puts between
    # an fp32 evaluation of this network and the fp64 one.  tests/test_model_gpu.py::test_full_size_parity_given_fp64_decisions holds the
    # library, given the same decisions, to max(1e-4, 1.5 x this) per variable instead of a blanket bar
    del rec32, dec64, top64
    t0 = time.time()
    r32g = O.train_step(spec, w, st, x, ys, yd, dtype=torch.float32, routing=routing64, **kw)
    print(f"{which}: fp32 oracle step given the fp64 decisions {time.time() - t0:.0f} s", flush=True)
    del routing64
    g32g = r32g["grad"]
    out["out_err_fp32_given"] = np.array([np.abs(r32g["sed"] - r["sed"]).max() / np.abs(r["sed"]).max(),
                                          np.abs(r32g["doa"] - r["doa"]).max() / np.abs(r["doa"]).max()])
    # ---- gradients
    off = 0
    bars, norms, maxes, bars_g, nbars_g = [], [], [], [], []
    for name, shape in tr:
        k = int(np.prod(shape))
        g = r["grad"][off:off + k]
        out["g." + name] = g[sample_index(k)]
        norms.append(np.linalg.norm(g))
        maxes.append(np.abs(g).max())
        bars.append(np.abs(g32[off:off + k].astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-300))
        bars_g.append(np.abs(g32g[off:off + k].astype(np.float64) - g).max() / max(np.abs(g).max(), 1e-300))
        nbars_g.append(abs(np.linalg.norm(g32g[off:off + k].astype(np.float64)) - norms[-1]) / max(norms[-1], 1e-300))
        off += k
    out["grad_norms"], out["grad_max"], out["bar_fp32"] = np.array(norms), np.array(maxes), np.array(bars)
    out["bar_fp32_given"], out["norm_bar_fp32_given"] = np.array(bars_g), np.array(nbars_g)
    out["new_w_err_fp32_given"] = np.abs(r32g["new_w"].astype(np.float64) - r["new_w"]).max()
    out["state_err_fp32_given"] = np.abs(r32g["new_state"].astype(np.float64) - r["new_state"]).max() / np.abs(r["new_state"]).max()
    out["sed"] = r["sed"].reshape(-1)[out_sample_index(r["sed"].size)]
    out["doa"] = r["doa"].reshape(-1)[out_sample_index(r["doa"].size)]
    out["sloss"] = r["sloss"]
    dlv = np.asarray(r["dloss"]).reshape(-1)
    out["dloss"] = dlv[out_sample_index(dlv.size)]
    out["dloss_sum"] = dlv.sum()
    out["new_state"] = r["new_state"]
    out["new_w"] = r["new_w"][out_sample_index(r["new_w"].size)]
    out["out_err_fp32"] = np.array([np.abs(r32["sed"] - r["sed"]).max() / np.abs(r["sed"]).max(),
                                    np.abs(r32["doa"] - r["doa"]).max() / np.abs(r["doa"]).max()])
    f64 = ("grad_norms", "grad_max", "bar_fp32", "sloss", "dloss_sum", "fp32_flips", "out_err_fp32", "bar_fp32_given", "norm_bar_fp32_given", "out_err_fp32_given", "new_w_err_fp32_given", "state_err_fp32_given")
    keep = ("meta", "dec_names")
    out = {k: (v if k in keep or k.startswith("dec.") else np.asarray(v, np.float64 if k in f64 else np.float32)) for k, v in out.items()}
    path = fixture_path(which)
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")
    print("  fp32 oracle vs fp64 oracle at the outputs (sed, doa):", out["out_err_fp32"].tolist())
    order = np.argsort(-np.array(bars))[:12]
    for i in order:
        print("  %-28s fp32-oracle bar %.3e" % (tr[i][0], bars[i]))
    print("  fp32 oracle GIVEN the fp64 decisions vs the fp64 oracle: outputs", out["out_err_fp32_given"].tolist())
    for i in np.argsort(-np.array(bars_g))[:12]:
        print("  %-28s fp32-oracle-given-decisions bar %.3e (norm %.3e)" % (tr[i][0], bars_g[i], nbars_g[i]))


if __name__ == "__main__":
    for which_ in (sys.argv[1:] or list(MODELS)):
        main(which_)
