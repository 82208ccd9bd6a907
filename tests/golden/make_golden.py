"""Generates tests/golden/seldnet_*.npz from the CPU oracle (fp64 arithmetic, stored as fp32).

Run from the repo root:  python tests/golden/make_golden.py
The reference cannot be imported in the build container (TensorFlow absent), so
these vectors come from the oracle restatement and pin IT against drift; the HIP path is checked
against the same files on the GPU box.  Inputs/weights are regenerated from seeds by the test
(numpy default_rng is platform-stable), so only outputs are stored: KB-sized files."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import seldnet_oracle as O  # noqa: E402
from __graft_entry__ import SELDNET_CONFIG  # noqa: E402

CASES = [("b2_t50_mse", 2, 50, "MSE"), ("b2_t50_mmse", 2, 50, "MMSE"), ("b3_t100_mse", 3, 100, "MSE")]
SAMPLE = 997  # stride of the stored gradient / weight samples


def main():
    spec = O.Spec.from_config(SELDNET_CONFIG)
    for name, B, T, dl in CASES:
        w, st = O.random_weights(spec, 0)
        x, ys, yd = O.synthetic_batch(B, T, seed=1234)
        r = O.train_step(spec, w, st, x, ys, yd, doa_loss=dl, loss_weight=(1.0, 1000.0), lr=1e-3, step=1, dtype=torch.float64)
        t = O.test_step(spec, w, st, x, ys, yd, dl, dtype=torch.float64)
        tr, _ = O.variable_specs(spec)
        norms, off = [], 0
        for _, shape in tr:
            k = int(np.prod(shape))
            norms.append(np.linalg.norm(r["grad"][off:off + k]))
            off += k
        out = {
            "meta": np.array([B, T, {"MSE": 0, "MMSE": 1}[dl]]),
            "train_sed": r["sed"], "train_doa": r["doa"], "train_sloss": r["sloss"], "train_dloss": r["dloss"],
            "grad_sample": r["grad"][::SAMPLE], "grad_norms": np.array(norms), "new_state": r["new_state"],
            "new_w_sample": r["new_w"][::SAMPLE],
            "test_sed": t["sed"], "test_doa": t["doa"], "test_sloss": t["sloss"], "test_dloss": t["dloss"],
        }
        out = {k: np.asarray(v, np.float32) if k != "meta" else v for k, v in out.items()}
        path = os.path.join(ROOT, "tests", "golden", f"seldnet_{name}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
