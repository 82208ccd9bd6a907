#!/usr/bin/env python3
"""Keras HDF5 checkpoint <-> seld_amd .npz (reference seams: train.py:372-380 `save_model(model, 'bestscore_*.hdf5',
include_optimizer=False)`, train.py:322-331 `--resume`, evaluator.py:57 `model.load_weights`).

    python tools/keras_h5_to_npz.py bestscore_0.344.hdf5 seldnet.npz          # reference checkpoint -> SeldNet.load_weights
    python tools/keras_h5_to_npz.py --to-h5 seldnet.npz template.hdf5 out.hdf5   # our weights into a copy of a Keras file

Needs h5py, which the build image lacks: run it where the reference runs.  The name mapping itself
(`map_keras_variables`) is a pure function of the Keras variable NAMES and SHAPES and is unit-tested without h5py
(tests/test_host_logic_cpu.py).

Keras names its layers per type in creation order ('conv2d', 'conv2d_1', ...; models.py:24-30 builds FIRST, SECOND, the SED block,
`sed_out`, the DOA block, `doa_out`), so for model_config/seldnet.json:
    conv2d[_i]/kernel, bias                         -> conv{i}.kernel, conv{i}.bias               (HWIO, as stored here)
    batch_normalization[_i]/gamma, beta, moving_*   -> bn{i}.gamma, .beta, .moving_mean, .moving_variance
    bidirectional[_i]/forward_*/gru_cell*/{kernel,recurrent_kernel,bias}   -> gru{i}.fwd.*       ([in,3u] z|r|h, bias [2,3u])
    bidirectional[_i]/backward_*/...                                       -> gru{i}.bwd.*
    conv1d[_j]/kernel, bias                         -> sed.dense{j}.* for the first len(SED units) layers, then doa.dense{j}.*
    sed_out/kernel, bias ; doa_out/kernel, bias     -> sed.out.*, doa.out.*"""
from __future__ import annotations

import re
import sys
from typing import Dict, List, Sequence, Tuple

import numpy as np


def _layer_index(layer: str, base: str):
    """'conv2d' -> 0, 'conv2d_3' -> 3 for base 'conv2d'; None if `layer` is not of that type."""
    m = re.fullmatch(re.escape(base) + r"(?:_(\d+))?", layer)
    return None if m is None else int(m.group(1) or 0)


def map_keras_variables(keras_vars: Sequence[Tuple[str, Tuple[int, ...]]], n_conv: int = 3, n_gru: int = 2,
                        n_sed_dense: int = 1, n_doa_dense: int = 1) -> Dict[str, str]:
    """{our variable name: keras variable name} for the seldnet.json architecture.
    keras_vars: (name, shape) of every variable in the file, names as Keras stores them ('conv2d_1/kernel:0',
    'bidirectional/forward_gru/gru_cell_1/kernel:0', ...).  Raises ValueError when a layer is missing or ambiguous."""
    by_layer: Dict[str, List[str]] = {}
    for name, _ in keras_vars:
        by_layer.setdefault(name.split("/")[0], []).append(name)

    def ranked(base):       # layers of one type in creation order (Keras' numeric suffix), re-based at 0
        found = sorted((i, l) for l in by_layer for i in [_layer_index(l, base)] if i is not None)
        return [l for _, l in found]

    def pick(layer, leaf, must_contain=None):
        c = [n for n in by_layer[layer] if n.split("/")[-1].split(":")[0] == leaf and (must_contain is None or must_contain in n)]
        if len(c) != 1:
            raise ValueError(f"expected exactly one '{leaf}' in Keras layer '{layer}' (filter {must_contain!r}), found {c}")
        return c[0]

    out: Dict[str, str] = {}
    convs, bns, bis, c1d = ranked("conv2d"), ranked("batch_normalization"), ranked("bidirectional"), ranked("conv1d")
    if len(convs) != n_conv or len(bns) != n_conv or len(bis) != n_gru or len(c1d) != n_sed_dense + n_doa_dense:
        raise ValueError(f"layer census does not match the architecture: conv2d {convs}, batch_normalization {bns}, "
                         f"bidirectional {bis}, conv1d {c1d}")
    for i in range(n_conv):
        out[f"conv{i}.kernel"], out[f"conv{i}.bias"] = pick(convs[i], "kernel"), pick(convs[i], "bias")
        for leaf in ("gamma", "beta", "moving_mean", "moving_variance"):
            out[f"bn{i}.{leaf}"] = pick(bns[i], leaf)
    for i in range(n_gru):
        for d, tag in (("fwd", "forward"), ("bwd", "backward")):
            for leaf in ("kernel", "recurrent_kernel", "bias"):
                out[f"gru{i}.{d}.{leaf}"] = pick(bis[i], leaf, must_contain="/" + tag)
    for j in range(n_sed_dense):
        out[f"sed.dense{j}.kernel"], out[f"sed.dense{j}.bias"] = pick(c1d[j], "kernel"), pick(c1d[j], "bias")
    for j in range(n_doa_dense):
        out[f"doa.dense{j}.kernel"], out[f"doa.dense{j}.bias"] = pick(c1d[n_sed_dense + j], "kernel"), pick(c1d[n_sed_dense + j], "bias")
    for head in ("sed", "doa"):
        if f"{head}_out" not in by_layer:
            raise ValueError(f"Keras layer '{head}_out' (models.py:28-30) not found")
        out[f"{head}.out.kernel"], out[f"{head}.out.bias"] = pick(f"{head}_out", "kernel"), pick(f"{head}_out", "bias")
    return out


def check_shapes(mapping: Dict[str, str], keras_vars, our_vars) -> None:
    """our_vars: {name: shape} (SeldNet.variables + state_variables).  Keras Conv1D kernels are [1,in,out] like ours."""
    ks = dict(keras_vars)
    for ours, theirs in mapping.items():
        if tuple(ks[theirs]) != tuple(our_vars[ours]):
            raise ValueError(f"{theirs} has shape {tuple(ks[theirs])}, {ours} expects {tuple(our_vars[ours])}")
    missing = set(our_vars) - set(mapping)
    if missing:
        raise ValueError(f"no Keras variable for {sorted(missing)}")


def _walk_h5(f):
    """[(variable name, h5 dataset)] of a Keras file: `save_model` keeps the weights under 'model_weights', `save_weights` at the root."""
    import h5py
    root = f["model_weights"] if "model_weights" in f else f
    out = []

    def visit(name, obj):
        if isinstance(obj, h5py.Dataset):
            # 'conv2d/conv2d/kernel:0' -> 'conv2d/kernel:0' (Keras repeats the layer name as the group)
            parts = name.split("/")
            out.append(("/".join(parts[1:]) if len(parts) > 2 and parts[0] == parts[1] else name, obj))
    root.visititems(visit)
    return out


def our_variable_shapes(in_ch: int = 7, n_classes: int = 12) -> Dict[str, Tuple[int, ...]]:
    """Names and shapes of model_config/seldnet.json's variables (the C library's layout, seld_variable_info), restated so that
    the converter runs without a GPU."""
    v: Dict[str, Tuple[int, ...]] = {}
    cin = in_ch
    for i in range(3):
        v[f"conv{i}.kernel"], v[f"conv{i}.bias"] = (3, 3, cin, 64), (64,)
        for leaf in ("gamma", "beta", "moving_mean", "moving_variance"):
            v[f"bn{i}.{leaf}"] = (64,)
        cin = 64
    for i in range(2):
        for d in ("fwd", "bwd"):
            v[f"gru{i}.{d}.kernel"], v[f"gru{i}.{d}.recurrent_kernel"], v[f"gru{i}.{d}.bias"] = (128, 384), (128, 384), (2, 384)
    for head, n in (("sed", n_classes), ("doa", 3 * n_classes)):
        v[f"{head}.dense0.kernel"], v[f"{head}.dense0.bias"] = (1, 128, 128), (128,)
        v[f"{head}.out.kernel"], v[f"{head}.out.bias"] = (128, n), (n,)
    return v


def h5_to_npz(h5_path: str, npz_path: str, in_ch: int = 7, n_classes: int = 12) -> None:
    import h5py
    with h5py.File(h5_path, "r") as f:
        items = _walk_h5(f)
        kv = [(n, tuple(d.shape)) for n, d in items]
        mapping = map_keras_variables(kv)
        ours = our_variable_shapes(in_ch, n_classes)
        check_shapes(mapping, kv, ours)
        data = dict(items)
        np.savez(npz_path, **{o: np.asarray(data[k], np.float32) for o, k in mapping.items()})
    print(f"{h5_path}: {len(mapping)} variables -> {npz_path}")


def npz_to_h5(npz_path: str, template_h5: str, out_h5: str) -> None:
    """Overwrite the datasets of a COPY of `template_h5` (any checkpoint of the same architecture) with our weights."""
    import shutil
    import h5py
    shutil.copyfile(template_h5, out_h5)
    z = np.load(npz_path)
    with h5py.File(out_h5, "r+") as f:
        items = _walk_h5(f)
        mapping = map_keras_variables([(n, tuple(d.shape)) for n, d in items])
        data = dict(items)
        for o, k in mapping.items():
            data[k][...] = np.asarray(z[o], np.float32).reshape(data[k].shape)
    print(f"{npz_path} -> {out_h5} ({len(mapping)} variables, structure of {template_h5})")


if __name__ == "__main__":
    a = sys.argv[1:]
    if len(a) == 4 and a[0] == "--to-h5":
        npz_to_h5(a[1], a[2], a[3])
    elif len(a) == 2:
        h5_to_npz(a[0], a[1])
    else:
        sys.exit(__doc__)
