"""CPU restatement (PyTorch, any dtype) of the reference's configurable 2-D blocks — TEST INFRASTRUCTURE ONLY: imported by tests/ (and
nothing in the product path; seld_amd/modules.py is the HIP-backed counterpart and never touches this file).

Restated, line by line:
  * modules.mother_block  (/root/reference/modules.py:184-298): up to three Conv2D(k, 'same') + BatchNormalization layers (strides on the
    second), each optionally summed with identity / Conv2D(1x1)+BatchNormalization-projected skips of the block input and of the
    earlier layers, Activation; a skipped layer concatenates the tensors its `connect` list names (1x1 strided convolutions where the
    extents differ); squeeze-and-excitation tail (reduce_mean, Conv2D(se_filters, 1, se_activation), Conv2D(C, 1, sigmoid), product);
    the ValueErrors of modules.py:202-222;
  * modules.mother_stage  (/root/reference/modules.py:15-43): `depth` mother_blocks, the strides applied in the first only;
  * models.seldnet around them (models.py:18-32): FIRST -> bidirectional_GRU_block -> simple_dense_block heads, via seldnet_oracle's
    restatements of those blocks.
Keras defaults the reference relies on: Conv2D use_bias=True, glorot_uniform / zeros; BatchNormalization eps 1e-3, momentum 0.99
(seldnet_oracle.batchnorm); 'same' padding is TensorFlow's (out = ceil(in / stride), the extra padding element at the END).
Pinned by the reference's own known answers for these blocks: the output SHAPES of modules_test.py:8-28 (mother_stage) and :154-200
(mother_block, with and without squeeze-excite) — tests/test_modules_cpu.py.  Numeric values: parity unpinned (TensorFlow is absent),
as for everything else in oracle/ (DESIGN.md section 0).
Variable order = the order the reference's code creates the Keras layers in (what tf.keras.Model.trainable_variables enumerates)."""
from __future__ import annotations

import copy
import math
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import seldnet_oracle as O

ACTS = {None: lambda t: t, "linear": lambda t: t, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
        "swish": lambda t: t * torch.sigmoid(t)}


def safe_tuple(v, n=2):
    """utils.safe_tuple: an int becomes (v, v)."""
    return tuple(int(a) for a in v) if isinstance(v, (list, tuple)) else (int(v),) * n


def same_out(n: int, s: int) -> int:
    return -(-n // s)


def conv2d_same(x, kernel, bias, strides=(1, 1)):
    """tf.keras.layers.Conv2D(filters, k, strides, padding='same') on NHWC x, HWIO kernel."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    H, W = x.shape[1], x.shape[2]
    ph = max((same_out(H, strides[0]) - 1) * strides[0] + kh - H, 0)
    pw = max((same_out(W, strides[1]) - 1) * strides[1] + kw - W, 0)
    xp = F.pad(x.permute(0, 3, 1, 2), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    y = F.conv2d(xp, kernel.permute(3, 2, 0, 1), bias, stride=strides)
    return y.permute(0, 2, 3, 1)


def check_mother_config(cfg: dict) -> None:
    """the ValueErrors of modules.py:202-222"""
    f = [cfg[f"filters{i}"] for i in range(3)]
    k = [cfg[f"kernel_size{i}"] for i in range(3)]
    c0, c1, c2 = cfg["connect0"], cfg["connect1"], cfg["connect2"]
    strides = safe_tuple(cfg.get("strides", (1, 1)))
    for i in range(3):
        if (f[i] == 0) != (k[i] == 0):
            raise ValueError(f"{i}) skipped layer must have 0 filters, 0 kernel size")
    if f[0] == 0 and max(c1[1], c2[1]):
        raise ValueError("cannot link skipped layer (first layer)")
    if f[1] == 0 and c2[2] > 0:
        raise ValueError("cannot link skipped layer (second layer)")
    if (f[0] != 0) + sum(c0) == 0:
        raise ValueError("cannot pass zero inputs to the second layer")
    if (f[1] != 0) + sum(c1) == 0:
        raise ValueError("cannot pass zero inputs to the third layer")
    if (f[2] != 0) + sum(c2) == 0:
        raise ValueError("cannot pass zero inputs to the final output")
    if f[1] == 0 and tuple(strides) != (1, 1):
        raise ValueError("if strides are set, the second layer must be active")


def dead_layers(cfg: dict) -> Tuple[bool, bool]:
    """(first layer dead, second layer dead): a layer whose output reaches neither the block's output nor a later layer.  The reference's checks
    (modules.py:202-222) accept such configurations (e.g. filters1 > 0, filters2 = 0, connect2 = [1, 0, 0]); its Keras functional model then simply
    does not contain the layer — no variables, no computation — and neither does this restatement."""
    f = [int(cfg[f"filters{i}"]) for i in range(3)]
    c1, c2 = cfg["connect1"], cfg["connect2"]
    need2 = f[2] > 0 or c2[2] == 1                                  # outputs[2] (the second layer's) reaches the third layer / the output
    need1 = c2[1] == 1 or (need2 and (f[1] > 0 or c1[1] == 1))      # outputs[1] (the first layer's)
    return (f[0] > 0 and not need1), (not need2)


def mother_block_plan(cfg: dict, in_shape: Tuple[int, int, int], prefix: str):
    """-> (trainable [(name, shape)], state [(name, shape)], out_shape (H, W, C)) in Keras creation order."""
    check_mother_config(cfg)
    tr: List[Tuple[str, Tuple[int, ...]]] = []
    nt: List[Tuple[str, Tuple[int, ...]]] = []

    def conv(name, k, cin, cout):
        tr.append((f"{prefix}.{name}.kernel", (k, k, cin, cout)))
        tr.append((f"{prefix}.{name}.bias", (cout,)))

    def bn(name, c):
        tr.append((f"{prefix}.{name}.gamma", (c,)))
        tr.append((f"{prefix}.{name}.beta", (c,)))
        nt.append((f"{prefix}.{name}.moving_mean", (c,)))
        nt.append((f"{prefix}.{name}.moving_variance", (c,)))

    f = [int(cfg[f"filters{i}"]) for i in range(3)]
    k = [int(cfg[f"kernel_size{i}"]) for i in range(3)]
    conn = [cfg["connect0"], cfg["connect1"], cfg["connect2"]]
    strides = safe_tuple(cfg.get("strides", (1, 1)))
    shapes = [tuple(in_shape)]       # outputs[i]
    dead0, dead1 = dead_layers(cfg)
    real_conv, real_bn = conv, bn
    # first layer
    if dead0:
        conv = bn = lambda *a: None       # a layer whose output reaches nothing has no variables (dead_layers)
    if f[0] > 0:
        conv("c0", k[0], shapes[-1][2], f[0]); bn("bn0", f[0])
        out = (shapes[-1][0], shapes[-1][1], f[0])
        if conn[0][0] == 1 and shapes[-1] != out:
            conv("p0_0", 1, shapes[-1][2], f[0]); bn("pbn0_0", f[0])
    else:
        out = shapes[-1]
    shapes.append(out)
    conv, bn = (lambda *a: None, lambda *a: None) if dead1 else (real_conv, real_bn)
    # second layer (strides)
    if f[1] > 0:
        conv("c1", k[1], shapes[-1][2], f[1]); bn("bn1", f[1])
        out = (same_out(shapes[-1][0], strides[0]), same_out(shapes[-1][1], strides[1]), f[1])
        for i in range(2):
            if conn[1][i] == 1 and shapes[i] != out:
                conv(f"p1_{i}", 1, shapes[i][2], f[1]); bn(f"pbn1_{i}", f[1])
    else:
        cs = [shapes[i] for i in range(2) if conn[1][i] == 1]
        out = (cs[0][0], cs[0][1], sum(s[2] for s in cs))
    shapes.append(out)
    conv, bn = real_conv, real_bn
    # third layer
    if f[2] > 0:
        conv("c2", k[2], shapes[-1][2], f[2]); bn("bn2", f[2])
        out = (shapes[-1][0], shapes[-1][1], f[2])
        for i in range(3):
            if conn[2][i] == 1 and shapes[i] != out:
                conv(f"p2_{i}", 1, shapes[i][2], f[2]); bn(f"pbn2_{i}", f[2])
    else:
        cs = []
        for i in range(3):
            if conn[2][i] == 1:
                s = shapes[i]
                if conn[2][-1] == 1 and tuple(strides) != (1, 1) and i < 2:
                    conv(f"s2_{i}", 1, s[2], s[2])
                    s = (same_out(s[0], strides[0]), same_out(s[1], strides[1]), s[2])
                cs.append(s)
        out = (cs[0][0], cs[0][1], sum(s[2] for s in cs))
    sq = float(cfg.get("squeeze_ratio", 0))
    if sq > 0:
        se_filters = int(sq * out[2])
        conv("se0", 1, out[2], se_filters)
        conv("se1", 1, se_filters, out[2])
    return tr, nt, out


def mother_block_forward(cfg: dict, w: Dict[str, torch.Tensor], st: Dict[str, torch.Tensor], new_st: dict, x, training: bool, prefix: str):
    """modules.py:224-296 on NHWC x."""
    check_mother_config(cfg)
    f = [int(cfg[f"filters{i}"]) for i in range(3)]
    conn = [cfg["connect0"], cfg["connect1"], cfg["connect2"]]
    strides = safe_tuple(cfg.get("strides", (1, 1)))
    act = ACTS[cfg.get("activation", "relu")]

    def conv(name, t, s=(1, 1)):
        return conv2d_same(t, w[f"{prefix}.{name}.kernel"], w[f"{prefix}.{name}.bias"], s)

    def bn(name, t):
        y, m, v = O.batchnorm(t, w[f"{prefix}.{name}.gamma"], w[f"{prefix}.{name}.beta"], st[f"{prefix}.{name}.moving_mean"],
                              st[f"{prefix}.{name}.moving_variance"], training)
        new_st[f"{prefix}.{name}.moving_mean"], new_st[f"{prefix}.{name}.moving_variance"] = m, v
        return y

    dead0, dead1 = dead_layers(cfg)
    outputs = [x]
    if dead0:
        out = None
    elif f[0] > 0:
        out = bn("bn0", conv("c0", outputs[-1]))
        if conn[0][0] == 1:
            skip = outputs[-1]
            if skip.shape[-3:] != out.shape[-3:]:
                skip = bn("pbn0_0", conv("p0_0", skip))
            out = out + skip
        out = act(out)
    else:
        out = outputs[-1]
    outputs.append(out)
    if dead1:
        out = None
    elif f[1] > 0:
        out = bn("bn1", conv("c1", outputs[-1], strides))
        for i in range(2):
            if conn[1][i] == 1:
                skip = outputs[i]
                if skip.shape[-3:] != out.shape[-3:]:
                    skip = bn(f"pbn1_{i}", conv(f"p1_{i}", skip, strides))
                out = out + skip
        out = act(out)
    else:
        out = torch.cat([outputs[i] for i in range(2) if conn[1][i] == 1], dim=-1)
    outputs.append(out)
    if f[2] > 0:
        out = bn("bn2", conv("c2", outputs[-1]))
        for i in range(3):
            if conn[2][i] == 1:
                skip = outputs[i]
                if skip.shape[-3:] != out.shape[-3:]:
                    skip = bn(f"pbn2_{i}", conv(f"p2_{i}", skip, (1, 1) if i == 2 else strides))
                out = out + skip
        out = act(out)
    else:
        cs = []
        for i in range(3):
            if conn[2][i] == 1:
                skip = outputs[i]
                if conn[2][-1] == 1 and tuple(strides) != (1, 1) and i < 2:
                    skip = conv(f"s2_{i}", skip, strides)       # connect with strided outputs
                cs.append(skip)
        out = torch.cat(cs, dim=-1)
    if float(cfg.get("squeeze_ratio", 0)) > 0:
        se = out.mean(dim=(-3, -2), keepdim=True)
        se = ACTS[cfg.get("se_activation", "relu")](conv("se0", se))
        se = torch.sigmoid(conv("se1", se))
        out = se * out
    return out


def stage_configs(cfg: dict) -> List[dict]:
    """mother_stage (modules.py:15-43): depth blocks, the strides in the first one only."""
    cfgs = []
    c = copy.deepcopy(cfg)
    for _ in range(int(cfg["depth"])):
        cfgs.append(copy.deepcopy(c))
        c["strides"] = (1, 1)
    return cfgs


def first_configs(model_config: dict) -> List[dict]:
    if model_config["FIRST"] == "mother_stage":
        return stage_configs(model_config["FIRST_ARGS"])
    if model_config["FIRST"] == "mother_block":
        return [copy.deepcopy(model_config["FIRST_ARGS"])]
    raise ValueError("modules_oracle restates mother_block / mother_stage as FIRST")


def _tail_spec(model_config: dict) -> O.Spec:
    """SECOND / SED / DOA of the config through seldnet_oracle's Spec (its FIRST fields are unused here)."""
    mc = copy.deepcopy(model_config)
    mc["FIRST"], mc["FIRST_ARGS"] = "simple_conv_block", {"filters": [], "pool_size": []}
    return O.Spec.from_config(mc)


def variable_specs(model_config: dict, input_shape):
    """-> (trainable, state) [(name, shape)] of models.seldnet(input_shape, model_config) with a mother FIRST block."""
    shape = tuple(int(v) for v in input_shape[-3:])
    tr, nt = [], []
    for d, cfg in enumerate(first_configs(model_config)):
        t, n, shape = mother_block_plan(cfg, shape, f"mb{d}")
        tr += t
        nt += n
    sp = _tail_spec(model_config)
    fin = shape[1] * shape[2]
    for i, u in enumerate(sp.gru_units):
        for dn in ("fwd", "bwd"):
            tr += [(f"gru{i}.{dn}.kernel", (fin, 3 * u)), (f"gru{i}.{dn}.recurrent_kernel", (u, 3 * u)), (f"gru{i}.{dn}.bias", (2, 3 * u))]
        fin = u
    for head, units, n_out in (("sed", sp.sed_units, sp.n_classes), ("doa", sp.doa_units, 3 * sp.n_classes)):
        a = fin
        for j, u in enumerate(units):
            tr += [(f"{head}.dense{j}.kernel", (1, a, u)), (f"{head}.dense{j}.bias", (u,))]
            a = u
        tr += [(f"{head}.out.kernel", (a, n_out)), (f"{head}.out.bias", (n_out,))]
    return tr, nt


def random_weights(model_config: dict, input_shape, seed: int = 0):
    """Keras initial values (glorot_uniform kernels, orthogonal recurrent kernels, zeros / ones) from numpy's default_rng(seed), flat fp32
    (trainable, state) in variable_specs order; BatchNorm's gamma / beta / moving statistics are perturbed so that they matter."""
    rng = np.random.default_rng(seed)
    tr, nt = variable_specs(model_config, input_shape)
    out = []
    for name, sh in tr:
        n = int(np.prod(sh))
        if name.endswith("recurrent_kernel"):
            u = sh[0]
            q = np.concatenate([np.linalg.qr(rng.standard_normal((u, u)))[0] for _ in range(sh[1] // u)], axis=1)
            out.append(q.reshape(-1))
        elif name.endswith("kernel"):
            fan_in = int(np.prod(sh[:-1]))
            fan_out = int(sh[-1]) * (int(np.prod(sh[:-2])) if len(sh) > 2 else 1)
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            out.append(rng.uniform(-lim, lim, n))
        elif name.endswith("gamma"):
            out.append(1.0 + 0.1 * rng.standard_normal(n))
        elif name.endswith("beta") or name.endswith("bias"):
            out.append(0.05 * rng.standard_normal(n))
        else:
            out.append(np.zeros(n))
    w = np.concatenate(out).astype(np.float32)
    s = []
    for name, sh in nt:
        n = int(np.prod(sh))
        s.append(0.1 * rng.standard_normal(n) if name.endswith("moving_mean") else 1.0 + 0.1 * rng.random(n))
    return w, (np.concatenate(s).astype(np.float32) if s else np.zeros(0, np.float32))


def forward(model_config: dict, w: Dict[str, torch.Tensor], st: Dict[str, torch.Tensor], x, training: bool):
    """models.seldnet (models.py:18-32) with FIRST = mother_block / mother_stage -> (sed, doa, new_state)."""
    new_st = dict(st)
    h = x
    for d, cfg in enumerate(first_configs(model_config)):
        h = mother_block_forward(cfg, w, st, new_st, h, training, f"mb{d}")
    sp = _tail_spec(model_config)
    B, S = h.shape[0], h.shape[1]
    h = h.reshape(B, S, -1)          # layers.force_1d_inputs (layers.py:41-47)
    for i in range(len(sp.gru_units)):
        h = O.bigru_mul(h, w, f"gru{i}")
    outs = []
    hid = {None: lambda t: t, "linear": lambda t: t, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}
    for head, units, act, hact in (("sed", sp.sed_units, torch.sigmoid, hid[sp.sed_dense_act]), ("doa", sp.doa_units, torch.tanh, hid[sp.doa_dense_act])):
        a = h
        for j in range(len(units)):
            a = hact(a @ w[f"{head}.dense{j}.kernel"][0] + w[f"{head}.dense{j}.bias"])
        outs.append(act(a @ w[f"{head}.out.kernel"] + w[f"{head}.out.bias"]))
    return outs[0], outs[1], new_st


def train_step(model_config: dict, input_shape, flat_w, flat_state, x, y_sed, y_doa, *, doa_loss="MSE", loss_weight=(1.0, 1000.0), lr=1e-3,
               step=1, dtype=torch.float64):
    """train.trainstep (train.py:22-36) -> dict(sed, doa, sloss, dloss, grad, new_w, new_state), all numpy."""
    tr, nt = variable_specs(model_config, input_shape)
    fw = torch.tensor(np.asarray(flat_w), dtype=dtype, requires_grad=True)
    wd = O.unflatten(fw, tr)
    sd = O.unflatten(torch.tensor(np.asarray(flat_state), dtype=dtype), nt)
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    sed, doa, new_st = forward(model_config, wd, sd, t(x), True)
    obj, sloss, dloss = O.losses_and_objective(sed, doa, t(y_sed), t(y_doa), doa_loss, loss_weight)
    (g,) = torch.autograd.grad(obj, fw)
    m = torch.zeros_like(fw)
    v = torch.zeros_like(fw)
    new_w, _, _ = O.adam_update(fw.detach(), g, m, v, step, lr=lr)
    ns = torch.cat([new_st[n].detach().reshape(-1) for n, _ in nt]) if nt else torch.zeros(0, dtype=dtype)
    return {"sed": sed.detach().numpy(), "doa": doa.detach().numpy(), "sloss": sloss.detach().numpy(), "dloss": dloss.detach().numpy(),
            "grad": g.numpy(), "new_w": new_w.numpy(), "new_state": ns.numpy()}
