"""CPU oracle for the SELDnet train/inference hot path.

*** TEST INFRASTRUCTURE ONLY. ***
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product (``seld_amd``) never does; it
fails loudly when the HIP library is missing.

*** PARITY UNPINNED (numerics). ***
The reference (IRIS-AUDIO/SELD) is TensorFlow/Keras code whose arithmetic lives
in un-vendored ``tensorflow>=2.4.1`` (``requirements.txt:1-8``); TensorFlow is
not installable in the build container (ordinary ``ModuleNotFoundError``), and
the reference's own tests for this path pin only shapes and parameter counts
(``modules_test.py:202-258``, ``complexity_test.py:205-221,292-307``).  This
file is therefore a restatement of the published Keras semantics, anchored on
the reference's call sites.  What IS pinned by reference fixtures is checked in
``tests/test_oracle_pins.py`` (parameter counts 513 840 / 198 144 / 9 360,
output shapes, polar<->cartesian tables).  Independent cross-checks against
``torch.nn.GRU`` / ``torch.nn.functional`` are in the same test file.

Reference call sites restated here (file:line under /root/reference):
  models.py:18-32        seldnet(): FIRST -> SECOND -> SED/DOA heads
  layers.py:14-38        conv2d_bn(): Conv2D(use_bias) -> BatchNormalization -> ReLU
  layers.py:41-47        force_1d_inputs(): [B,T,F,C] -> [B,T,F*C]
  modules.py:302-319     bidirectional_GRU_block(): Bidirectional(GRU, merge_mode='mul')
  modules.py:350-376     simple_dense_block(): Conv1D(units, k=1), no activation
  model_config/seldnet.json:2-8   simple_conv_block (NOT in snapshot; SURVEY.md §8 A3
                         assumption: conv2d_bn(f, 3) -> MaxPooling2D(pool) -> Dropout(0))
  losses.py:4-13         MMSE
  train.py:22-44         trainstep / teststep (incl. the non-scalar Keras MSE quirk)
  train.py:311-320       Adam(lr), BinaryCrossentropy(), tf.keras.losses.MSE
  utils.py:86-96         adaptive_clip_grad

Keras defaults restated (TF 2.4 .. 2.6 semantics):
  BatchNormalization: momentum 0.99, epsilon 1e-3, fused: normalise with the biased
      batch variance, moving_variance updated with the Bessel-corrected one.
  GRU: reset_after=True, recurrent_activation=sigmoid, gate order z|r|h,
      kernel [in,3u], recurrent_kernel [u,3u], bias [2,3u] (row0 input, row1 recurrent).
  BinaryCrossentropy: p clipped to [1e-7, 1-1e-7], log(p + 1e-7), mean over all elements.
  Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+1e-7).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
BCE_EPS = 1e-7
ADAM_EPS = 1e-7


# --------------------------------------------------------------------------- spec
@dataclass
class Spec:
    """Architecture read from a reference model_config dict (model_config/seldnet.json)."""
    in_ch: int
    n_freq: int
    filters: List[int]
    pools: List[Tuple[int, int]]
    gru_units: List[int]
    sed_units: List[int]
    doa_units: List[int]
    n_classes: int = 12
    first: str = "simple_conv_block"   # or "xception_block" / "resnet50_block" (spec/*.md: absent from the reference snapshot)
    xc_blocks: int = 0                 # xception_block: number of middle-flow modules (block_num)
    rn_filters: int = 0                # resnet50_block: base width (FIRST_ARGS.filters)
    rn_blocks: Tuple[int, ...] = ()    # resnet50_block: bottleneck blocks per stage (FIRST_ARGS.block_num)
    sed_dense_act: str | None = None   # simple_dense_block's dense_activation (modules.py:356): activation of the heads' hidden Conv1D layers
    doa_dense_act: str | None = None
    sed_kernel_size: int = 1           # simple_dense_block's kernel_size (modules.py:355, 370-372): Conv1D('same') over the frames of a clip
    doa_kernel_size: int = 1
    sed_dropout: float = 0.0           # simple_dense_block's dropout_rate (modules.py:357, 373-374), training only
    doa_dropout: float = 0.0
    conv_dropout: float = 0.0          # simple_conv_block's dropout_rate (model_config/seldnet.json:7): Dropout behind every MaxPooling2D
    gru_dropout: float = 0.0           # bidirectional_GRU_block's dropout_rate: GRU(dropout=rate, recurrent_dropout=rate) (modules.py:306, 312-314)
    output_coupling: bool = False      # models.seldnet_v1 (models.py:36-52): doa_out = tanh(doa * Concatenate([sed] * 3))
    dropout_seed: int = 0x5e1d5e1d5e1d5e1d   # the library's default (api.hip); option "dropout_seed" v -> 0x5e1d5e1d00000000 ^ v

    @staticmethod
    def from_config(model_config: dict, in_ch: int = 7, n_freq: int = 64) -> "Spec":
        import copy
        model_config = copy.deepcopy(model_config)
        # the reference's stage wrappers build the blocks restated here: bidirectional_GRU_stage (modules.py:46-61), simple_dense_stage
        # (modules.py:86-103: `activation` becomes dense_activation), identity_block (modules.py:639-642)
        if model_config.get("SECOND") == "bidirectional_GRU_stage":
            model_config["SECOND"] = "bidirectional_GRU_block"
            model_config["SECOND_ARGS"]["units"] = [int(model_config["SECOND_ARGS"]["units"])] * int(model_config["SECOND_ARGS"]["depth"])
        for key in ("SED", "DOA"):
            args = model_config.setdefault(key + "_ARGS", {})
            if model_config.get(key) == "simple_dense_stage":
                args["units"] = [int(args["units"])] * int(args["depth"])
                args["dense_activation"] = args.get("activation", None)
            elif model_config.get(key) == "identity_block":
                args["units"] = []
        sp = Spec._from_config(model_config, in_ch, n_freq)
        sp.sed_dense_act = model_config["SED_ARGS"].get("dense_activation")
        sp.doa_dense_act = model_config["DOA_ARGS"].get("dense_activation")
        sp.sed_kernel_size = int(model_config["SED_ARGS"].get("kernel_size", 1))
        sp.doa_kernel_size = int(model_config["DOA_ARGS"].get("kernel_size", 1))
        sp.sed_dropout = float(model_config["SED_ARGS"].get("dropout_rate", 0))
        sp.doa_dropout = float(model_config["DOA_ARGS"].get("dropout_rate", 0))
        sp.conv_dropout = float(model_config["FIRST_ARGS"].get("dropout_rate", 0) or 0) if sp.first == "simple_conv_block" else 0.0
        sp.gru_dropout = float(model_config["SECOND_ARGS"].get("dropout_rate", 0) or 0)
        for a in (sp.sed_dense_act, sp.doa_dense_act):
            if a not in (None, "linear", "relu", "tanh", "sigmoid"):
                raise ValueError(f"dense_activation {a!r} not restated")
        return sp

    @staticmethod
    def _from_config(model_config: dict, in_ch: int = 7, n_freq: int = 64) -> "Spec":
        if model_config["FIRST"] not in ("simple_conv_block", "xception_block", "resnet50_block"):
            raise ValueError("oracle restates simple_conv_block, xception_block and resnet50_block only")
        if model_config["SECOND"] != "bidirectional_GRU_block":
            raise ValueError("oracle restates bidirectional_GRU_block only")
        fa = model_config["FIRST_ARGS"]
        if model_config["FIRST"] == "xception_block":
            # spec/XCEPTION_BLOCK.md: entry conv2d_bn(2 filters) + MaxPool(5,4), block_num residual modules of three
            # SeparableConv2D(2 filters) + BN, exit ReLU + MaxPool(1,8)
            return Spec(in_ch=in_ch, n_freq=n_freq, filters=[2 * int(fa["filters"])], pools=[(5, 4)],
                        gru_units=list(model_config["SECOND_ARGS"]["units"]), sed_units=list(model_config["SED_ARGS"]["units"]),
                        doa_units=list(model_config["DOA_ARGS"]["units"]), n_classes=int(model_config.get("n_classes", 12)),
                        first="xception_block", xc_blocks=int(fa["block_num"]))
        if model_config["FIRST"] == "resnet50_block":
            # spec/RESNET50_BLOCK.md: entry conv2d_bn(2 filters) + MaxPool(5,4), four stages of bottleneck blocks
            return Spec(in_ch=in_ch, n_freq=n_freq, filters=[2 * int(fa["filters"])], pools=[(5, 4)],
                        gru_units=list(model_config["SECOND_ARGS"]["units"]), sed_units=list(model_config["SED_ARGS"]["units"]),
                        doa_units=list(model_config["DOA_ARGS"]["units"]), n_classes=int(model_config.get("n_classes", 12)),
                        first="resnet50_block", rn_filters=int(fa["filters"]), rn_blocks=tuple(int(v) for v in fa["block_num"]))
        return Spec(
            in_ch=in_ch, n_freq=n_freq,
            filters=list(fa["filters"]),
            pools=[tuple(p) for p in fa["pool_size"]],
            gru_units=list(model_config["SECOND_ARGS"]["units"]),
            sed_units=list(model_config["SED_ARGS"]["units"]),
            doa_units=list(model_config["DOA_ARGS"]["units"]),
            # train.py:306-307 forces n_classes = 12 regardless of the JSON
            n_classes=int(model_config.get("n_classes", 12)),
        )


def resnet_plan(spec: Spec):
    """spec/RESNET50_BLOCK.md as data: [(stage, block, cin, width, stride_f, projection)] in forward order, cin = the block's input channels"""
    plan, cin = [], spec.filters[-1]
    for s, nb in enumerate(spec.rn_blocks):
        w = spec.rn_filters * (2 ** s)
        for b in range(nb):
            st = 2 if (b == 0 and s > 0) else 1
            plan.append((s, b, cin, w, st, b == 0))
            cin = 4 * w
    return plan


def variable_specs(spec: Spec) -> Tuple[List[Tuple[str, Tuple[int, ...]]], List[Tuple[str, Tuple[int, ...]]]]:
    """(trainable, non_trainable) variable (name, shape) lists in Keras creation order."""
    tr: List[Tuple[str, Tuple[int, ...]]] = []
    nt: List[Tuple[str, Tuple[int, ...]]] = []
    cin = spec.in_ch
    for i, f in enumerate(spec.filters):
        tr += [(f"conv{i}.kernel", (3, 3, cin, f)), (f"conv{i}.bias", (f,)),
               (f"bn{i}.gamma", (f,)), (f"bn{i}.beta", (f,))]
        nt += [(f"bn{i}.moving_mean", (f,)), (f"bn{i}.moving_variance", (f,))]
        cin = f
    fr = spec.n_freq
    for p in spec.pools:
        fr //= p[1]
    if spec.first == "xception_block":
        for b in range(spec.xc_blocks):
            for u in range(3):
                # Keras SeparableConv2D(use_bias=False): depthwise_kernel [kh,kw,in,depth_multiplier], pointwise_kernel [1,1,in,out]
                tr += [(f"xc{b}.{u}.depthwise_kernel", (3, 3, cin, 1)), (f"xc{b}.{u}.pointwise_kernel", (1, 1, cin, cin)),
                       (f"xc{b}.{u}.gamma", (cin,)), (f"xc{b}.{u}.beta", (cin,))]
                nt += [(f"xc{b}.{u}.moving_mean", (cin,)), (f"xc{b}.{u}.moving_variance", (cin,))]
        fr //= 8      # exit MaxPooling2D((1, 8))
    cout = spec.filters[-1]
    if spec.first == "resnet50_block":
        for s_, b, ci, wd, stf, proj in resnet_plan(spec):
            pre = f"rn{s_}.{b}"
            for i, shape in enumerate(((1, 1, ci, wd), (3, 3, wd, wd), (1, 1, wd, 4 * wd))):
                tr += [(f"{pre}.c{i}.kernel", shape), (f"{pre}.c{i}.gamma", (shape[-1],)), (f"{pre}.c{i}.beta", (shape[-1],))]
                nt += [(f"{pre}.c{i}.moving_mean", (shape[-1],)), (f"{pre}.c{i}.moving_variance", (shape[-1],))]
            if proj:
                tr += [(f"{pre}.sc.kernel", (1, 1, ci, 4 * wd)), (f"{pre}.sc.gamma", (4 * wd,)), (f"{pre}.sc.beta", (4 * wd,))]
                nt += [(f"{pre}.sc.moving_mean", (4 * wd,)), (f"{pre}.sc.moving_variance", (4 * wd,))]
            fr //= stf
            cout = 4 * wd
    feat = fr * cout
    for i, u in enumerate(spec.gru_units):
        for d in ("fwd", "bwd"):
            tr += [(f"gru{i}.{d}.kernel", (feat, 3 * u)),
                   (f"gru{i}.{d}.recurrent_kernel", (u, 3 * u)),
                   (f"gru{i}.{d}.bias", (2, 3 * u))]
        feat = u
    for head, units, nout, ks in (("sed", spec.sed_units, spec.n_classes, spec.sed_kernel_size),
                                  ("doa", spec.doa_units, 3 * spec.n_classes, spec.doa_kernel_size)):
        fin = feat
        for j, u in enumerate(units):
            # a Conv1D kernel is rank 3 in Keras: [kernel_size, in, out] (seldnet.json: kernel_size 1)
            tr += [(f"{head}.dense{j}.kernel", (ks, fin, u)), (f"{head}.dense{j}.bias", (u,))]
            fin = u
        tr += [(f"{head}.out.kernel", (fin, nout)), (f"{head}.out.bias", (nout,))]
    return tr, nt


def param_count(spec: Spec) -> int:
    return sum(int(np.prod(s)) for _, s in variable_specs(spec)[0])


def unflatten(flat: torch.Tensor, specs) -> Dict[str, torch.Tensor]:
    out, off = {}, 0
    for name, shape in specs:
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    assert off == flat.numel(), (off, flat.numel())
    return out


def random_weights(spec: Spec, seed: int = 0, dtype=np.float32):
    """Non-degenerate random weights for parity tests (every bias/gamma/beta/moving stat
    is non-trivial so that no term can silently drop out)."""
    rng = np.random.default_rng(seed)
    tr, nt = variable_specs(spec)
    parts = []
    for name, shape in tr:
        n = int(np.prod(shape))
        if name.endswith("gamma"):
            v = rng.uniform(0.5, 1.5, n) * np.where(rng.random(n) < 0.1, -1.0, 1.0)
        elif name.endswith("beta") or name.endswith("bias"):
            v = rng.normal(0, 0.1, n)
        elif name.endswith("recurrent_kernel"):
            v = rng.normal(0, 1.0 / math.sqrt(shape[0]), n)
        elif name.endswith("depthwise_kernel"):
            v = rng.normal(0, 1.0 / 3.0, n)          # 9 taps per channel
        else:
            fan_in = int(np.prod(shape[:-1]))
            v = rng.normal(0, 1.0 / math.sqrt(fan_in), n)
        parts.append(v)
    flat = np.concatenate(parts).astype(dtype)
    sparts = []
    for name, shape in nt:
        n = int(np.prod(shape))
        sparts.append(rng.normal(0, 0.1, n) if name.endswith("mean") else rng.uniform(0.5, 1.5, n))
    state = np.concatenate(sparts).astype(dtype)
    return flat, state


# --------------------------------------------------------------------------- layers
def conv2d_same_nhwc(x, kernel_hwio, bias):
    """Keras Conv2D(k=3, strides 1, padding='same', use_bias=True) on NHWC (layers.py:27-32)."""
    w = kernel_hwio.permute(3, 2, 0, 1)  # HWIO -> OIHW
    y = F.conv2d(x.permute(0, 3, 1, 2), w, bias, stride=1, padding=1)
    return y.permute(0, 2, 3, 1)


# Diagnostic aid (tools/diag_fp32_draws.py): "shifted" makes the FORWARD VALUE of a training-mode BatchNormalization z * scale + shift with
# scale = gamma * invstd and shift = beta - mean * scale rounded to z's dtype first — the one-fma form a kernel applies — while the gradient
# stays that of the centred form.  In fp32 the shifted form loses |mean| / std more digits than the centred one.  Never set by tests or fixtures.
BN_FORM = "centred"


def batchnorm(z, gamma, beta, mov_mean, mov_var, training: bool, sync=None):
    """Keras BatchNormalization(axis=-1) (layers.py:33), fused semantics. Returns y, new stats.
    `sync = (allreduce_sum, world)`: synchronised statistics for data parallelism (what seld_set_sync_bn computes): the
    per-channel sums are summed over the ranks by the differentiable `allreduce_sum(tensor) -> tensor`, so that B/world
    clips per rank reproduce a single-device batch of B."""
    if training and sync is not None:
        allreduce_sum, world = sync
        n = (z.numel() // z.shape[-1]) * world
        s1 = allreduce_sum(z.sum(dim=(0, 1, 2)))
        s2 = allreduce_sum((z * z).sum(dim=(0, 1, 2)))
        mean = s1 / n
        var = s2 / n - mean * mean  # biased, from the global sums (the library's formula)
        y = (z - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta
        with torch.no_grad():
            f = 1.0 - BN_MOMENTUM
            new_mean = mov_mean * (1 - f) + mean * f
            new_var = mov_var * (1 - f) + var * (n / max(n - 1, 1)) * f
        return y, new_mean.detach(), new_var.detach()
    if training:
        n = z.numel() // z.shape[-1]
        mean = z.mean(dim=(0, 1, 2))
        var = ((z - mean) ** 2).mean(dim=(0, 1, 2))  # biased
        y = (z - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta
        if BN_FORM == "shifted":
            with torch.no_grad():
                sc = gamma * torch.rsqrt(var + BN_EPS)
                alt = z * sc + (beta - mean * sc)
            y = y + (alt - y).detach()
        with torch.no_grad():
            f = 1.0 - BN_MOMENTUM
            new_mean = mov_mean * (1 - f) + mean * f
            new_var = mov_var * (1 - f) + var * (n / max(n - 1, 1)) * f
        return y, new_mean.detach(), new_var.detach()
    y = (z - mov_mean) * torch.rsqrt(mov_var + BN_EPS) * gamma + beta
    return y, mov_mean, mov_var


def maxpool_nhwc(a, pool):
    """Keras MaxPooling2D(pool_size=pool) defaults: strides=pool, padding='valid'."""
    return F.max_pool2d(a.permute(0, 3, 1, 2), kernel_size=tuple(pool), stride=tuple(pool)).permute(0, 2, 3, 1)


def pool_windows(y, pool):
    """[B,H,W,C] -> [B,H/pt,W/pf,C,pt*pf]: the elements of every pooling window, position = row*pf + col."""
    B, H, W, C = y.shape
    pt, pf = pool
    return y.reshape(B, H // pt, pt, W // pf, pf, C).permute(0, 1, 3, 5, 2, 4).reshape(B, H // pt, W // pf, C, pt * pf)


def gru_direction(x, kernel, rec_kernel, bias, reverse: bool, return_aux: bool = False, in_mask=None, rec_mask=None):
    """Keras GRU(units, reset_after=True, return_sequences=True) over [B,S,I] (modules.py:312-315).
    reverse=True is Bidirectional's backward layer: consume time-reversed input, output re-reversed.
    in_mask [B,I] / rec_mask [B,u] (training with dropout / recurrent_dropout > 0; values 0 | 1/(1-rate), one row per clip for the whole
    sequence): Keras GRUCell.call, implementation 2 (the default) — `inputs = inputs * dp_mask[0]` before the kernel product and
    `h_tm1 = h_tm1 * rec_dp_mask[0]` before the recurrent product; h_tm1 is REASSIGNED there, so the blend z * h_tm1 + (1 - z) * hh
    sees the masked state too, while the emitted output (and the state handed to the next step, masked again there) is the unmasked h.
    Restated from the published Keras source (tensorflow>=2.4.1, requirements.txt:2), which is not in the snapshot: unpinned."""
    B, S, _ = x.shape
    u = rec_kernel.shape[0]
    if in_mask is not None:
        x = x * in_mask[:, None, :]
    gx = x @ kernel + bias[0]  # [B,S,3u]
    h = x.new_zeros(B, u)
    outs = [None] * S
    order = range(S - 1, -1, -1) if reverse else range(S)
    for t in order:
        if rec_mask is not None:
            h = h * rec_mask
        gh = h @ rec_kernel + bias[1]
        z = torch.sigmoid(gx[:, t, :u] + gh[:, :u])
        r = torch.sigmoid(gx[:, t, u:2 * u] + gh[:, u:2 * u])
        hh = torch.tanh(gx[:, t, 2 * u:] + r * gh[:, 2 * u:])
        h = z * h + (1 - z) * hh
        outs[t] = h
    return torch.stack(outs, dim=1)


def bigru_mul(x, w, prefix, masks=None):
    """Bidirectional(GRU, merge_mode='mul') (modules.py:311-316).  masks = ((in_f, rec_f), (in_b, rec_b)): each direction is its own cell
    with its own dropout draws."""
    mf, mb = masks if masks is not None else ((None, None), (None, None))
    hf = gru_direction(x, w[f"{prefix}.fwd.kernel"], w[f"{prefix}.fwd.recurrent_kernel"], w[f"{prefix}.fwd.bias"], False, in_mask=mf[0], rec_mask=mf[1])
    hb = gru_direction(x, w[f"{prefix}.bwd.kernel"], w[f"{prefix}.bwd.recurrent_kernel"], w[f"{prefix}.bwd.bias"], True, in_mask=mb[0], rec_mask=mb[1])
    return hf * hb


# --------------------------------------------------------------------------- model
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC11; Random123): uint64 arrays holding 32-bit
    counter / key words -> the four output words.  tests/test_oracle_cpu.py checks the published known-answer vectors."""
    m32 = np.uint64(0xffffffff)
    c = [np.asarray(v, np.uint64) & m32 for v in (c0, c1, c2, c3)]
    k0, k1 = np.uint64(k0) & m32, np.uint64(k1) & m32
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c[0], np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return c


def philox_uniform(n: int, seed: int, layer: int, step: int) -> np.ndarray:
    """The library's dropout uniforms (loss_adam.hip::dropout_kernel): word j of Philox4x32-10 at counter (i, layer, step, i >> 32) under the
    key (seed & 0xffffffff, seed >> 32) is element 4 i + j; u = (word >> 8) * 2^-24.  Not TensorFlow's generator (no other program
    reproduces that): the reference's Dropout (modules.py:373-374) fixes the distribution, this fixes the draws."""
    n4 = (n + 3) // 4
    i = np.arange(n4, dtype=np.uint64)
    c = philox4x32_10(i, np.full(n4, layer, np.uint64), np.full(n4, step, np.uint64), i >> np.uint64(32), seed, seed >> 32)
    words = np.stack(c, axis=1).reshape(-1)[:n]
    return (words >> np.uint64(8)).astype(np.float64) * 2.0 ** -24


def dropout_mask(shape, rate: float, seed: int, layer: int, step: int, dtype) -> torch.Tensor:
    """Keras dropout mask from the library's draws: 0 where uniform < rate, 1 / (1 - rate) elsewhere (rate as the fp32 the library holds).
    Draw streams (`layer`): heads 16 hd + j, simple_conv_block's Dropout behind pool i 64 + i, GRU layer i direction d input mask
    96 + 4 i + d and state mask 98 + 4 i + d (api.hip forward_impl)."""
    n = int(np.prod(shape))
    u = philox_uniform(n, seed, layer, step).reshape(shape)
    return torch.as_tensor((u >= np.float32(rate)).astype(np.float64) / (1.0 - float(np.float32(rate))), dtype=dtype)


def conv1d_same(a, kernel, bias):
    """Conv1D(units, kernel_size, padding='same') over axis 1 of a [B, S, C] tensor (modules.py:370-372); kernel [ks, C, units].
    TensorFlow's 'same': pad (ks - 1) // 2 frames in front, the rest behind."""
    ks = kernel.shape[0]
    if ks == 1:
        return a @ kernel[0] + bias
    S = a.shape[1]
    ap = torch.nn.functional.pad(a, (0, 0, (ks - 1) // 2, ks - 1 - (ks - 1) // 2))
    return sum(ap[:, j:j + S] @ kernel[j] for j in range(ks)) + bias


def forward(spec: Spec, w: Dict[str, torch.Tensor], st: Dict[str, torch.Tensor], x, training: bool,
            taps: dict | None = None, routing: dict | None = None, record_routing: dict | None = None, bn_sync=None,
            dropout_step: int = 0):
    """models.seldnet forward (models.py:18-32). x [B,T,F,C] -> sed [B,S,nc], doa [B,S,3nc].
    Returns (sed, doa, new_state). `taps` (optional dict) receives intermediate tensors.

    Test aids for full-size parity (tests/test_model_gpu.py::test_parity_given_identical_routing): MaxPool(ReLU(.)) of block i
    is a ROUTING decision per pooled element — which window position passes (argmax) and whether it passes at all (> 0).
    `routing[i] = (pos int64 [B,H/pt,W/pf,C], gate bool same shape)` replaces the block's own decision by a given one (the
    function stays differentiable: the gradient then flows exactly where the given routing says); `record_routing[i]`
    receives the free decision (pos, gate) and the fp64 margin behind it (top1 - top2 of the window, |top1|).
    resnet50_block's ReLUs take the same aids under the keys "rn{s}.{b}.y0" / ".y1" / ".out" (gate only)."""
    new_st = {}
    h = x
    for i in range(len(spec.filters)):
        z = conv2d_same_nhwc(h, w[f"conv{i}.kernel"], w[f"conv{i}.bias"])
        y, m, v = batchnorm(z, w[f"bn{i}.gamma"], w[f"bn{i}.beta"],
                            st[f"bn{i}.moving_mean"], st[f"bn{i}.moving_variance"], training, sync=bn_sync)
        new_st[f"bn{i}.moving_mean"], new_st[f"bn{i}.moving_variance"] = m, v
        if routing is not None and i in routing:
            pos, gate = routing[i]
            yw = pool_windows(y, spec.pools[i])
            h = torch.where(gate, yw.gather(-1, pos.unsqueeze(-1)).squeeze(-1), torch.zeros((), dtype=y.dtype))
        else:
            h = maxpool_nhwc(torch.relu(y), spec.pools[i])
        if record_routing is not None:
            with torch.no_grad():
                yw = pool_windows(y.detach(), spec.pools[i])
                k = min(2, yw.shape[-1])
                top, idx = yw.topk(k, dim=-1)
                gap = top[..., 0] - top[..., 1] if k == 2 else torch.full_like(top[..., 0], float("inf"))
                record_routing[i] = {"pos": idx[..., 0], "gate": top[..., 0] > 0, "gap": gap, "top": top[..., 0], "windows": yw}
        if taps is not None:
            taps[f"conv{i}.z"] = z
            taps[f"pool{i}"] = h
        if training and spec.conv_dropout > 0 and spec.first == "simple_conv_block":      # Dropout(dropout_rate) behind the pool (seldnet.json:7)
            h = h * dropout_mask(h.shape, spec.conv_dropout, spec.dropout_seed, 64 + i, dropout_step, h.dtype)
    if spec.first == "xception_block":
        # spec/XCEPTION_BLOCK.md: MIDDLE flow (residual modules of three ReLU -> SeparableConv2D -> BN) and EXIT (ReLU -> pool (1,8))
        C = h.shape[-1]
        for b in range(spec.xc_blocks):
            y = h
            for u in range(3):
                pre = f"xc{b}.{u}"
                if record_routing is not None:       # the unit's ReLU gate and the pre-activation behind it (tests/golden/make_golden_blocks.py)
                    record_routing[f"{pre}.in"] = {"gate": y.detach() > 0, "pre": y.detach()}
                if routing is not None and f"{pre}.in" in routing:      # a GIVEN gate (round 5: the fp32 oracle evaluated on the fp64 decisions)
                    y = torch.where(routing[f"{pre}.in"], y, torch.zeros((), dtype=y.dtype))
                else:
                    y = torch.relu(y)
                dw = w[f"{pre}.depthwise_kernel"].permute(2, 3, 0, 1)           # [kh,kw,in,1] -> [in,1,kh,kw]
                y = F.conv2d(y.permute(0, 3, 1, 2), dw, None, stride=1, padding=1, groups=C)
                pw = w[f"{pre}.pointwise_kernel"][0, 0]                        # [in,out]
                y = y.permute(0, 2, 3, 1) @ pw
                y, m, v = batchnorm(y, w[f"{pre}.gamma"], w[f"{pre}.beta"], st[f"{pre}.moving_mean"], st[f"{pre}.moving_variance"],
                                    training, sync=bn_sync)
                new_st[f"{pre}.moving_mean"], new_st[f"{pre}.moving_variance"] = m, v
            h = h + y
            if taps is not None:
                taps[f"xc{b}"] = h
        if record_routing is not None:           # the exit's MaxPool(ReLU(.)) routing, as the conv blocks record theirs
            Bq, Hq, Wq, Cq = h.shape
            yw = h.detach().reshape(Bq, Hq, Wq // 8, 8, Cq).permute(0, 1, 2, 4, 3)
            top, idx = yw.topk(2, dim=-1)
            record_routing["exit"] = {"pos": idx[..., 0], "gate": top[..., 0] > 0, "gap": top[..., 0] - top[..., 1], "top": top[..., 0], "windows": yw}
        if routing is not None and "exit" in routing:       # a GIVEN exit routing (pos, gate), as the conv blocks take theirs
            pos, gate = routing["exit"]
            Bq, Hq, Wq, Cq = h.shape
            yw = h.reshape(Bq, Hq, Wq // 8, 8, Cq).permute(0, 1, 2, 4, 3)
            h = torch.where(gate, yw.gather(-1, pos.unsqueeze(-1)).squeeze(-1), torch.zeros((), dtype=h.dtype))
        else:
            h = maxpool_nhwc(torch.relu(h), (1, 8))
    if spec.first == "resnet50_block":
        # spec/RESNET50_BLOCK.md: bottleneck blocks, strides on the frequency axis
        def conv_bn(t, pre, stride_f, k):
            kern = w[f"{pre}.kernel"].permute(3, 2, 0, 1)
            t = F.conv2d(t.permute(0, 3, 1, 2), kern, None, stride=(1, stride_f), padding=k // 2).permute(0, 2, 3, 1)
            t, m, v = batchnorm(t, w[f"{pre}.gamma"], w[f"{pre}.beta"], st[f"{pre}.moving_mean"], st[f"{pre}.moving_variance"],
                                training, sync=bn_sync)
            new_st[f"{pre}.moving_mean"], new_st[f"{pre}.moving_variance"] = m, v
            return t
        def relu_gate(t, key):
            # routing[key] (bool, t's shape): the ReLU's gate given from outside (the decision an fp32 evaluation took); the
            # function stays differentiable and the gradient passes exactly where the given gate says.  record_routing[key]
            # receives the free decision and the pre-activation behind it.
            if record_routing is not None:
                record_routing[key] = {"gate": t.detach() > 0, "pre": t.detach()}
            if routing is not None and key in routing:
                return torch.where(routing[key], t, torch.zeros((), dtype=t.dtype))
            return torch.relu(t)
        for s_, b, ci, wd, stf, proj in resnet_plan(spec):
            pre = f"rn{s_}.{b}"
            y = relu_gate(conv_bn(h, f"{pre}.c0", stf, 1), f"{pre}.y0")
            y = relu_gate(conv_bn(y, f"{pre}.c1", 1, 3), f"{pre}.y1")
            y = conv_bn(y, f"{pre}.c2", 1, 1)
            r = conv_bn(h, f"{pre}.sc", stf, 1) if proj else h
            h = relu_gate(y + r, f"{pre}.out")
            if taps is not None:
                taps[pre] = h
    B, S = h.shape[0], h.shape[1]
    h = h.reshape(B, S, -1)  # layers.force_1d_inputs: feature index = f*C + c
    for i in range(len(spec.gru_units)):
        masks = None
        if training and spec.gru_dropout > 0:      # GRU(dropout=rate, recurrent_dropout=rate) (modules.py:312-314): per direction, per clip
            masks = tuple((dropout_mask((B, h.shape[-1]), spec.gru_dropout, spec.dropout_seed, 96 + 4 * i + d, dropout_step, h.dtype),
                           dropout_mask((B, spec.gru_units[i]), spec.gru_dropout, spec.dropout_seed, 98 + 4 * i + d, dropout_step, h.dtype))
                          for d in range(2))
        h = bigru_mul(h, w, f"gru{i}", masks)
        if taps is not None:
            taps[f"gru{i}"] = h
    outs = []
    hidden = {None: lambda t: t, "linear": lambda t: t, "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}
    for hd, (head, units, act, hact, rate) in enumerate((("sed", spec.sed_units, torch.sigmoid, hidden[spec.sed_dense_act], spec.sed_dropout),
                                                         ("doa", spec.doa_units, torch.tanh, hidden[spec.doa_dense_act], spec.doa_dropout))):
        a = h
        for j in range(len(units)):
            # Conv1D(units, kernel_size, padding='same', activation=dense_activation) [+ Dropout] (modules.py:355-357, 368-374)
            a = hact(conv1d_same(a, w[f"{head}.dense{j}.kernel"], w[f"{head}.dense{j}.bias"]))
            if training and rate > 0:
                a = a * dropout_mask(a.shape, rate, spec.dropout_seed, 16 * hd + j, dropout_step, a.dtype)
        outs.append(act(a @ w[f"{head}.out.kernel"] + w[f"{head}.out.bias"]))
    if spec.output_coupling:                     # models.seldnet_v1 (models.py:48-50)
        outs[1] = torch.tanh(outs[1] * torch.cat([outs[0]] * 3, dim=-1))
    return outs[0], outs[1], new_st


# --------------------------------------------------------------------------- losses
def bce(y_true, p):
    """tf.keras.losses.BinaryCrossentropy() (train.py:312-313): scalar mean over all elements."""
    p = torch.clamp(p, BCE_EPS, 1.0 - BCE_EPS)
    return (-(y_true * torch.log(p + BCE_EPS) + (1 - y_true) * torch.log(1 - p + BCE_EPS))).mean()


def keras_mse_fn(y_true, y_pred):
    """tf.keras.losses.MSE *function*: mean over the last axis only -> [B,S] (train.py:317-320)."""
    return ((y_true - y_pred) ** 2).mean(dim=-1)


def mmse(y_true, y_pred):
    """losses.MMSE (losses.py:4-13)."""
    sh = y_true.shape
    sed = y_true.reshape(*sh[:-1], 3, -1)
    sed = torch.round((sed ** 2).sum(dim=-2))
    sed = torch.cat([sed] * 3, dim=-1)
    return (((y_true - y_pred) ** 2) * sed).sum() / sed.sum()


def losses_and_objective(sed, doa, y_sed, y_doa, doa_loss: str, loss_weight):
    """train.py:24-31. Returns (objective whose gradient tape.gradient computes, sloss, dloss).
    With the Keras MSE function dloss is [B,S]; `loss` is then non-scalar and tape.gradient
    differentiates its SUM."""
    sloss = bce(y_sed, sed)
    if doa_loss == "MSE":
        dloss = keras_mse_fn(y_doa, doa)
    elif doa_loss == "MAE":       # tf.keras.losses.MAE function (train.py:317-318 with --doa_loss MAE): mean over the last axis
        dloss = (y_doa - doa).abs().mean(dim=-1)
    elif doa_loss == "MSLE":      # tf.keras.losses.MSLE function: log(max(., epsilon()) + 1), epsilon() = 1e-7
        eps = torch.tensor(1e-7, dtype=doa.dtype)
        dloss = ((torch.log(torch.maximum(doa, eps) + 1.0) - torch.log(torch.maximum(y_doa, eps) + 1.0)) ** 2).mean(dim=-1)
    elif doa_loss == "MMSE":
        dloss = mmse(y_doa, doa)
    else:
        raise ValueError(doa_loss)
    loss = sloss * loss_weight[0] + dloss * loss_weight[1]
    return loss.sum(), sloss, dloss


# --------------------------------------------------------------------------- AGC / Adam
def unitwise_norm(x):
    """utils.py:66-84."""
    if x.dim() <= 1:
        return (x ** 2).sum() ** 0.5
    if x.dim() in (2, 3):  # rank-3 = Conv1D kernel [1,in,out]: norm over the size-1 axis
        return (x ** 2).sum(dim=0, keepdim=True) ** 0.5
    if x.dim() == 4:
        return (x ** 2).sum(dim=(0, 1, 2), keepdim=True) ** 0.5
    raise ValueError("unsupported rank")


def adaptive_clip_grad(params, grads, clip_factor=0.01, eps=1e-3):
    """utils.py:86-96."""
    out = []
    for p, g in zip(params, grads):
        max_norm = torch.clamp(unitwise_norm(p), min=eps) * clip_factor
        gn = unitwise_norm(g)
        clipped = g * (max_norm / torch.clamp(gn, min=1e-6))
        out.append(torch.where(gn < max_norm, g, clipped))
    return out


def adam_update(theta, g, m, v, step: int, lr=1e-3, b1=0.9, b2=0.999, eps=ADAM_EPS):
    """Keras Adam (train.py:311), ResourceApplyAdam form. `step` is 1-based."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    theta = theta - lr_t * m / (torch.sqrt(v) + eps)
    return theta, m, v


# --------------------------------------------------------------------------- steps
def _as_t(a, dtype):
    return torch.as_tensor(np.asarray(a)).to(dtype)


def test_step(spec: Spec, flat_w, flat_state, x, y_sed, y_doa, doa_loss="MSE", dtype=torch.float32):
    """train.teststep (train.py:39-44): forward(training=False) + losses."""
    tr, nt = variable_specs(spec)
    with torch.no_grad():
        w = unflatten(_as_t(flat_w, dtype), tr)
        st = unflatten(_as_t(flat_state, dtype), nt)
        sed, doa, _ = forward(spec, w, st, _as_t(x, dtype), training=False)
        _, sloss, dloss = losses_and_objective(sed, doa, _as_t(y_sed, dtype), _as_t(y_doa, dtype), doa_loss, (1.0, 1.0))
    return {"sed": sed.numpy(), "doa": doa.numpy(), "sloss": sloss.numpy(), "dloss": dloss.numpy()}


def train_step(spec: Spec, flat_w, flat_state, x, y_sed, y_doa, *, doa_loss="MSE", loss_weight=(1.0, 1000.0),
               lr=1e-3, step=1, m=None, v=None, agc=False, dtype=torch.float32, want_taps=False, routing=None,
               record_routing=None, dropout_step=0):
    """train.trainstep (train.py:22-36). Returns dict with outputs, losses, flat grads,
    updated flat weights / BN state / Adam slots."""
    tr, nt = variable_specs(spec)
    fw = _as_t(flat_w, dtype).clone().requires_grad_(True)
    w = unflatten(fw, tr)
    st = unflatten(_as_t(flat_state, dtype), nt)
    taps = {} if want_taps else None
    sed, doa, new_st = forward(spec, w, st, _as_t(x, dtype), training=True, taps=taps, routing=routing,
                               record_routing=record_routing, dropout_step=dropout_step)
    obj, sloss, dloss = losses_and_objective(sed, doa, _as_t(y_sed, dtype), _as_t(y_doa, dtype), doa_loss, loss_weight)
    (g,) = torch.autograd.grad(obj, fw)
    with torch.no_grad():
        if agc:
            gs = adaptive_clip_grad(list(unflatten(fw, tr).values()), list(unflatten(g, tr).values()))
            g = torch.cat([t.reshape(-1) for t in gs])
        m0 = torch.zeros_like(fw) if m is None else _as_t(m, dtype)
        v0 = torch.zeros_like(fw) if v is None else _as_t(v, dtype)
        new_w, m1, v1 = adam_update(fw.detach(), g, m0, v0, step, lr=lr)
        new_state = torch.cat([new_st[name].reshape(-1) for name, _ in nt])
    out = {"sed": sed.detach().numpy(), "doa": doa.detach().numpy(),
           "sloss": sloss.detach().numpy(), "dloss": dloss.detach().numpy(),
           "grad": g.numpy(), "new_w": new_w.numpy(), "new_state": new_state.numpy(),
           "m": m1.numpy(), "v": v1.numpy()}
    if want_taps:
        out["taps"] = {k: t.detach().numpy() for k, t in taps.items()}
    return out


# --------------------------------------------------------------------------- synthetic data
def synthetic_batch(B: int, T: int, F_: int = 64, C: int = 7, n_classes: int = 12, seed: int = 1234, pool_t: int = 5):
    """SURVEY.md §8(d) synthetic data: x~N(0,1); sed~Bernoulli(0.1) (>=1 active);
    doa = unit vectors * sed in [x|y|z] block layout (transforms.py:117-119)."""
    rng = np.random.default_rng(seed)
    S = T // pool_t
    x = rng.standard_normal((B, T, F_, C), dtype=np.float32)
    sed = (rng.random((B, S, n_classes)) < 0.1).astype(np.float32)
    sed[0, 0, 0] = 1.0
    vec = rng.standard_normal((B, S, 3, n_classes))
    vec /= np.linalg.norm(vec, axis=2, keepdims=True)
    doa = (vec * sed[:, :, None, :]).reshape(B, S, 3 * n_classes).astype(np.float32)
    return x, sed, doa
