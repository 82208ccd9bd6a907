"""Host-side mirror of the reference's models.py for the SELDnet hot path.

`seldnet(input_shape, model_config)` (reference models.py:18-32, called at train.py:308) returns a
`SeldNet` whose arithmetic runs entirely in libseld_hip.so.  The block names are resolved the way the
reference resolves them (`getattr(modules, model_config['FIRST'])`, models.py:24-29), but only the
blocks of model_config/seldnet.json have kernels; anything else raises ValueError, as the
reference's block factories do for bad configs (modules.py:202-222).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

SUPPORTED_FIRST = ("simple_conv_block", "xception_block", "resnet50_block")
SUPPORTED_SECOND = ("bidirectional_GRU_block",)
SUPPORTED_HEAD = ("simple_dense_block",)


def canonical_config(model_config: dict) -> dict:
    """The reference's stage wrappers and identity_block as the blocks they build: `bidirectional_GRU_stage` (modules.py:46-61: depth x
    [units]) -> bidirectional_GRU_block, `simple_dense_stage` (modules.py:86-103: depth x [units], `activation` -> dense_activation) ->
    simple_dense_block, `identity_block` (modules.py:639-642) as a head -> simple_dense_block without hidden layers."""
    import copy
    cfg = copy.deepcopy(model_config)
    if cfg.get("SECOND") == "bidirectional_GRU_stage":
        a = cfg["SECOND_ARGS"]
        cfg["SECOND"], a["units"] = "bidirectional_GRU_block", [int(a["units"])] * int(a["depth"])
    for key in ("SED", "DOA"):
        a = cfg.setdefault(key + "_ARGS", {})
        if cfg.get(key) == "simple_dense_stage":
            cfg[key], a["units"], a["dense_activation"] = "simple_dense_block", [int(a["units"])] * int(a["depth"]), a.get("activation", None)
        elif cfg.get(key) == "identity_block":
            cfg[key], a["units"] = "simple_dense_block", []
    return cfg


def _arch_from_config(model_config: dict, in_ch: int, n_freq: int, output_coupling: bool = False) -> _lib.Arch:
    model_config = canonical_config(model_config)
    for key, ok in (("FIRST", SUPPORTED_FIRST), ("SECOND", SUPPORTED_SECOND), ("SED", SUPPORTED_HEAD), ("DOA", SUPPORTED_HEAD)):
        if model_config.get(key) not in ok:
            raise ValueError(f"model_config[{key!r}]={model_config.get(key)!r}: only {ok} has MI355X kernels")
    fa = model_config["FIRST_ARGS"]
    xception = model_config["FIRST"] == "xception_block"
    resnet = model_config["FIRST"] == "resnet50_block"
    if resnet:
        # model_config/resnet50_gru.json:2-11; absent from the reference snapshot: spec/RESNET50_BLOCK.md is ours
        if int(fa["filters"]) != 32 or len(fa["block_num"]) != 4:
            raise ValueError("resnet50_block kernels are built for filters = 32 and four stages")
        filters, pools = [2 * int(fa["filters"])], [(5, 4)]
    elif xception:
        # model_config/xception_gru.json:2-11; the block is absent from the reference snapshot: spec/XCEPTION_BLOCK.md is ours
        # (entry conv2d_bn(2 filters) + MaxPool (5,4), block_num residual modules of 3 x [ReLU, SeparableConv2D, BN], ReLU + MaxPool (1,8))
        if int(fa["filters"]) != 32:
            raise ValueError("xception_block kernels are built for filters = 32 (width 64)")
        filters, pools = [2 * int(fa["filters"])], [(5, 4)]
    else:
        filters, pools = list(fa["filters"]), [tuple(p) for p in fa["pool_size"]]
    if len(filters) != len(pools):
        raise ValueError("filters and pool_size must have the same length")
    # dropout_rate of the FIRST / SECOND blocks (0.0 in every shipped config): simple_conv_block's Dropout behind each pool; Keras GRU
    # dropout = recurrent_dropout = rate (modules.py:306, 312-314).  The other FIRST blocks' specs have no Dropout: refuse instead of ignoring.
    conv_dropout, gru_dropout = float(fa.get("dropout_rate", 0.0) or 0.0), float(model_config["SECOND_ARGS"].get("dropout_rate", 0.0) or 0.0)
    for name, r in (("FIRST_ARGS", conv_dropout), ("SECOND_ARGS", gru_dropout)):
        if not 0.0 <= r < 1.0:
            raise ValueError(f"{name}['dropout_rate']={r!r}: [0, 1)")
    if conv_dropout and (xception or resnet):
        raise ValueError(f"{model_config['FIRST']} has no Dropout in its spec: FIRST_ARGS['dropout_rate'] must be 0")
    gru = list(model_config["SECOND_ARGS"]["units"])
    sed, doa = list(model_config["SED_ARGS"]["units"]), list(model_config["DOA_ARGS"]["units"])
    # simple_dense_block honours these keys (modules.py:350-376): `dense_activation` None (seldnet.json: what lets W1 W2 fold into one
    # product), relu, tanh or sigmoid, `kernel_size` (Conv1D 'same' over the frames of a clip), `dropout_rate`; anything else must fail
    # loudly instead of training a different network.  `kernel_regularizer` is accepted: it only feeds model.losses, which
    # train.trainstep (train.py:22-36) never adds to the objective.
    for key in ("SED_ARGS", "DOA_ARGS"):
        ha = model_config[key]
        if ha.get("dense_activation") not in _lib.SELD_ACT:
            raise ValueError(f"{key}['dense_activation']={ha.get('dense_activation')!r}: the head kernels implement {sorted(k for k in _lib.SELD_ACT if k)} and None")
        if not 1 <= int(ha.get("kernel_size", 1)) <= 15:
            raise ValueError(f"{key}['kernel_size']={ha.get('kernel_size')!r}: 1 .. 15")
        if not 0.0 <= float(ha.get("dropout_rate", 0)) < 1.0:
            raise ValueError(f"{key}['dropout_rate']={ha.get('dropout_rate')!r}: [0, 1)")
    for lst, name in ((filters, "filters"), (gru, "SECOND units"), (sed, "SED units"), (doa, "DOA units")):
        if len(lst) > _lib.MAX_LAYERS:
            raise ValueError(f"{name}: at most {_lib.MAX_LAYERS} layers")
    a = _lib.Arch()
    a.in_ch, a.n_freq = in_ch, n_freq
    a.n_conv = len(filters)
    for i, (f, p) in enumerate(zip(filters, pools)):
        a.filters[i], a.pool_t[i], a.pool_f[i] = f, p[0], p[1]
    a.n_gru = len(gru)
    for i, u in enumerate(gru):
        a.gru_units[i] = u
    a.n_sed_dense, a.n_doa_dense = len(sed), len(doa)
    for i, u in enumerate(sed):
        a.sed_units[i] = u
    for i, u in enumerate(doa):
        a.doa_units[i] = u
    # models.py:19 default 14; train.py:306-307 overrides to 12 before building
    a.n_classes = int(model_config.get("n_classes", 14))
    a.sed_dense_act = _lib.SELD_ACT[model_config["SED_ARGS"].get("dense_activation")]
    a.doa_dense_act = _lib.SELD_ACT[model_config["DOA_ARGS"].get("dense_activation")]
    a.sed_kernel_size = int(model_config["SED_ARGS"].get("kernel_size", 1))
    a.doa_kernel_size = int(model_config["DOA_ARGS"].get("kernel_size", 1))
    a.sed_dropout = float(model_config["SED_ARGS"].get("dropout_rate", 0))
    a.doa_dropout = float(model_config["DOA_ARGS"].get("dropout_rate", 0))
    a.output_coupling = 1 if output_coupling else 0
    a.conv_dropout, a.gru_dropout = conv_dropout, gru_dropout
    a.first_kind = 2 if resnet else (1 if xception else 0)
    a.xc_blocks = int(fa["block_num"]) if xception else 0
    if resnet:
        a.rn_filters = int(fa["filters"])
        for i, nb in enumerate(fa["block_num"]):
            a.rn_blocks[i] = int(nb)
    return a


class _DevPtr:
    """Exposes a raw device pointer owned by the HIP library to torch (zero copy)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 3}


class SeldNet:
    """The object `models.seldnet` returns: `model(x, training)`, `trainable_variables`,
    `get_weights/set_weights`, `summary()`, `save_weights/load_weights`."""

    def __init__(self, input_shape: Sequence[int], model_config: dict, device: int | None = None, dtype: str = "float32",
                 output_coupling: bool = False):
        if len(input_shape) != 4 or input_shape[0] is None:
            raise ValueError("input_shape must be [B, T, F, C] with a concrete batch size")
        B, T, F, Cc = (int(v) for v in input_shape)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SeldLibraryError("no HIP device visible: seld_amd has no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.arch = _arch_from_config(model_config, Cc, F, output_coupling)
        self.input_shape = (B, T, F, Cc)
        self.model_config = model_config
        ctx = C.c_void_p()
        if dtype not in ("float32", "bfloat16"):
            raise ValueError("dtype must be 'float32' (fp32-equivalent products, the default) or 'bfloat16' (one bf16 MFMA product per fp32 "
                             "product: BASELINE configs[1]'s \"bf16\", include/seld_hip.h SELD_DTYPE_BF16)")
        self.dtype = dtype
        _lib.check(self.lib.seld_create(C.byref(self.arch), B, T, _lib.SELD_DTYPE_BF16 if dtype == "bfloat16" else _lib.SELD_DTYPE_F32,
                                        self.device, C.byref(ctx)))
        self.ctx = ctx
        self.n_params = int(self.lib.seld_param_count(ctx))
        self.n_state = int(self.lib.seld_state_count(ctx))
        self.n_classes = int(self.arch.n_classes)
        st = T
        for i in range(self.arch.n_conv):
            st //= self.arch.pool_t[i]
        self.S = st
        self.variables = self._enumerate(1)
        self.state_variables = self._enumerate(0)
        self._dev = torch.device("cuda", self.device)
        self.set_weights(initial_weights(self.variables, seed=0), initial_state(self.state_variables))

    # ------------------------------------------------------------------ variables
    def _enumerate(self, trainable: int) -> List[Tuple[str, int, Tuple[int, ...]]]:
        out = []
        n = self.lib.seld_variable_count(self.ctx, trainable)
        for i in range(n):
            name = C.create_string_buffer(96)
            off, rank = C.c_int64(), C.c_int32()
            shape = (C.c_int64 * 4)()
            _lib.check(self.lib.seld_variable_info(self.ctx, trainable, i, name, 96, C.byref(off), C.byref(rank), shape), self.ctx)
            out.append((name.value.decode(), int(off.value), tuple(int(shape[k]) for k in range(rank.value))))
        return out

    @property
    def trainable_variables(self) -> Dict[str, np.ndarray]:
        flat = self.get_weights()[0]
        return {n: flat[o:o + int(np.prod(s))].reshape(s) for n, o, s in self.variables}

    def get_weights(self) -> Tuple[np.ndarray, np.ndarray]:
        w = np.empty(self.n_params, np.float32)
        s = np.empty(self.n_state, np.float32)
        _lib.check(self.lib.seld_get_weights_host(self.ctx, w.ctypes.data, self.n_params), self.ctx)
        _lib.check(self.lib.seld_get_state_host(self.ctx, s.ctypes.data, self.n_state), self.ctx)
        return w, s

    def set_weights(self, flat_w: np.ndarray, flat_state: np.ndarray | None = None) -> None:
        w = np.ascontiguousarray(flat_w, np.float32)
        _lib.check(self.lib.seld_set_weights_host(self.ctx, w.ctypes.data, w.size), self.ctx)
        if flat_state is not None:
            s = np.ascontiguousarray(flat_state, np.float32)
            _lib.check(self.lib.seld_set_state_host(self.ctx, s.ctypes.data, s.size), self.ctx)

    def set_option(self, key: str, value: int) -> None:
        """Kernel-selection knobs of the C library (`seld_set_option`): "conv64_split_bf16", "gemm_split_bf16", "conv1_split_bf16",
        "conv1_pool_fused", "conv1_gram", "conv64_dbuf", "gru_wgrad_batch", "xc_fused_fwd", "xc_fused_pw_bwd", "xc_fused_dw_bwd", "xc_fused_bn_sums", "xc_w16", "xc_xcd_map", "xc_wgrad_side", "rn_split_bf16", "rn_wgrad_side", "rn_epi_stats", "rn_epi_add"."""
        _lib.check(self.lib.seld_set_option(self.ctx, key.encode(), int(value)), self.ctx)

    def get_grads(self) -> np.ndarray:
        g = np.empty(self.n_params, np.float32)
        _lib.check(self.lib.seld_get_grads_host(self.ctx, g.ctypes.data, self.n_params), self.ctx)
        return g

    def grad_tensor(self) -> torch.Tensor:
        """The flat gradient buffer as a torch CUDA tensor (zero copy) — the DP all-reduce operand."""
        return torch.as_tensor(_DevPtr(int(self.lib.seld_grad_ptr(self.ctx)), self.n_params), device=self._dev)

    def param_tensor(self) -> torch.Tensor:
        return torch.as_tensor(_DevPtr(int(self.lib.seld_param_ptr(self.ctx)), self.n_params), device=self._dev)

    def save_weights(self, path: str) -> None:
        """Counterpart of tf.keras.models.save_model(..., include_optimizer=False) (train.py:377-380):
        an .npz keyed by variable name (Keras HDF5 needs h5py, which this image lacks)."""
        w, s = self.get_weights()
        d = {n: w[o:o + int(np.prod(sh))].reshape(sh) for n, o, sh in self.variables}
        d.update({n: s[o:o + int(np.prod(sh))].reshape(sh) for n, o, sh in self.state_variables})
        np.savez(path, **d)

    def load_weights(self, path: str) -> None:
        z = np.load(path)
        w, s = self.get_weights()
        for n, o, sh in self.variables:
            w[o:o + int(np.prod(sh))] = np.asarray(z[n], np.float32).reshape(-1)
        for n, o, sh in self.state_variables:
            s[o:o + int(np.prod(sh))] = np.asarray(z[n], np.float32).reshape(-1)
        self.set_weights(w, s)

    def summary(self) -> str:
        lines = [f"SeldNet input {self.input_shape} -> sed [B,{self.S},{self.n_classes}], doa [B,{self.S},{3 * self.n_classes}]"]
        for n, o, sh in self.variables:
            lines.append(f"  {n:32s} {str(sh):20s} {int(np.prod(sh)):8d}")
        lines.append(f"Trainable params: {self.n_params}; non-trainable: {self.n_state}")
        text = "\n".join(lines)
        print(text)
        return text

    # ------------------------------------------------------------------ calls
    def _prep(self, x: torch.Tensor) -> torch.Tensor:
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            x = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self._dev)
        x = x.to(torch.float32).contiguous()
        Bm, T, F, Cc = self.input_shape
        if tuple(x.shape[1:]) != (T, F, Cc) or not (1 <= x.shape[0] <= Bm):
            raise ValueError(f"x shape {tuple(x.shape)} incompatible with model input {self.input_shape}")
        _lib.check(self.lib.seld_set_batch(self.ctx, int(x.shape[0])), self.ctx)
        _lib.check(self.lib.seld_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)), self.ctx)
        return x

    def _outputs(self, B: int):
        sed = torch.empty((B, self.S, self.n_classes), dtype=torch.float32, device=self._dev)
        doa = torch.empty((B, self.S, 3 * self.n_classes), dtype=torch.float32, device=self._dev)
        return sed, doa

    def __call__(self, x, training: bool = False):
        x = self._prep(x)
        sed, doa = self._outputs(x.shape[0])
        _lib.check(self.lib.seld_forward(self.ctx, x.data_ptr(), sed.data_ptr(), doa.data_ptr(), int(bool(training))), self.ctx)
        return [sed, doa]

    def close(self) -> None:
        """Destroy the HIP context now (and its RCCL communicator, if the library owns one): call it on every rank before the host's process
        group is torn down, instead of leaving it to interpreter shutdown."""
        self.__del__()

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.seld_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass


COMPOSED_FIRST = ("mother_block", "mother_stage")      # seld_amd/modules.py: composed from the module operators, no fused ctx


def seldnet(input_shape, model_config, device=None, dtype: str = "float32"):
    """reference models.seldnet (models.py:18-32).  `dtype="bfloat16"`: bf16 single-product mode (SELD_DTYPE_BF16).  FIRST = mother_block /
    mother_stage (modules.py:15-43, 184-298) builds a modules.ComposedSeldNet (layer-by-layer module operators) instead of a fused ctx."""
    if model_config.get("FIRST") in COMPOSED_FIRST:
        if dtype != "float32":
            raise ValueError("composed models compute in float32")
        from .modules import ComposedSeldNet
        return ComposedSeldNet(input_shape, model_config, device)
    return SeldNet(input_shape, model_config, device, dtype)


def seldnet_v1(input_shape, model_config, device=None, dtype: str = "float32") -> SeldNet:
    """reference models.seldnet_v1 (models.py:36-52; model_config/seldnet_v1.json): seldnet whose DOA output is coupled to the SED
    output, doa_out = tanh(doa * Concatenate([sed] * 3)) (seld_arch.output_coupling)."""
    return SeldNet(input_shape, model_config, device, dtype, output_coupling=True)


# ---------------------------------------------------------------------- Keras initialisers
def initial_weights(variables, seed: int = 0) -> np.ndarray:
    """Keras default initialisers: glorot_uniform kernels, orthogonal GRU recurrent kernels,
    zero biases, BatchNorm gamma = 1 / beta = 0 (numpy default_rng(seed))."""
    rng = np.random.default_rng(seed)
    n = sum(int(np.prod(s)) for _, _, s in variables)
    flat = np.zeros(n, np.float32)
    for name, off, shape in variables:
        size = int(np.prod(shape))
        if name.endswith("recurrent_kernel"):
            rows, cols = shape
            a = rng.standard_normal((max(rows, cols), min(rows, cols)))
            q, r = np.linalg.qr(a)
            q = q * np.sign(np.diag(r))
            v = (q.T if rows < cols else q)[:rows, :cols]
        elif name.endswith("kernel"):
            receptive = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in, fan_out = receptive * shape[-2], receptive * shape[-1]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            v = rng.uniform(-lim, lim, size)
        elif name.endswith("gamma"):
            v = np.ones(size)
        else:
            v = np.zeros(size)
        flat[off:off + size] = np.asarray(v, np.float32).reshape(-1)
    return flat


def initial_state(state_variables) -> np.ndarray:
    n = sum(int(np.prod(s)) for _, _, s in state_variables)
    flat = np.zeros(n, np.float32)
    for name, off, shape in state_variables:
        if name.endswith("moving_variance"):
            flat[off:off + int(np.prod(shape))] = 1.0
    return flat
