"""Host-side mirror of the reference's modules.py for the blocks that have no fused kernels of their own: a model is COMPOSED here,
layer by layer, from the module operators of libseld_hip.so (include/seld_hip.h: seld_m_*, plus seld_k_gru_* / seld_k_losses /
seld_k_adam) — the way the reference composes Keras layers (modules.py, models.py:18-32).  Every arithmetic operation is a HIP kernel
behind the C ABI; this file holds shapes, the order of calls and the buffers (torch tensors: device memory and the stream, nothing
else).  There is no CPU or PyTorch fallback.

  mother_block  (reference modules.py:184-298)   Conv2D(k,'same') + BatchNormalization (+ skips: identity / 1x1 Conv2D+BN projection)
                                                  + Activation, three times (strides on the second), concatenation where a layer is
                                                  skipped, squeeze-and-excitation tail; the ValueErrors of modules.py:202-222
  mother_stage  (modules.py:15-43)                `depth` mother_blocks, strides in the first only
  bidirectional_GRU_block (modules.py:302-319), simple_dense_block (modules.py:350-376; Conv1D kernel_size 1): at ANY feature width

`ComposedSeldNet` = models.seldnet(input_shape, model_config) for FIRST in {mother_block, mother_stage}: same surface as
seld_amd.models.SeldNet (variables in Keras creation order, get / set_weights, __call__, train.trainstep / teststep).  The three
BASELINE configurations do NOT run through here — their blocks are fused kernels inside a seld_ctx (models.SeldNet)."""
from __future__ import annotations

import copy
import ctypes as C
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

ACT = {None: 0, "linear": 0, "sigmoid": 1, "tanh": 2, "relu": 3, "swish": 4}
BN_EPS, BN_MOMENTUM = 1e-3, 0.99


def safe_tuple(v, n=2):
    return tuple(int(a) for a in v) if isinstance(v, (list, tuple)) else (int(v),) * n


def check_mother_config(cfg: dict) -> None:
    """reference modules.py:202-222"""
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
    for key in ("activation", "se_activation"):
        if cfg.get(key, "relu") not in ACT:
            raise ValueError(f"{key} {cfg.get(key)!r}: the module operators know {sorted(k for k in ACT if k)}")


def dead_layers(cfg: dict):
    """(first layer dead, second layer dead): a layer whose output reaches neither the block's output nor a later layer — configurations the
    reference accepts (modules.py:202-222) and whose Keras functional model simply does not contain the layer: no variables, no computation"""
    f = [int(cfg[f"filters{i}"]) for i in range(3)]
    c1, c2 = cfg["connect1"], cfg["connect2"]
    need2 = f[2] > 0 or c2[2] == 1
    need1 = c2[1] == 1 or (need2 and (f[1] > 0 or c1[1] == 1))
    return (f[0] > 0 and not need1), (not need2)


class _Rt:
    """What every layer shares: the library, the device, the variable store and the operator calls (all on torch's current stream)."""

    def __init__(self, device):
        self.lib = _lib.load()
        self.dev = device
        self.tr: List[Tuple[str, Tuple[int, ...]]] = []
        self.nt: List[Tuple[str, Tuple[int, ...]]] = []
        self._views: Dict[str, torch.Tensor] = {}
        self.params = self.grads = self.state = None
        self._slab = None
        self._bn_scratch = None

    # ---- variables
    def var(self, name, shape, trainable=True):
        (self.tr if trainable else self.nt).append((name, tuple(int(s) for s in shape)))

    def finalize(self):
        n = sum(int(np.prod(s)) for _, s in self.tr)
        ns = sum(int(np.prod(s)) for _, s in self.nt)
        z = lambda k: torch.zeros(max(k, 1), dtype=torch.float32, device=self.dev)
        self.params, self.grads, self.adam_m, self.adam_v, self.state = z(n), z(n), z(n), z(n), z(ns)
        off = 0
        self.variables, self.state_variables = [], []
        for name, sh in self.tr:
            k = int(np.prod(sh))
            self.variables.append((name, off, sh))
            self._views[name] = self.params[off:off + k]
            self._views["d:" + name] = self.grads[off:off + k]
            off += k
        off = 0
        for name, sh in self.nt:
            k = int(np.prod(sh))
            self.state_variables.append((name, off, sh))
            self._views[name] = self.state[off:off + k]
            off += k
        self.n_params, self.n_state = n, ns

    def w(self, name):
        return self._views[name]

    def g(self, name):
        return self._views["d:" + name]

    def empty(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def bn_scratch(self, C):
        """the BatchNormalization sums' first-stage partials: one buffer per channel count, shared by every layer (calls are in stream order)"""
        need = int(self.lib.seld_m_bn_scratch(int(C)))
        if self._bn_scratch is None or self._bn_scratch.numel() < need:
            self._bn_scratch = self.empty(need)        # (torch allocations are 256-byte aligned: the kernels read it as doubles)
        return self._bn_scratch

    # ---- operator calls
    def st(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    @staticmethod
    def p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def ck(self, rc):
        _lib.check(rc)

    def gemm(self, A, Bm, bias, Cm, M, N, K, transb=0, accumulate=0):
        self.ck(self.lib.seld_m_gemm(self.p(A), self.p(Bm), self.p(bias), self.p(Cm), M, N, K, transb, accumulate, self.st()))

    def gemm_tn(self, A, Bm, Cm, colsum, M, K1, N, seq=0, shift=0):
        need = int(self.lib.seld_m_gemm_tn_scratch(K1, N))
        if self._slab is None or self._slab.numel() < need:
            self._slab = self.empty(need)
        self.ck(self.lib.seld_m_gemm_tn(self.p(A), self.p(Bm), self.p(Cm), self.p(colsum), self.p(self._slab), M, K1, N, seq, shift, self.st()))

    def act(self, x, y, kind):
        self.ck(self.lib.seld_m_act(self.p(x), self.p(y), x.numel(), kind, self.st()))

    def act_bwd(self, x, dy, dx, kind, accumulate=0):
        self.ck(self.lib.seld_m_act_bwd(self.p(x), self.p(dy), self.p(dx), x.numel(), kind, accumulate, self.st()))

    def axpy(self, dst, src, alpha=1.0):
        self.ck(self.lib.seld_m_axpy(self.p(dst), self.p(src), dst.numel(), alpha, self.st()))



class Conv2D:
    """tf.keras.layers.Conv2D(filters, k, strides, padding='same', use_bias=True): im2col (skipped for 1x1 stride 1) + the MFMA GEMM."""

    def __init__(self, rt: _Rt, name: str, in_shape, filters: int, k: int, strides=(1, 1), B: int = 1):
        self.rt, self.name, self.k, self.s = rt, name, int(k), tuple(strides)
        self.H, self.W, self.Cin = in_shape
        self.N = int(filters)
        self.Ho, self.Wo = -(-self.H // self.s[0]), -(-self.W // self.s[1])
        self.K = self.k * self.k * self.Cin
        self.out_shape = (self.Ho, self.Wo, self.N)
        rt.var(f"{name}.kernel", (self.k, self.k, self.Cin, self.N))
        rt.var(f"{name}.bias", (self.N,))
        self.direct = self.k == 1 and self.s == (1, 1)
        self.B = B
        self.col = None if self.direct else rt.empty(B * self.Ho * self.Wo, self.K)
        self.dcol = None if self.direct else rt.empty(B * self.Ho * self.Wo, self.K)
        self.z = rt.empty(B, self.Ho, self.Wo, self.N)

    def forward(self, x, B):
        rt = self.rt
        M = B * self.Ho * self.Wo
        if self.direct:
            self.a = x
        else:
            rt.ck(rt.lib.seld_m_im2col(rt.p(x), rt.p(self.col), B, self.H, self.W, self.Cin, self.k, self.k, self.s[0], self.s[1], rt.st()))
            self.a = self.col
        rt.gemm(self.a, rt.w(f"{self.name}.kernel"), rt.w(f"{self.name}.bias"), self.z, M, self.N, self.K)
        return self.z[:B]

    def backward(self, dz, dx, B, accumulate):
        """dz [B,Ho,Wo,N] -> kernel / bias gradients; dx (+)= the input gradient (dx None: not needed)"""
        rt = self.rt
        M = B * self.Ho * self.Wo
        rt.gemm_tn(self.a, dz, rt.g(f"{self.name}.kernel"), rt.g(f"{self.name}.bias"), M, self.K, self.N)
        if dx is None:
            return
        if self.direct:
            rt.gemm(dz, rt.w(f"{self.name}.kernel"), None, dx, M, self.K, self.N, transb=1, accumulate=int(accumulate))
        else:
            rt.gemm(dz, rt.w(f"{self.name}.kernel"), None, self.dcol, M, self.K, self.N, transb=1)
            rt.ck(rt.lib.seld_m_col2im(rt.p(self.dcol), rt.p(dx), B, self.H, self.W, self.Cin, self.k, self.k, self.s[0], self.s[1],
                                       int(accumulate), rt.st()))


class BatchNorm:
    """tf.keras.layers.BatchNormalization() on the channel axis: batch statistics in training (moving statistics updated), the moving ones
    otherwise."""

    def __init__(self, rt: _Rt, name: str, shape, B: int):
        self.rt, self.name = rt, name
        self.H, self.W, self.C = shape
        rt.var(f"{name}.gamma", (self.C,))
        rt.var(f"{name}.beta", (self.C,))
        rt.var(f"{name}.moving_mean", (self.C,), trainable=False)
        rt.var(f"{name}.moving_variance", (self.C,), trainable=False)
        self.mean, self.var = rt.empty(self.C), rt.empty(self.C)
        self.scratch = rt.bn_scratch(self.C)

    def forward(self, z, out, B, training, accumulate):
        rt, n = self.rt, self.name
        npix = B * self.H * self.W
        self.z = z
        if training:
            rt.ck(rt.lib.seld_m_bn_stats(rt.p(z), npix, self.C, rt.p(self.mean), rt.p(self.var), rt.p(self.scratch), rt.st()))
            mean, var = self.mean, self.var
            rt.ck(rt.lib.seld_m_bn_moving(rt.p(mean), rt.p(var), rt.p(rt.w(f"{n}.moving_mean")), rt.p(rt.w(f"{n}.moving_variance")), self.C,
                                          BN_MOMENTUM, npix, rt.st()))
        else:
            mean, var = rt.w(f"{n}.moving_mean"), rt.w(f"{n}.moving_variance")
        rt.ck(rt.lib.seld_m_bn_apply(rt.p(z), rt.p(mean), rt.p(var), rt.p(rt.w(f"{n}.gamma")), rt.p(rt.w(f"{n}.beta")), BN_EPS, rt.p(out), npix,
                                     self.C, int(accumulate), rt.st()))

    def backward(self, dy, dz, B):
        rt, n = self.rt, self.name
        rt.ck(rt.lib.seld_m_bn_bwd(rt.p(self.z), rt.p(dy), rt.p(self.mean), rt.p(self.var), rt.p(rt.w(f"{n}.gamma")), BN_EPS, rt.p(dz),
                                   rt.p(rt.g(f"{n}.gamma")), rt.p(rt.g(f"{n}.beta")), B * self.H * self.W, self.C, rt.p(self.scratch), rt.st()))


class _ConvBN:
    """Conv2D + BatchNormalization: the pair every branch of mother_block is made of"""

    def __init__(self, rt, cname, bname, in_shape, filters, k, strides, B):
        self.conv = Conv2D(rt, cname, in_shape, filters, k, strides, B)
        self.bn = BatchNorm(rt, bname, self.conv.out_shape, B)
        self.out_shape = self.conv.out_shape
        self.dz = rt.empty(B, *self.out_shape)

    def forward(self, x, out, B, training, accumulate):
        self.bn.forward(self.conv.forward(x, B), out, B, training, accumulate)

    def backward(self, dy, dx, B, accumulate):
        self.bn.backward(dy, self.dz, B)
        self.conv.backward(self.dz, dx, B, accumulate)


class MotherBlock:
    """reference modules.mother_block (modules.py:184-298)."""

    def __init__(self, rt: _Rt, cfg: dict, in_shape, prefix: str, B: int):
        check_mother_config(cfg)
        self.rt, self.cfg, self.B = rt, cfg, B
        f = [int(cfg[f"filters{i}"]) for i in range(3)]
        k = [int(cfg[f"kernel_size{i}"]) for i in range(3)]
        conn = [list(cfg["connect0"]), list(cfg["connect1"]), list(cfg["connect2"])]
        strides = safe_tuple(cfg.get("strides", (1, 1)))
        self.act = ACT[cfg.get("activation", "relu")]
        self.f, self.conn, self.strides = f, conn, strides
        shapes = [tuple(in_shape)]
        self.layers = []      # per layer: dict(kind='conv'|'alias'|'cat'|'dead', ...)
        dead = dead_layers(cfg) + (False,)
        for L in range(3):
            src = shapes[-1]
            n_in = L + 1      # outputs[0..L] exist
            lay = {"L": L}
            if dead[L]:
                # the layer's output reaches nothing: Keras' functional model does not contain it: no variables, no computation
                lay.update(kind="dead")
                out = src if f[L] == 0 else (-(-src[0] // (strides[0] if L == 1 else 1)), -(-src[1] // (strides[1] if L == 1 else 1)), f[L])
            elif f[L] > 0:
                s = strides if L == 1 else (1, 1)
                main = _ConvBN(rt, f"{prefix}.c{L}", f"{prefix}.bn{L}", src, f[L], k[L], s, B)
                out = main.out_shape
                skips = []
                for i in range(n_in):
                    if conn[L][i] != 1:
                        continue
                    if shapes[i] != out:
                        ps = (1, 1) if (L == 2 and i == 2) or L == 0 else strides
                        skips.append((i, _ConvBN(rt, f"{prefix}.p{L}_{i}", f"{prefix}.pbn{L}_{i}", shapes[i], f[L], 1, ps, B)))
                    else:
                        skips.append((i, None))
                lay.update(kind="conv", main=main, skips=skips, pre=rt.empty(B, *out), y=rt.empty(B, *out), dpre=rt.empty(B, *out))
            elif L == 0:
                out = src
                lay.update(kind="alias")
            else:
                parts = []
                for i in range(n_in):
                    if conn[L][i] != 1:
                        continue
                    s = shapes[i]
                    cv = None
                    if L == 2 and conn[2][-1] == 1 and strides != (1, 1) and i < 2:
                        cv = Conv2D(rt, f"{prefix}.s2_{i}", s, s[2], 1, strides, B)      # connect with strided outputs
                        s = cv.out_shape
                    parts.append((i, cv, s))
                if len({p[2][:2] for p in parts}) != 1:
                    raise ValueError("mother_block: the concatenated tensors differ in their spatial extents")
                out = (parts[0][2][0], parts[0][2][1], sum(p[2][2] for p in parts))
                lay.update(kind="cat", parts=parts, y=rt.empty(B, *out), tmp=rt.empty(B, *out))
            lay["out"] = out
            self.layers.append(lay)
            if L < 2:
                shapes.append(out)
        self.shapes = shapes
        self.out_shape = self.layers[2]["out"]
        sq = float(cfg.get("squeeze_ratio", 0))
        self.se = None
        if sq > 0:
            Cc = self.out_shape[2]
            sf = int(sq * Cc)
            rt.var(f"{prefix}.se0.kernel", (1, 1, Cc, sf)); rt.var(f"{prefix}.se0.bias", (sf,))
            rt.var(f"{prefix}.se1.kernel", (1, 1, sf, Cc)); rt.var(f"{prefix}.se1.bias", (Cc,))
            self.se = {"p": prefix, "sf": sf, "act": ACT[cfg.get("se_activation", "relu")], "m": rt.empty(B, Cc), "a1": rt.empty(B, sf),
                       "s1": rt.empty(B, sf), "a2": rt.empty(B, Cc), "s2": rt.empty(B, Cc), "y": rt.empty(B, *self.out_shape),
                       "ds2": rt.empty(B, Cc), "da2": rt.empty(B, Cc), "ds1": rt.empty(B, sf), "da1": rt.empty(B, sf), "dm": rt.empty(B, Cc),
                       "din": rt.empty(B, *self.out_shape)}
        # gradients w.r.t. outputs[0..2] (outputs[0] = the block input) are accumulated here during backward
        self.gout = [rt.empty(B, *s) for s in shapes]

    # ---------------------------------------------------------------- forward
    def forward(self, x, B, training):
        rt = self.rt
        outputs = [x]
        res = None
        for lay in self.layers:
            L = lay["L"]
            if lay["kind"] == "dead":
                y = None
            elif lay["kind"] == "alias":
                y = outputs[-1]
            elif lay["kind"] == "conv":
                pre = lay["pre"][:B]
                lay["main"].forward(outputs[-1], pre, B, training, accumulate=0)
                for i, proj in lay["skips"]:
                    if proj is None:
                        rt.axpy(pre, outputs[i])
                    else:
                        proj.forward(outputs[i], pre, B, training, accumulate=1)
                y = lay["y"][:B]
                rt.act(pre, y, self.act)
            else:
                y = lay["y"][:B]
                Cd, off = lay["out"][2], 0
                rows = B * lay["out"][0] * lay["out"][1]
                for i, cv, s in lay["parts"]:
                    src = outputs[i] if cv is None else cv.forward(outputs[i], B)
                    rt.ck(rt.lib.seld_m_copy_channels(rt.p(src), rt.p(y), rows, s[2], Cd, off, 0, rt.st()))
                    off += s[2]
            if L < 2:
                outputs.append(y)
            else:
                res = y
        self.outputs = outputs
        if self.se is not None:
            se = self.se
            Hh, Ww, Cc = self.out_shape
            p_ = se["p"]
            self.se_in = res
            rt.ck(rt.lib.seld_m_mean_hw(rt.p(res), rt.p(se["m"]), B, Hh * Ww, Cc, rt.st()))
            rt.gemm(se["m"], rt.w(f"{p_}.se0.kernel"), rt.w(f"{p_}.se0.bias"), se["a1"], B, se["sf"], Cc)
            rt.act(se["a1"][:B], se["s1"][:B], se["act"])
            rt.gemm(se["s1"], rt.w(f"{p_}.se1.kernel"), rt.w(f"{p_}.se1.bias"), se["a2"], B, Cc, se["sf"])
            rt.act(se["a2"][:B], se["s2"][:B], ACT["sigmoid"])
            rt.ck(rt.lib.seld_m_scale_hw(rt.p(res), rt.p(se["s2"]), rt.p(se["y"]), B, Hh * Ww, Cc, rt.st()))
            res = se["y"][:B]
        return res

    # ---------------------------------------------------------------- backward
    def backward(self, dy, dx, B, need_dx=True):
        """dy: gradient w.r.t. the block output; dx (= the caller's buffer for the block input's gradient, overwritten) or None"""
        rt = self.rt
        if self.se is not None:
            se = self.se
            Hh, Ww, Cc = self.out_shape
            p_ = se["p"]
            rt.ck(rt.lib.seld_m_scale_hw_bwd_ds(rt.p(self.se_in), rt.p(dy), rt.p(se["ds2"]), B, Hh * Ww, Cc, rt.st()))
            rt.act_bwd(se["a2"][:B], se["ds2"][:B], se["da2"][:B], ACT["sigmoid"])
            rt.gemm_tn(se["s1"], se["da2"], rt.g(f"{p_}.se1.kernel"), rt.g(f"{p_}.se1.bias"), B, se["sf"], Cc)
            rt.gemm(se["da2"], rt.w(f"{p_}.se1.kernel"), None, se["ds1"], B, se["sf"], Cc, transb=1)
            rt.act_bwd(se["a1"][:B], se["ds1"][:B], se["da1"][:B], se["act"])
            rt.gemm_tn(se["m"], se["da1"], rt.g(f"{p_}.se0.kernel"), rt.g(f"{p_}.se0.bias"), B, Cc, se["sf"])
            rt.gemm(se["da1"], rt.w(f"{p_}.se0.kernel"), None, se["dm"], B, Cc, se["sf"], transb=1)
            rt.ck(rt.lib.seld_m_scale_hw_bwd_dx(rt.p(dy), rt.p(se["s2"]), rt.p(se["dm"]), rt.p(se["din"]), B, Hh * Ww, Cc, 0, rt.st()))
            dy = se["din"][:B]
        # gradient slots of outputs[0..2]; outputs[1] may alias outputs[0] (first layer skipped): then they share one slot
        alias0 = self.layers[0]["kind"] == "alias"
        slots = [self.gout[0][:B], self.gout[0][:B] if alias0 else self.gout[1][:B], self.gout[2][:B]]
        written = [False, False, False]

        def slot_of(i):
            return 0 if (i == 1 and alias0) else i

        def add_to(i, src):
            j = slot_of(i)
            if written[j]:
                rt.axpy(slots[j], src)
            else:
                slots[j].copy_(src)       # device-to-device copy on the current stream
                written[j] = True

        for lay in reversed(self.layers):
            L = lay["L"]
            d_out = dy if L == 2 else (slots[slot_of(L + 1)] if written[slot_of(L + 1)] else None)
            if lay["kind"] in ("alias", "dead"):
                continue          # outputs[1] IS outputs[0]: its gradient already sits in slot 0 / a dead layer has neither variables nor a gradient
            if d_out is None:
                raise RuntimeError("mother_block: a layer's output reaches nothing (the configuration checks should have refused it)")
            src_i = L             # the layer's main input is outputs[L]
            if lay["kind"] == "conv":
                dpre = lay["dpre"][:B]
                rt.act_bwd(lay["pre"][:B], d_out, dpre, self.act)
                j = slot_of(src_i)
                lay["main"].backward(dpre, slots[j], B, accumulate=written[j])
                written[j] = True
                for i, proj in lay["skips"]:
                    if proj is None:
                        add_to(i, dpre)
                    else:
                        jj = slot_of(i)
                        proj.backward(dpre, slots[jj], B, accumulate=written[jj])
                        written[jj] = True
            else:
                Cd, off = lay["out"][2], 0
                rows = B * lay["out"][0] * lay["out"][1]
                for i, cv, s in lay["parts"]:
                    if cv is None:
                        jj = slot_of(i)
                        if not written[jj]:
                            slots[jj].zero_()
                            written[jj] = True
                        rt.ck(rt.lib.seld_m_copy_channels(rt.p(slots[jj]), rt.p(d_out), rows, s[2], Cd, off, 1, rt.st()))
                    else:
                        piece = lay["tmp"][:B].reshape(-1)[:rows * s[2]].view(B, s[0], s[1], s[2])
                        piece.zero_()
                        rt.ck(rt.lib.seld_m_copy_channels(rt.p(piece), rt.p(d_out), rows, s[2], Cd, off, 1, rt.st()))
                        jj = slot_of(i)
                        cv.backward(piece, slots[jj], B, accumulate=written[jj])
                        written[jj] = True
                    off += s[2]
        if dx is not None:
            dx.copy_(slots[0])
        return slots[0]


class ComposedSeldNet:
    """models.seldnet(input_shape, model_config) (reference models.py:18-32) composed from module operators: FIRST = mother_block |
    mother_stage, SECOND = bidirectional_GRU_block (units 128: the recurrence kernels), SED / DOA = simple_dense_block (kernel_size 1)."""

    def __init__(self, input_shape, model_config: dict, device=None):
        from .models import canonical_config
        if not torch.cuda.is_available():
            raise RuntimeError("seld_amd needs a HIP device: there is no CPU fallback")
        cfg = canonical_config(model_config)
        if cfg.get("SECOND") != "bidirectional_GRU_block" or cfg.get("SED") != "simple_dense_block" or cfg.get("DOA") != "simple_dense_block":
            raise ValueError("composed models: SECOND = bidirectional_GRU_block, SED / DOA = simple_dense_block")
        self._dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        B, T, Fq, Ch = (int(v) for v in input_shape)
        self.input_shape = (B, T, Fq, Ch)
        self.n_classes = int(cfg.get("n_classes", 14))
        rt = self.rt = _Rt(self._dev)
        self.lib = rt.lib
        fa = cfg["FIRST_ARGS"]
        if cfg["FIRST"] == "mother_stage":
            cfgs, c = [], copy.deepcopy(fa)
            for _ in range(int(fa["depth"])):
                cfgs.append(copy.deepcopy(c))
                c["strides"] = (1, 1)       # modules.py:41
        elif cfg["FIRST"] == "mother_block":
            cfgs = [copy.deepcopy(fa)]
        else:
            raise ValueError(f"composed models: FIRST = mother_block | mother_stage, not {cfg['FIRST']!r}")
        shape = (T, Fq, Ch)
        self.blocks = []
        for d, c in enumerate(cfgs):
            blk = MotherBlock(rt, c, shape, f"mb{d}", B)
            self.blocks.append(blk)
            shape = blk.out_shape
        self.S = shape[0]
        feat = shape[1] * shape[2]
        sa = cfg["SECOND_ARGS"]
        if float(sa.get("dropout_rate", 0.0)) != 0.0:
            raise ValueError("GRU dropout is not implemented")
        self.gru = []
        fin = feat
        for i, u in enumerate(sa["units"]):
            if int(u) != 128:
                raise ValueError("the recurrence kernels are built for 128 units")
            for dn in ("fwd", "bwd"):
                rt.var(f"gru{i}.{dn}.kernel", (fin, 384)); rt.var(f"gru{i}.{dn}.recurrent_kernel", (128, 384)); rt.var(f"gru{i}.{dn}.bias", (2, 384))
            R = B * self.S
            self.gru.append({"in": fin, "gx": [rt.empty(R, 384) for _ in range(2)], "h": [rt.empty(R, 128) for _ in range(2)],
                             "sv": [rt.empty(R, 512) for _ in range(2)], "out": rt.empty(R, 128), "dgx": [rt.empty(R, 384) for _ in range(2)],
                             "dgh": [rt.empty(R, 384) for _ in range(2)], "din": rt.empty(R, fin)})
            fin = 128
        self.heads = []
        for head, key, n_out in (("sed", "SED_ARGS", self.n_classes), ("doa", "DOA_ARGS", 3 * self.n_classes)):
            ha = cfg[key]
            if int(ha.get("kernel_size", 1)) != 1 or float(ha.get("dropout_rate", 0)) != 0:
                raise ValueError("composed heads: kernel_size 1, no dropout")
            hact = ha.get("dense_activation", None)
            if hact not in ACT:
                raise ValueError(f"dense_activation {hact!r}")
            a = fin
            lays = []
            for j, u in enumerate(ha.get("units", [])):
                rt.var(f"{head}.dense{j}.kernel", (1, a, int(u))); rt.var(f"{head}.dense{j}.bias", (int(u),))
                lays.append({"n": f"{head}.dense{j}", "in": a, "out": int(u), "pre": rt.empty(B * self.S, int(u)), "y": rt.empty(B * self.S, int(u)),
                             "dpre": rt.empty(B * self.S, int(u)), "dy": rt.empty(B * self.S, int(u))})
                a = int(u)
            rt.var(f"{head}.out.kernel", (a, n_out)); rt.var(f"{head}.out.bias", (n_out,))
            self.heads.append({"name": head, "hact": ACT[hact], "layers": lays, "in": a, "out": n_out, "pre": rt.empty(B * self.S, n_out),
                               "dpre": rt.empty(B * self.S, n_out), "act": ACT["sigmoid"] if head == "sed" else ACT["tanh"]})
        rt.finalize()
        self.variables, self.state_variables = rt.variables, rt.state_variables
        self.n_params, self.n_state = rt.n_params, rt.n_state
        self.dfeat = rt.empty(B * self.S, 128)
        self.loss_scratch = rt.empty(int(rt.lib.seld_m_losses_scratch(B * self.S)))
        self.dfirst = rt.empty(B, *shape)
        self.adam_step = 0
        self._marks = None          # profile(True): [(group name, torch event)] of the current step (bench.py's mother_stage record)
        self._init_weights()

    # ---------------------------------------------------------------- weights
    def _init_weights(self, seed: int = 0):
        rng = np.random.default_rng(seed)
        w = np.zeros(self.n_params, np.float32)
        for n, off, sh in self.variables:
            k = int(np.prod(sh))
            if n.endswith("recurrent_kernel"):
                w[off:off + k] = np.concatenate([np.linalg.qr(rng.standard_normal((128, 128)))[0] for _ in range(3)], axis=1).reshape(-1)
            elif n.endswith("kernel"):
                fan_in = int(np.prod(sh[:-1])); fan_out = int(sh[-1]) * (int(np.prod(sh[:-2])) if len(sh) > 2 else 1)
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                w[off:off + k] = rng.uniform(-lim, lim, k)
            elif n.endswith("gamma"):
                w[off:off + k] = 1.0
        s = np.zeros(max(self.n_state, 1), np.float32)
        for n, off, sh in self.state_variables:
            if n.endswith("moving_variance"):
                s[off:off + int(np.prod(sh))] = 1.0
        self.set_weights(w, s[:self.n_state])

    def get_weights(self):
        return self.rt.params[:self.n_params].cpu().numpy().copy(), self.rt.state[:self.n_state].cpu().numpy().copy()

    def set_weights(self, w, state=None):
        self.rt.params[:self.n_params].copy_(torch.as_tensor(np.asarray(w, np.float32)))
        if state is not None and self.n_state:
            self.rt.state[:self.n_state].copy_(torch.as_tensor(np.asarray(state, np.float32)))

    def get_grads(self):
        return self.rt.grads[:self.n_params].cpu().numpy().copy()

    def summary(self) -> str:
        lines = [f"ComposedSeldNet input {self.input_shape} -> sed [B,{self.S},{self.n_classes}], doa [B,{self.S},{3 * self.n_classes}]"]
        lines += [f"  {n:32s} {str(sh):20s} {int(np.prod(sh)):8d}" for n, _, sh in self.variables]
        lines.append(f"Trainable params: {self.n_params}; non-trainable: {self.n_state}")
        text = "\n".join(lines)
        print(text)
        return text

    def save_weights(self, path: str) -> None:
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

    def close(self) -> None:
        pass

    # ---------------------------------------------------------------- measurement aid
    def profile(self, on: bool) -> None:
        """on: every step records a HIP event (torch's current stream) at its phase boundaries; `phase_ms()` gives the phases' durations"""
        self._marks = [] if on else None

    def _mark(self, name: str) -> None:
        if self._marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(self._dev))
            self._marks.append((name, e))

    def phase_ms(self) -> dict:
        """{phase: ms summed over the recorded steps}; call after a synchronize.  A phase lasts from the previous mark to its own."""
        out: Dict[str, float] = {}
        marks, self._marks = self._marks or [], []
        for (_, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            if n1 != "step_start":
                out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        return out

    def conv_macs(self) -> int:
        """multiply-adds of the FIRST stage's convolutions (main, projection and strided-concatenation convs) per clip, forward"""
        tot = 0
        for blk in self.blocks:
            for lay in blk.layers:
                convs = []
                if lay["kind"] == "conv":
                    convs = [lay["main"].conv] + [p.conv for _, p in lay["skips"] if p is not None]
                elif lay["kind"] == "cat":
                    convs = [cv for _, cv, _ in lay["parts"] if cv is not None]
                tot += sum(cv.Ho * cv.Wo * cv.K * cv.N for cv in convs)
        return tot

    # ---------------------------------------------------------------- forward / backward
    def _prep(self, x):
        if not (isinstance(x, torch.Tensor) and x.is_cuda):
            x = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self._dev)
        x = x.to(torch.float32).contiguous()
        Bm, T, Fq, Ch = self.input_shape
        # any batch 1 <= B <= the one the buffers were sized for, as SeldNet._prep: train.main builds the model for the largest of the train /
        # val / test batch sizes and every loader ends on a ragged batch (drop_remainder=False)
        if tuple(x.shape[1:]) != (T, Fq, Ch) or not 1 <= x.shape[0] <= Bm:
            raise ValueError(f"x shape {tuple(x.shape)}: a composed model built for {self.input_shape} runs batches of 1..{Bm} clips of {(T, Fq, Ch)}")
        return x

    def _forward(self, x, training: bool):
        rt = self.rt
        B = x.shape[0]
        h = x
        for blk in self.blocks:
            h = blk.forward(h, B, training)
        self._mark("first_fwd")
        R = B * self.S
        feat = h.reshape(R, -1)          # layers.force_1d_inputs (layers.py:41-47): feature = f * C + c
        self.feat0 = feat
        for i, G in enumerate(self.gru):
            G["x"] = feat
            for d, dn in enumerate(("fwd", "bwd")):
                b = rt.w(f"gru{i}.{dn}.bias")
                rt.gemm(feat, rt.w(f"gru{i}.{dn}.kernel"), b[:384], G["gx"][d], R, 384, G["in"])
            bf, bb = rt.w(f"gru{i}.fwd.bias"), rt.w(f"gru{i}.bwd.bias")
            rt.ck(rt.lib.seld_m_gru_fwd(rt.p(G["gx"][0]), rt.p(G["gx"][1]), rt.p(rt.w(f"gru{i}.fwd.recurrent_kernel")),
                                        rt.p(rt.w(f"gru{i}.bwd.recurrent_kernel")), rt.p(bf[384:]), rt.p(bb[384:]), rt.p(G["h"][0]), rt.p(G["h"][1]),
                                        rt.p(G["sv"][0]) if training else None, rt.p(G["sv"][1]) if training else None, rt.p(G["out"]), B, self.S, 128,
                                        rt.st()))
            feat = G["out"]
        sed = rt.empty(B, self.S, self.n_classes)
        doa = rt.empty(B, self.S, 3 * self.n_classes)
        for Hd, out in zip(self.heads, (sed, doa)):
            a = feat
            for lay in Hd["layers"]:
                lay["x"] = a
                rt.gemm(a, rt.w(lay["n"] + ".kernel"), rt.w(lay["n"] + ".bias"), lay["pre"], R, lay["out"], lay["in"])
                rt.act(lay["pre"][:R], lay["y"][:R], Hd["hact"])         # the buffers hold the LARGEST batch's rows: this batch's R only
                a = lay["y"]
            Hd["x"] = a
            rt.gemm(a, rt.w(Hd["name"] + ".out.kernel"), rt.w(Hd["name"] + ".out.bias"), Hd["pre"], R, Hd["out"], Hd["in"])
            rt.act(Hd["pre"][:R], out.view(R, -1), Hd["act"])
        self._mark("gru_heads_fwd")
        return sed, doa

    def __call__(self, x, training: bool = False):
        x = self._prep(x)
        sed, doa = self._forward(x, bool(training))
        return [sed, doa]

    def _backward(self, B):
        """from the heads' pre-activation gradients (Hd['dpre'], written by seld_k_losses) to every variable's gradient"""
        rt = self.rt
        R = B * self.S
        first = True
        for Hd in self.heads:
            n = Hd["name"]
            rt.gemm_tn(Hd["x"], Hd["dpre"], rt.g(n + ".out.kernel"), rt.g(n + ".out.bias"), R, Hd["in"], Hd["out"])
            if Hd["layers"]:
                rt.gemm(Hd["dpre"], rt.w(n + ".out.kernel"), None, Hd["layers"][-1]["dy"], R, Hd["in"], Hd["out"], transb=1)
            else:
                rt.gemm(Hd["dpre"], rt.w(n + ".out.kernel"), None, self.dfeat, R, Hd["in"], Hd["out"], transb=1, accumulate=0 if first else 1)
            for j in range(len(Hd["layers"]) - 1, -1, -1):
                lay = Hd["layers"][j]
                rt.act_bwd(lay["pre"][:R], lay["dy"][:R], lay["dpre"][:R], Hd["hact"])
                rt.gemm_tn(lay["x"], lay["dpre"], rt.g(lay["n"] + ".kernel"), rt.g(lay["n"] + ".bias"), R, lay["in"], lay["out"])
                if j > 0:
                    rt.gemm(lay["dpre"], rt.w(lay["n"] + ".kernel"), None, Hd["layers"][j - 1]["dy"], R, lay["in"], lay["out"], transb=1)
                else:
                    rt.gemm(lay["dpre"], rt.w(lay["n"] + ".kernel"), None, self.dfeat, R, lay["in"], lay["out"], transb=1, accumulate=0 if first else 1)
            first = False
        dout = self.dfeat
        for i in range(len(self.gru) - 1, -1, -1):
            G = self.gru[i]
            rt.ck(rt.lib.seld_m_gru_bwd(rt.p(dout), rt.p(G["h"][0]), rt.p(G["h"][1]), rt.p(G["sv"][0]), rt.p(G["sv"][1]),
                                        rt.p(rt.w(f"gru{i}.fwd.recurrent_kernel")), rt.p(rt.w(f"gru{i}.bwd.recurrent_kernel")), rt.p(G["dgx"][0]),
                                        rt.p(G["dgx"][1]), rt.p(G["dgh"][0]), rt.p(G["dgh"][1]), B, self.S, 128, rt.st()))
            for d, dn in enumerate(("fwd", "bwd")):
                gb = rt.g(f"gru{i}.{dn}.bias")
                rt.gemm_tn(G["x"], G["dgx"][d], rt.g(f"gru{i}.{dn}.kernel"), gb[:384], R, G["in"], 384)
                # recurrent kernel: h_prev^T dgh — the forward direction saw h[t-1], the backward direction h[t+1]
                rt.gemm_tn(G["h"][d], G["dgh"][d], rt.g(f"gru{i}.{dn}.recurrent_kernel"), gb[384:], R, 128, 384, seq=self.S, shift=-1 if d == 0 else 1)
                rt.gemm(G["dgx"][d], rt.w(f"gru{i}.{dn}.kernel"), None, G["din"], R, G["in"], 384, transb=1, accumulate=d)
            dout = G["din"]
        self._mark("heads_gru_bwd")
        dy = dout[:R].view(B, *self.blocks[-1].out_shape)
        for bi in range(len(self.blocks) - 1, -1, -1):
            dy = self.blocks[bi].backward(dy, None, B)
        self._mark("first_bwd")

    def _labels(self, y, B):
        ys = torch.as_tensor(y[0], dtype=torch.float32, device=self._dev).contiguous()
        yd = torch.as_tensor(y[1], dtype=torch.float32, device=self._dev).contiguous()
        if tuple(ys.shape) != (B, self.S, self.n_classes) or tuple(yd.shape) != (B, self.S, 3 * self.n_classes):
            raise ValueError(f"label shapes {tuple(ys.shape)}, {tuple(yd.shape)} do not match the model output")
        return ys, yd

    def _losses(self, sed, doa, ys, yd, cfg, want_grads):
        from . import losses
        rt = self.rt
        B = sed.shape[0]
        sloss = torch.empty((), dtype=torch.float32, device=self._dev)
        dloss = torch.empty((B, self.S) if cfg.doa_loss != 1 else (), dtype=torch.float32, device=self._dev)
        rt.ck(rt.lib.seld_m_losses(rt.p(sed), rt.p(doa), rt.p(ys), rt.p(yd), C.byref(cfg), rt.p(sloss), rt.p(dloss),
                                   rt.p(self.heads[0]["dpre"]) if want_grads else None, rt.p(self.heads[1]["dpre"]) if want_grads else None,
                                   rt.p(self.loss_scratch), B, self.S, self.n_classes, rt.st()))
        return sloss, dloss

    def train_step(self, x, y, cfg, optimizer, agc: bool = False):
        """train.trainstep (train.py:22-36) -> ([sed, doa], sloss, dloss)"""
        if agc:
            raise ValueError("adaptive gradient clipping is not wired for composed models")
        rt = self.rt
        x = self._prep(x)
        B = x.shape[0]
        ys, yd = self._labels(y, B)
        self._mark("step_start")
        sed, doa = self._forward(x, True)
        sloss, dloss = self._losses(sed, doa, ys, yd, cfg, True)
        self._mark("losses")
        self._backward(B)
        self.adam_step += 1
        rt.ck(rt.lib.seld_m_adam(rt.p(rt.params), rt.p(rt.grads), rt.p(rt.adam_m), rt.p(rt.adam_v), self.n_params, optimizer.learning_rate,
                                 optimizer.beta_1, optimizer.beta_2, optimizer.epsilon, self.adam_step, rt.st()))
        self._mark("adam")
        return [sed, doa], sloss, dloss

    def test_step(self, x, y, cfg):
        x = self._prep(x)
        ys, yd = self._labels(y, x.shape[0])
        sed, doa = self._forward(x, False)
        sloss, dloss = self._losses(sed, doa, ys, yd, cfg, False)
        return [sed, doa], sloss, dloss


def mother_block(model_config: dict):
    """reference modules.mother_block(model_config) -> a factory `(input_shape, batch) -> MotherBlock`; the configuration errors of
    modules.py:202-222 are raised here, as the reference raises them at construction."""
    check_mother_config(model_config)

    def build(input_shape, rt=None, prefix="mb0"):
        B = int(input_shape[0])
        rt = rt or _Rt(torch.device("cuda", torch.cuda.current_device()))
        return MotherBlock(rt, model_config, tuple(int(v) for v in input_shape[-3:]), prefix, B)
    return build
