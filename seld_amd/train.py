"""Host-side mirror of the reference's train.py step functions (train.py:22-44) plus the
data-parallel wrapper the reference lacks."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib, losses, parallel
from .models import SeldNet


class Adam:
    """tf.keras.optimizers.Adam(learning_rate) (train.py:311): the slots live in the HIP ctx."""

    def __init__(self, learning_rate: float = 1e-3, beta_1: float = 0.9, beta_2: float = 0.999, epsilon: float = 1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon


def _cfg(doa_loss, loss_weight: Sequence[float], sed_grad_scale: float = 1.0, mmse_den: float = 0.0) -> _lib.LossCfg:
    if not isinstance(doa_loss, (losses._MSE, losses._MMSE)):
        raise ValueError("doa_loss must be seld_amd.losses.MSE or .MMSE")
    return _lib.LossCfg(doa_loss.code, float(loss_weight[0]), float(loss_weight[1]), float(sed_grad_scale), float(mmse_den))


def _labels(model: SeldNet, y, B: int):
    ys = torch.as_tensor(y[0], dtype=torch.float32, device=model._dev).contiguous()
    yd = torch.as_tensor(y[1], dtype=torch.float32, device=model._dev).contiguous()
    if tuple(ys.shape) != (B, model.S, model.n_classes) or tuple(yd.shape) != (B, model.S, 3 * model.n_classes):
        raise ValueError(f"label shapes {tuple(ys.shape)}, {tuple(yd.shape)} do not match the model output")
    return ys, yd


def _loss_outputs(model: SeldNet, doa_loss, B: int):
    sloss = torch.empty((), dtype=torch.float32, device=model._dev)
    dshape = (B, model.S) if isinstance(doa_loss, losses._MSE) else ()
    dloss = torch.empty(dshape, dtype=torch.float32, device=model._dev)
    return sloss, dloss


def trainstep(model: SeldNet, x, y, sed_loss, doa_loss, loss_weight, optimizer: Adam, agc: bool = False,
              process_group=None):
    """reference train.trainstep (train.py:22-36) -> (y_p, sloss, dloss).

    With torch.distributed initialised (one process per GPU) the flat gradient buffer is summed over
    ranks with one RCCL all-reduce between backward and Adam; BatchNorm statistics stay per replica."""
    if not isinstance(sed_loss, losses.BinaryCrossentropy):
        raise ValueError("sed_loss must be seld_amd.losses.BinaryCrossentropy()")
    x = model._prep(x)
    B = x.shape[0]
    ys, yd = _labels(model, y, B)
    sed, doa = model._outputs(B)
    sloss, dloss = _loss_outputs(model, doa_loss, B)
    world = parallel.world_size(process_group)
    is_mmse = isinstance(doa_loss, losses._MMSE)
    dent = None
    if world > 1 and is_mmse:
        # scalar objective: BCE is a mean over the GLOBAL batch, MMSE divides by the GLOBAL sum(mask)
        dent = torch.empty(1, dtype=torch.float32, device=model._dev)
        _lib.check(model.lib.seld_mmse_den(model.ctx, yd.data_ptr(), dent.data_ptr()), model.ctx)
    sed_scale, den = parallel.loss_scaling(is_mmse, dent, process_group)
    cfg = _cfg(doa_loss, loss_weight, sed_scale, den)
    _lib.check(model.lib.seld_train_fwd_bwd(model.ctx, x.data_ptr(), ys.data_ptr(), yd.data_ptr(), C.byref(cfg),
                                            sed.data_ptr(), doa.data_ptr(), sloss.data_ptr(), dloss.data_ptr()), model.ctx)
    if world > 1:
        parallel.allreduce_gradients(model.grad_tensor(), process_group)
    _lib.check(model.lib.seld_adam_step(model.ctx, optimizer.learning_rate, optimizer.beta_1, optimizer.beta_2,
                                        optimizer.epsilon, int(bool(agc))), model.ctx)
    return [sed, doa], sloss, dloss


def teststep(model: SeldNet, x, y, sed_loss, doa_loss):
    """reference train.teststep (train.py:39-44) -> (y_p, sloss, dloss)."""
    x = model._prep(x)
    B = x.shape[0]
    ys, yd = _labels(model, y, B)
    sed, doa = model._outputs(B)
    sloss, dloss = _loss_outputs(model, doa_loss, B)
    cfg = _cfg(doa_loss, (1.0, 1.0))
    _lib.check(model.lib.seld_test_step(model.ctx, x.data_ptr(), ys.data_ptr(), yd.data_ptr(), C.byref(cfg),
                                        sed.data_ptr(), doa.data_ptr(), sloss.data_ptr(), dloss.data_ptr()), model.ctx)
    return [sed, doa], sloss, dloss
