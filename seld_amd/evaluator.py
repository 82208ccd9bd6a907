"""Host-side mirror of the reference's sliding-window inference (evaluator.py:16-50, identical code at
trainv2.py:158-192): frame a clip into overlapping windows, run the model on them in batches, and
overlap-add-average the per-window outputs back to one sequence."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .models import SeldNet


def ensemble_outputs(model: SeldNet, xs: list, win_size: int = 300, step_size: int = 5, batch_size: int = 256):
    """reference ensemble_outputs -> [(sed [n_label_frames, nc], doa [n_label_frames, 3nc]), ...] (device tensors).

    `model` must have been built for windows of `win_size` frames and a batch of at least `batch_size`
    (the reference's Keras model bakes the window length in the same way, models.py:22)."""
    Bm, T_model, F, Cc = model.input_shape
    if T_model != win_size:
        raise ValueError(f"model was built for {T_model}-frame windows, not {win_size}")
    batch_size = min(batch_size, Bm)
    lib, dev = model.lib, model._dev
    fused = isinstance(model, SeldNet)      # a fused seld_ctx writes its outputs in place; a modules.ComposedSeldNet (FIRST = mother_block /
                                            # mother_stage) has no ctx: its forward is called per batch and the outputs copied into place
    st = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    L = model.S
    if L * step_size != win_size:
        raise ValueError("win_size // step_size must equal the model's label frames per window")
    out = []
    # work buffers live with the model (one set per (windows, batch) geometry): a file costs its kernels, not three allocations and
    # two device-to-device copies per batch
    cache = model.__dict__.setdefault("_infer_bufs", {})
    if fused:
        _lib.check(lib.seld_set_stream(model.ctx, st()), model.ctx)
    elif not callable(model):
        raise ValueError(f"ensemble_outputs: {type(model).__name__} is neither a SeldNet nor a composed model")
    for x in xs:
        x = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x).to(dev, torch.float32).contiguous()
        T = int(x.shape[0])
        n_win = 1 + (T - win_size) // step_size          # tf.signal.frame(pad_end=False)
        if n_win < 1 or tuple(x.shape[1:]) != (F, Cc):
            raise ValueError(f"clip shape {tuple(x.shape)} does not fit {win_size}-frame windows of [{F},{Cc}]")
        key = (n_win, batch_size)
        if key not in cache:
            cache.clear()
            cache[key] = (torch.empty((n_win, L, model.n_classes), dtype=torch.float32, device=dev),
                          torch.empty((n_win, L, 3 * model.n_classes), dtype=torch.float32, device=dev),
                          torch.empty((batch_size, win_size, F, Cc), dtype=torch.float32, device=dev))
        seds, doas, win = cache[key]
        for i in range(math.ceil(n_win / batch_size)):
            w0 = i * batch_size
            n = min(batch_size, n_win - w0)
            _lib.check(lib.seld_frame_windows(x.data_ptr(), win.data_ptr(), T, F * Cc, win_size, step_size, w0, n, st()))
            if fused:
                _lib.check(lib.seld_set_batch(model.ctx, n), model.ctx)
                # the forward writes its outputs straight into rows [w0, w0 + n) of the per-window tensors
                _lib.check(lib.seld_forward(model.ctx, win.data_ptr(), seds[w0:].data_ptr(), doas[w0:].data_ptr(), 0), model.ctx)
            else:
                s_, d_ = model(win[:n], training=False)
                seds[w0:w0 + n].copy_(s_)
                doas[w0:w0 + n].copy_(d_)
        T_out = n_win - 1 + L
        sed = torch.empty((T_out, model.n_classes), dtype=torch.float32, device=dev)
        doa = torch.empty((T_out, 3 * model.n_classes), dtype=torch.float32, device=dev)
        _lib.check(lib.seld_overlap_average(seds.data_ptr(), sed.data_ptr(), n_win, L, model.n_classes, st()))
        _lib.check(lib.seld_overlap_average(doas.data_ptr(), doa.data_ptr(), n_win, L, 3 * model.n_classes, st()))
        out.append((sed, doa))
    if fused:
        _lib.check(lib.seld_set_batch(model.ctx, Bm), model.ctx)      # leave the context at the batch it was built for, as it was found
    return out
