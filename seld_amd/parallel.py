"""Data-parallel host logic (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The reference has no distributed path; this is the MI355X-native addition.  The
SELDnet step shards by clips: every rank holds a full 2 MB weight replica and B/world clips; the
only exchange is the all-reduce(SUM) of the flat fp32 gradient buffer (513 840 floats = 2.06 MB,
latency-bound on xGMI) between backward and Adam — in two buckets, the large one overlapped with the conv backward —, plus one scalar all-reduce for the MMSE mask
count.  BatchNorm statistics stay per replica.

Loss-reduction rules that make the summed gradient equal the single-device gradient of the
global batch (derivation in DESIGN.md §5):
  MSE  (Keras function form; tape.gradient sums the [B,S] loss tensor):
       objective = B_g*S*w0*bce_global + w1*sum_rows mse  -> local objectives simply add: scale 1.
  MMSE (scalar): objective = w0*bce_global + w1*num_global/den_global
       -> BCE gradient scaled by 1/world, MMSE divided by the all-reduced den.
"""
from __future__ import annotations

import torch


def world_size(group=None) -> int:
    d = torch.distributed
    return d.get_world_size(group) if (d.is_available() and d.is_initialized()) else 1


def loss_scaling(is_mmse: bool, local_den: torch.Tensor | None, group=None):
    """-> (sed_grad_scale, mmse_den).  local_den: 1-element tensor holding this rank's sum(mask)."""
    w = world_size(group)
    if w == 1 or not is_mmse:
        return 1.0, 0.0
    den = local_den.clone()
    torch.distributed.all_reduce(den, group=group)
    return 1.0 / w, float(den.item())


def allreduce_gradients(flat_grad: torch.Tensor, group=None, model=None, force: bool = False) -> None:
    """Sum the flat gradient buffer in place over ranks.

    With `model` (a SeldNet on a GPU) the buffer goes in two buckets: the GRU + head gradients (86 % of the bytes)
    are final early in the backward pass — the library produces them on its side stream — so their all-reduce is
    issued on a communication stream that waits for exactly that (seld_grads_tail_ready) and runs UNDER the conv
    backward still executing on the main stream; the conv/BN bucket follows on the main stream.  Both are complete
    (for the main stream) on return.  Without `model` (CPU tensors, gloo tests): one bucket."""
    if world_size(group) <= 1 and not force:        # force: exercise the collective path on a one-rank group (tests)
        return
    d = torch.distributed
    if model is None or not flat_grad.is_cuda:
        d.all_reduce(flat_grad, group=group)
        return
    import ctypes as C
    from . import _lib
    comm = getattr(model, "_comm_stream", None)
    if comm is None:
        comm = model._comm_stream = torch.cuda.Stream(device=flat_grad.device)
    off = C.c_int64()
    _lib.check(model.lib.seld_grads_tail_ready(model.ctx, C.c_void_p(comm.cuda_stream), C.byref(off)), model.ctx)
    with torch.cuda.stream(comm):
        tail = d.all_reduce(flat_grad[off.value:], group=group, async_op=True)
    head = d.all_reduce(flat_grad[:off.value], group=group, async_op=True)
    head.wait()
    tail.wait()          # the current (main) stream waits for both before Adam
