"""Data-parallel host logic (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The reference has no distributed path; this is the MI355X-native addition.  The
SELDnet step shards by clips: every rank holds a full 2 MB weight replica and B/world clips; the
only exchange is ONE all-reduce(SUM) of the flat fp32 gradient buffer (513 840 floats = 2.06 MB,
latency-bound on xGMI) between backward and Adam, plus one scalar all-reduce for the MMSE mask
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


def allreduce_gradients(flat_grad: torch.Tensor, group=None) -> None:
    """One bucket: the whole flat gradient buffer, summed in place over ranks."""
    if world_size(group) > 1:
        torch.distributed.all_reduce(flat_grad, group=group)
