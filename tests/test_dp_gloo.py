"""Data-parallel host logic on CPU: 2 ranks over gloo (127.0.0.1).  The gradient engine here is the
ORACLE (the HIP kernels need a GPU); what is under test is seld_amd.parallel — the loss-scaling rule
and the single flat all-reduce — i.e. that the summed per-rank gradients equal the single-process
gradient of the global-batch objective (with per-replica BatchNorm, as DESIGN.md documents)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT  # noqa: F401


def _objective_grad(O, spec, w, st, x, ys, yd, mode, sed_scale, den, lw=(1.0, 1000.0)):
    """per-rank gradient with the library's cfg semantics (sed_grad_scale, mmse_den), via autograd"""
    tr, nt = O.variable_specs(spec)
    fw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    sed, doa, _ = O.forward(spec, O.unflatten(fw, tr), O.unflatten(torch.tensor(st, dtype=torch.float64), nt),
                            torch.tensor(x, dtype=torch.float64), True)
    ys, yd = torch.tensor(ys, dtype=torch.float64), torch.tensor(yd, dtype=torch.float64)
    sl = O.bce(ys, sed)
    if mode == "MSE":
        obj = (sl * lw[0] * sed_scale + O.keras_mse_fn(yd, doa) * lw[1]).sum()
    else:
        sh = yd.shape
        m = torch.round((yd.reshape(*sh[:-1], 3, -1) ** 2).sum(-2))
        m = torch.cat([m] * 3, -1)
        obj = sl * lw[0] * sed_scale + lw[1] * (((yd - doa) ** 2) * m).sum() / den
    (g,) = torch.autograd.grad(obj, fw)
    return g


def _worker(rank, world, port, mode, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import seldnet_oracle as O
    from seld_amd import parallel
    from __graft_entry__ import SELDNET_CONFIG
    spec = O.Spec.from_config(SELDNET_CONFIG)
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(2 * world, 50, seed=99)
    sl = slice(2 * rank, 2 * rank + 2)
    xs, yss, yds = x[sl], ys[sl], yd[sl]
    local_den = torch.tensor([float(3 * np.round((yds.reshape(2, 10, 3, 12) ** 2).sum(2)).sum())], dtype=torch.float64)
    sed_scale, den = parallel.loss_scaling(mode == "MMSE", local_den, None)
    g = _objective_grad(O, spec, w, st, xs, yss, yds, mode, sed_scale, den if den > 0 else float(local_den))
    parallel.allreduce_gradients(g)
    if rank == 0:
        q.put((g.numpy(), sed_scale, den))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["MSE", "MMSE"])
def test_two_rank_gradient_equals_global_objective(mode):
    from oracle import seldnet_oracle as O
    from __graft_entry__ import SELDNET_CONFIG
    world, port = 2, 29500 + (os.getpid() % 500) + (0 if mode == "MSE" else 500)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    g_dp, sed_scale, den = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference: the same global objective, BatchNorm statistics per 2-clip replica
    spec = O.Spec.from_config(SELDNET_CONFIG)
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(2 * world, 50, seed=99)
    tr, nt = O.variable_specs(spec)
    fw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    wd, sd = O.unflatten(fw, tr), O.unflatten(torch.tensor(st, dtype=torch.float64), nt)
    outs = [O.forward(spec, wd, sd, torch.tensor(x[2 * r:2 * r + 2], dtype=torch.float64), True) for r in range(world)]
    sed = torch.cat([o[0] for o in outs]); doa = torch.cat([o[1] for o in outs])
    obj, _, _ = O.losses_and_objective(sed, doa, torch.tensor(ys, dtype=torch.float64), torch.tensor(yd, dtype=torch.float64), mode, (1.0, 1000.0))
    (g_ref,) = torch.autograd.grad(obj, fw)
    err = np.abs(g_dp - g_ref.numpy()).max() / np.abs(g_ref.numpy()).max()
    print(f"[dp] {mode}: sed_scale={sed_scale} den={den} rel_err={err:.2e}")
    assert err < 1e-9
    if mode == "MMSE":
        assert sed_scale == 0.5 and den > 0


class _AllReduceSum(torch.autograd.Function):
    """differentiable all-reduce(SUM): the backward of a sum over ranks is the sum over ranks of the incoming gradients"""

    @staticmethod
    def forward(ctx, t):
        t = t.clone()
        dist.all_reduce(t)
        return t

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g)
        return g


def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import seldnet_oracle as O
    from seld_amd import parallel
    from __graft_entry__ import SELDNET_CONFIG
    spec = O.Spec.from_config(SELDNET_CONFIG)
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(2 * world, 50, seed=99)
    sl = slice(2 * rank, 2 * rank + 2)
    tr, nt = O.variable_specs(spec)
    fw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    sed, doa, new_st = O.forward(spec, O.unflatten(fw, tr), O.unflatten(torch.tensor(st, dtype=torch.float64), nt),
                                 torch.tensor(x[sl], dtype=torch.float64), True, bn_sync=(_AllReduceSum.apply, world))
    obj, _, _ = O.losses_and_objective(sed, doa, torch.tensor(ys[sl], dtype=torch.float64), torch.tensor(yd[sl], dtype=torch.float64),
                                       "MSE", (1.0, 1000.0))
    (g,) = torch.autograd.grad(obj, fw)
    parallel.allreduce_gradients(g)
    if rank == 0:
        q.put((g.numpy(), sed.detach().numpy(), torch.cat([new_st[n].reshape(-1) for n, _ in nt]).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sync_batchnorm_equals_single_process_batch():
    """The SyncBN rule the library implements (seld_set_sync_bn: global sums -> mean/var and c1/c2, per-rank dgamma/dbeta):
    2 ranks x 2 clips with synchronised statistics == ONE process on the 4-clip batch — gradient, outputs, moving statistics."""
    from oracle import seldnet_oracle as O
    from __graft_entry__ import SELDNET_CONFIG
    world, port = 2, 30600 + (os.getpid() % 500)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    g_dp, sed0, state = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    spec = O.Spec.from_config(SELDNET_CONFIG)
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(2 * world, 50, seed=99)
    ref = O.train_step(spec, w, st, x, ys, yd, doa_loss="MSE", loss_weight=(1.0, 1000.0), dtype=torch.float64)
    err = np.abs(g_dp - ref["grad"]).max() / np.abs(ref["grad"]).max()
    print(f"[dp] SyncBN: rel_err={err:.2e}")
    assert err < 1e-9
    assert np.abs(sed0 - ref["sed"][:2]).max() < 1e-12
    assert np.abs(state - ref["new_state"]).max() < 1e-9

