"""Data-parallel HIP path with more than one rank: two PROCESSES on the one test GPU, torch.distributed over gloo (RCCL refuses two
ranks on one device; gloo carries device tensors through the host), each driving the real library — seld_train_fwd_bwd, the
bucketed gradient all-reduce of seld_amd.parallel (seld_grads_bucket_ready events, communication stream), the MMSE denominator
all-reduce, the synchronised-BatchNorm callback, seld_adam_step — through train.trainstep exactly as bench.py and train.main do.
The updated weights of both ranks must be identical and equal the oracle's single-process step on the whole batch."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import check

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, mode, sync_bn, B, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from __graft_entry__ import SELDNET_CONFIG
        from seld_amd import losses, models, parallel, train
        from seld_amd.synthetic import synthetic_batch
        from oracle import seldnet_oracle as O          # weights only: the gradient engine is the HIP library
        spec = O.Spec.from_config(SELDNET_CONFIG)
        w, st = O.random_weights(spec, 0)
        x, ys, yd = synthetic_batch(B, T, seed=77)
        per = B // world
        sl = slice(rank * per, (rank + 1) * per)
        model = models.seldnet((per, T, 64, 7), SELDNET_CONFIG)
        model.set_weights(w, st)
        if sync_bn:
            parallel.enable_sync_batchnorm(model)
        doa = losses.MSE if mode == "MSE" else losses.MMSE
        y_p, sl_, dl_ = train.trainstep(model, x[sl], (ys[sl], yd[sl]), losses.BinaryCrossentropy(), doa, (1.0, 1000.0), train.Adam(1e-3))
        torch.cuda.synchronize()
        w1, st1 = model.get_weights()
        q.put((rank, w1, st1, model.get_grads(), y_p[0].cpu().numpy(), y_p[1].cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,sync_bn", [("MSE", True), ("MMSE", True), ("MSE", False)])
def test_two_process_dp_step_equals_single_process_batch(mode, sync_bn):
    from __graft_entry__ import SELDNET_CONFIG
    from oracle import seldnet_oracle as O
    world, B, T = 2, 4, 100
    port = 30100 + (os.getpid() % 400) + 400 * (["MSE", "MMSE"].index(mode) + 2 * int(sync_bn))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, sync_bn, B, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=600)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # both ranks hold the same all-reduced gradient and hence the same updated weights
    np.testing.assert_array_equal(got[0][2], got[1][2])
    np.testing.assert_array_equal(got[0][0], got[1][0])
    spec = O.Spec.from_config(SELDNET_CONFIG)
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(B, T, seed=77)
    sed = np.concatenate([got[r][3] for r in range(world)])
    doa = np.concatenate([got[r][4] for r in range(world)])
    if sync_bn:
        # synchronised BatchNorm: the two half-batches ARE the single-device batch of B (layers.py:33)
        ref = O.train_step(spec, w, st, x, ys, yd, doa_loss=mode, loss_weight=(1.0, 1000.0), lr=1e-3, step=1, dtype=torch.float64)
        check(f"dp2 {mode} sed", sed, ref["sed"])
        check(f"dp2 {mode} doa", doa, ref["doa"])
        tr, _ = O.variable_specs(spec)
        off = 0
        for name, shape in tr:
            n = int(np.prod(shape))
            if not (name.startswith("conv") and name.endswith("bias")):
                check(f"dp2 {mode} all-reduced grad {name}", got[0][2][off:off + n], ref["grad"][off:off + n])
            off += n
        check(f"dp2 {mode} BN moving stats", got[0][1], ref["new_state"])
        # Adam at step 1 moves every weight by lr * g / (|g| + eps): compare where the gradient is resolved
        big = np.abs(ref["grad"]) > 1e-3 * np.abs(ref["grad"]).max()
        assert np.abs(got[0][0] - ref["new_w"])[big].max() <= 2e-3 * 1e-3 + 1e-7
    else:
        # per-replica statistics (the documented default): the summed gradient is the gradient of the global objective with each
        # half normalised by its own statistics — the oracle evaluated that way
        tr, nt = O.variable_specs(spec)
        fw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        wd, sd = O.unflatten(fw, tr), O.unflatten(torch.tensor(st, dtype=torch.float64), nt)
        per = B // world
        outs = [O.forward(spec, wd, sd, torch.tensor(x[r * per:(r + 1) * per], dtype=torch.float64), True) for r in range(world)]
        s_, d_ = torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
        obj, _, _ = O.losses_and_objective(s_, d_, torch.tensor(ys, dtype=torch.float64), torch.tensor(yd, dtype=torch.float64), mode, (1.0, 1000.0))
        (g,) = torch.autograd.grad(obj, fw)
        off = 0
        for name, shape in tr:
            n = int(np.prod(shape))
            if not (name.startswith("conv") and name.endswith("bias")):
                check(f"dp2 per-replica BN grad {name}", got[0][2][off:off + n], g.numpy()[off:off + n])
            off += n


@pytest.mark.parametrize("launcher,model", [("torchrun", "seldnet"), ("torchrun", "xception_gru"), ("torchrun", "resnet50_gru"), ("self", "seldnet")])
def test_bench_two_ranks_rehearsal(launcher, model):
    """bench.py's N > 1 code path (rank environment, barrier + max-over-ranks timing, the gradient buckets, the exposed-communication
    leg, rank 0's JSON line) run as the driver launches it — `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` —
    but with both ranks on the one test GPU over gloo (SELD_BENCH_DEVICE / SELD_BENCH_BACKEND): a crash in this path would cost the
    round its scaling curve.  The block models too: their kernel gradients finish on the side stream, which the last gradient bucket
    has to wait for.  launcher = "self": the same run as plain `python bench.py --gpus 2` — bench.py starts its two ranks itself
    (spawn_ranks), which is what a driver command without torch.distributed.run gets."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, SELD_BENCH_DEVICE="0", SELD_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    port = 31500 + os.getpid() % 400 + 400 * ["seldnet", "xception_gru", "resnet50_gru"].index(model)
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "4", "--frames", "300", "--model", model]
    if launcher == "self":
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert lines[-1].startswith("{") and len(lines[-1]) < 3072, "the LAST stdout line is the compact record the driver parses"
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["dp_backend"] == "torch-gloo" and d["roofline"]["kernel"] and d["roofline"]["frac"] >= 0
    detail = json.loads([l for l in lines if l.startswith("BENCH_DETAIL ")][-1][len("BENCH_DETAIL "):])
    assert len(detail["comm"]["exposed_ms_per_step_by_rank"]) == 2 and detail["value"] == d["value"]


def _rccl_worker(rank, world, port, mode, B, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from __graft_entry__ import SELDNET_CONFIG
        from seld_amd import losses, models, parallel, train
        from seld_amd.synthetic import synthetic_batch
        from oracle import seldnet_oracle as O          # weights only
        spec = O.Spec.from_config(SELDNET_CONFIG)
        w, st = O.random_weights(spec, 0)
        x, ys, yd = synthetic_batch(B, T, seed=77)
        per = B // world
        sl = slice(rank * per, (rank + 1) * per)
        model = models.seldnet((per, T, 64, 7), SELDNET_CONFIG, device=rank)
        model.set_weights(w, st)
        assert parallel.init_library_dp(model) is True          # seld_dp_init(world = 2): the library's own communicator
        parallel.enable_sync_batchnorm(model)
        doa = losses.MSE if mode == "MSE" else losses.MMSE
        # evaluation is NOT a collective under library DP: rank 0 alone runs a test step (an MMSE one: its denominator is this rank's own)
        if rank == 0:
            _, _, dl_eval = train.teststep(model, x[sl], (ys[sl], yd[sl]), losses.BinaryCrossentropy(), doa)
            dl_eval = dl_eval.cpu().numpy().copy()
        else:
            dl_eval = None
        y_p, sl_, dl_ = train.trainstep(model, x[sl], (ys[sl], yd[sl]), losses.BinaryCrossentropy(), doa, (1.0, 1000.0), train.Adam(1e-3))
        torch.cuda.synchronize()
        w1, st1 = model.get_weights()
        q.put((rank, w1, st1, model.get_grads(), y_p[0].cpu().numpy(), y_p[1].cpu().numpy(), dl_eval))
        dist.barrier()
        model.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two devices: RCCL refuses two ranks on one (runs the day a multi-GPU box is leased)")
@pytest.mark.parametrize("mode", ["MSE", "MMSE"])
def test_two_devices_library_rccl_step_equals_single_process_batch(mode):
    """seld_dp_init(world = 2) / seld_dp_allreduce_grads / the on-device MMSE denominator all-reduce / SyncBN through the library's own RCCL
    communicator, one process per device: both ranks end with the same weights, equal to the oracle's step on the whole batch; a test
    step on one rank only neither hangs nor sees the other rank's mask count."""
    from __graft_entry__ import SELDNET_CONFIG
    from oracle import seldnet_oracle as O
    world, B, T = 2, 4, 100
    port = 32900 + (os.getpid() % 400) + 400 * ["MSE", "MMSE"].index(mode)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, mode, B, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=600)
        got[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(got[0][2], got[1][2])
    np.testing.assert_array_equal(got[0][0], got[1][0])
    spec = O.Spec.from_config(SELDNET_CONFIG)
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(B, T, seed=77)
    ref = O.train_step(spec, w, st, x, ys, yd, doa_loss=mode, loss_weight=(1.0, 1000.0), lr=1e-3, step=1, dtype=torch.float64)
    check(f"rccl dp2 {mode} sed", np.concatenate([got[r][3] for r in range(world)]), ref["sed"])
    tr, _ = O.variable_specs(spec)
    off = 0
    for name, shape in tr:
        n = int(np.prod(shape))
        if not (name.startswith("conv") and name.endswith("bias")):
            check(f"rccl dp2 {mode} all-reduced grad {name}", got[0][2][off:off + n], ref["grad"][off:off + n])
        off += n
    per = B // world
    ref_eval = O.test_step(spec, w, st, x[:per], ys[:per], yd[:per], mode, dtype=torch.float64)
    check(f"rccl dp2 {mode} rank-0-only test step dloss (local denominator)", got[0][5], ref_eval["dloss"])
