"""Data-parallel host logic (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The reference has no distributed path; this is the MI355X-native addition.  The
SELDnet step shards by clips: every rank holds a full 2 MB weight replica and B/world clips; the
only exchange is the all-reduce(SUM) of the flat fp32 gradient buffer (513 840 floats = 2.06 MB,
latency-bound on xGMI) between backward and Adam — in the buckets the backward pass finishes them in, all but the small conv/BN
one overlapped with the rest of the backward —, plus one scalar all-reduce for the MMSE mask count.  BatchNorm statistics are per
replica by default; `enable_sync_batchnorm` makes them global (six 1-KB all-reduces per step).

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


def init_library_dp(model, group=None, force: bool = False) -> bool:
    """Hand data parallelism to the LIBRARY (include/seld_hip.h, seld_dp_*): one RCCL communicator per ctx, created here from a
    ncclUniqueId that rank 0 draws (seld_dp_unique_id) and torch.distributed carries to the other ranks — the only thing the host's
    process group is used for.  From then on `train.trainstep` calls seld_dp_allreduce_grads (two collectives on the library's own
    communication stream), the MMSE mask count is all-reduced on the device (no `.item()`), and `enable_sync_batchnorm` routes
    through the same communicator.  Returns False (and leaves the torch.distributed path in charge) when the group's backend is not
    RCCL — the two-ranks-on-one-GPU rehearsal over gloo — or the group has one rank and `force` is not set (tests force a one-rank
    communicator).  Failure is COLLECTIVE: if binding RCCL or ncclCommInitRank fails on any rank, every rank raises RuntimeError
    (and none has the library path enabled), so that a caller's fallback is taken by all ranks or none."""
    import ctypes as C
    from . import _lib
    d = torch.distributed
    world = world_size(group)
    if getattr(model, "_lib_dp", False):
        return True
    if world == 1 and not force:
        return False
    if world > 1 and d.get_backend(group) != "nccl":
        return False
    rank = d.get_rank(group) if world > 1 else 0

    def agree(ok: bool, what: str, err):
        """Every rank learns whether EVERY rank succeeded (MIN all-reduce of a flag) before anyone acts on the outcome: a rank that
        raised on its own while its peers sat in the next collective would leave the job in mixed library / torch.distributed code
        paths (a hang, or a gradient all-reduce matched against a broadcast)."""
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=model._dev)
            d.all_reduce(flag, op=d.ReduceOp.MIN, group=group)
            all_ok = bool(int(flag.item()))
        else:
            all_ok = ok
        if not all_ok:
            raise RuntimeError(f"library-owned RCCL communicator: {what} failed on "
                               + (f"this rank ({err})" if not ok else "another rank") + "; no rank enables the library path")

    # 1. every rank binds RCCL; ONLY rank 0 draws an id (ncclGetUniqueId starts a bootstrap listener thread + socket: an id drawn on the
    #    other ranks and thrown away would leave world - 1 idle listeners per job).  The availability check happens BEFORE the broadcast
    buf = (C.c_ubyte * 128)()
    err = None
    try:
        if rank == 0:
            _lib.check(model.lib.seld_dp_unique_id(buf), model.ctx)
        elif not model.lib.seld_dp_available():
            raise RuntimeError("RCCL (librccl.so.1) could not be loaded")
    except Exception as e:      # noqa: BLE001 - reported collectively below
        err = e
    agree(err is None, "seld_dp_unique_id (binding RCCL)", err)
    ident = torch.zeros(128, dtype=torch.uint8, device=model._dev)
    if rank == 0:
        ident.copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
    if world > 1:
        d.broadcast(ident, src=d.get_global_rank(group, 0) if group is not None else 0, group=group)
    host = ident.cpu().numpy().tobytes()
    torch.cuda.synchronize(model._dev)
    # 2. the communicator itself: again agreed on by all ranks; a rank that succeeded alone gives its communicator back
    err = None
    try:
        _lib.check(model.lib.seld_dp_init(model.ctx, rank, world, host), model.ctx)
    except Exception as e:      # noqa: BLE001
        err = e
    try:
        agree(err is None, "seld_dp_init", err)
    except RuntimeError:
        model.lib.seld_dp_destroy(model.ctx)      # a no-op without a communicator; a rank that succeeded alone gives its own back
        raise
    model._lib_dp = True
    # the heads' dropout draws are a function of (seed, step, layer, element): every rank gets its own key, or all replicas would drop
    # the same elements of their different clips
    model.set_option("dropout_seed", 0x5e1d + rank)
    return True


def allreduce_gradients(flat_grad: torch.Tensor, group=None, model=None, force: bool = False) -> None:
    """Sum the flat gradient buffer in place over ranks.

    With `model` (a SeldNet on a GPU) the buffer goes in the buckets the backward pass produces it in
    (seld_grads_bucket_ready): the last GRU layer + the heads first — final while the earlier layers' recurrences are
    still running —, then each earlier GRU layer, each on a communication stream that waits for exactly that bucket's
    event; the conv/BN bucket follows on the main stream when the step's kernels are enqueued.  All are complete (for the
    main stream) on return.  Without `model` (CPU tensors, gloo tests): one bucket."""
    if model is not None and getattr(model, "_lib_dp", False):      # the library's own communicator, stream and bucket order
        from . import _lib
        _lib.check(model.lib.seld_dp_allreduce_grads(model.ctx), model.ctx)
        return
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
    nb = int(model.lib.seld_grads_bucket_count(model.ctx))
    works = []
    off, cnt = C.c_int64(), C.c_int64()
    for k in range(nb - 1):
        _lib.check(model.lib.seld_grads_bucket_ready(model.ctx, k, C.c_void_p(comm.cuda_stream), C.byref(off), C.byref(cnt)), model.ctx)
        with torch.cuda.stream(comm):
            works.append(d.all_reduce(flat_grad[off.value:off.value + cnt.value], group=group, async_op=True))
    # conv / BN variables: produced by the main stream itself
    cur = torch.cuda.current_stream(flat_grad.device)
    _lib.check(model.lib.seld_grads_bucket_ready(model.ctx, nb - 1, C.c_void_p(cur.cuda_stream), C.byref(off), C.byref(cnt)), model.ctx)
    works.append(d.all_reduce(flat_grad[off.value:off.value + cnt.value], group=group, async_op=True))
    for wk in works:
        wk.wait()          # the current (main) stream waits for every bucket before Adam


def enable_sync_batchnorm(model, group=None, force: bool = False) -> None:
    """Synchronised BatchNorm over the ranks of `group` (seld_set_sync_bn): the per-channel sums of every conv block's
    BatchNormalization are all-reduced in the training forward and in the backward pass, so that B/world clips per rank
    reproduce the reference's single-device batch of B (layers.py:33).  Six 1-KB collectives per step.
    Ranks may hold different numbers of clips (a partial last batch): each rank's element count is all-reduced with its sums.
    A failing callback is fatal for the process group (the peers block in their collectives): exit the job."""
    import ctypes as C
    from . import _lib
    world = world_size(group)
    if getattr(model, "_lib_dp", False):            # the library's communicator carries the sums (seld_dp_set_sync_bn)
        _lib.check(model.lib.seld_dp_set_sync_bn(model.ctx, 1), model.ctx)
        return
    if world <= 1 and not force:
        model.lib.seld_set_sync_bn(model.ctx, None, None, 1)
        model._sync_bn_cb = None
        return
    dev = model._dev

    def _cb(_user, buf, count, dtype, _stream):
        try:
            t = torch.as_tensor(_F64Ptr(int(buf), int(count)), device=dev) if dtype == _lib.SELD_DTYPE_F64 else None
            if t is None:
                return 1
            # the ABI says "enqueue on `stream`": torch.distributed enqueues on torch's CURRENT stream, which SeldNet._prep makes the
            # ctx stream before every call — anything else would be an unordered collective on the library's buffer
            if int(_stream or 0) != int(torch.cuda.current_stream(dev).cuda_stream):
                return 1
            torch.distributed.all_reduce(t, group=group)
            return 0
        except Exception:                                       # never let an exception cross the C ABI
            import traceback
            traceback.print_exc()
            return 1

    model._sync_bn_cb = _lib.ALLREDUCE_FN(_cb)                  # keep the trampoline alive as long as the ctx may call it
    _lib.check(model.lib.seld_set_sync_bn(model.ctx, C.cast(model._sync_bn_cb, C.c_void_p), None, world), model.ctx)


class _F64Ptr:
    """A raw device pointer to `n` doubles owned by the HIP library, exposed to torch without a copy."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}

