import sys, os, copy, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from oracle import seldnet_oracle as O
from seld_amd import _lib, losses, models, train
from __graft_entry__ import SELDNET_CONFIG
cfg = copy.deepcopy(SELDNET_CONFIG); cfg["FIRST"] = "resnet50_block"
cfg["FIRST_ARGS"] = {"filters": 32, "block_num": [1, 1, 1, 1], "kernel_regularizer": {"l1": 0, "l2": 1e-3}}
B, T = 4, 300
spec = O.Spec.from_config(cfg)
w, st = O.random_weights(spec, 7)
x, ys, yd = O.synthetic_batch(B, T, seed=19)
free = {}
ref = O.train_step(spec, w, st, x, ys, yd, doa_loss="MSE", loss_weight=(1.0, 1000.0), lr=1e-3, step=1, dtype=torch.float64, record_routing=free)
def run(inject_pool, inject_which):
    model = models.seldnet((B, T, 64, 7), cfg)
    model.set_weights(w, st)
    def inject(vals, kind, block, k):
        v = np.ascontiguousarray(vals.reshape(-1).astype(np.uint8)); idx = np.arange(v.size, dtype=np.int64)
        if kind == 0: _lib.check(model.lib.seld_debug_set_routing(model.ctx, block, v.size, C.c_void_p(idx.ctypes.data), C.c_void_p(v.ctypes.data)), model.ctx)
        else: _lib.check(model.lib.seld_debug_set_relu_gates(model.ctx, block, k, v.size, C.c_void_p(idx.ctypes.data), C.c_void_p(v.ctypes.data)), model.ctx)
    if inject_pool:
        f = free[0]; inject(np.where(f["gate"].numpy(), f["pos"].numpy() + 1, 0), 0, 0, 0)
    for bi, (s_, b, ci, wd, stf, proj) in enumerate(O.resnet_plan(spec)):
        for k, nm in enumerate(("y0", "y1", "out")):
            if k in inject_which: inject(free[f"rn{s_}.{b}.{nm}"]["gate"].numpy(), 1, bi, k)
    train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), train.Adam(1e-3))
    g = model.get_grads().astype(np.float64)
    out = []
    for n, off, sh in model.variables:
        k = int(np.prod(sh)); r = ref["grad"][off:off + k]
        if np.abs(r).max() < 1e-9 * np.abs(ref["grad"]).max(): continue
        out.append((n, np.abs(g[off:off + k] - r).max() / np.abs(r).max()))
    # how many decisions differ
    return out
for pool, which in ((False, ()), (True, ()), (True, (0,)), (True, (0, 1)), (True, (0, 1, 2)), (False, (0, 1, 2))):
    o = run(pool, which)
    worst = sorted(o, key=lambda t: -t[1])[:4]
    print(f"pool={pool} gates={which}: worst", [(n, f"{e:.1e}") for n, e in worst], "median", f"{np.median([e for _, e in o]):.1e}")

# read back what the backward saw
model = models.seldnet((B, T, 64, 7), cfg)
model.set_weights(w, st)
S = T // 5
bi, k = 0, 0
(s_, b, ci, wd, stf, proj) = O.resnet_plan(spec)[bi]
gt = free[f"rn{s_}.{b}.y0"]["gate"].numpy()
v = np.ascontiguousarray((~gt).reshape(-1).astype(np.uint8)); idx = np.arange(v.size, dtype=np.int64)
_lib.check(model.lib.seld_debug_set_relu_gates(model.ctx, bi, k, v.size, C.c_void_p(idx.ctypes.data), C.c_void_p(v.ctypes.data)), model.ctx)
train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), train.Adam(1e-3))
buf = torch.empty(B * S * 16 * 128, device="cuda"); cnt = C.c_int64()
_lib.check(model.lib.seld_debug_relu_output(model.ctx, bi, k, C.c_void_p(buf.data_ptr()), buf.numel(), C.byref(cnt)), model.ctx)
got = (buf[:cnt.value] > 0).cpu().numpy().reshape(gt.shape)
print("after injecting the COMPLEMENT of y0's gates: stored gate == complement:", float((got == ~gt).mean()), " == original:", float((got == gt).mean()), "count", cnt.value, gt.size)

def grads_with(fn):
    m = models.seldnet((B, T, 64, 7), cfg); m.set_weights(w, st); fn(m)
    train.trainstep(m, x, (ys, yd), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), train.Adam(1e-3))
    return m.get_grads().astype(np.float64)
g0 = grads_with(lambda m: None)
def inj_gate(k):
    def f(m):
        (s_, b, ci, wd, stf, proj) = O.resnet_plan(spec)[1]
        gt = free[f"rn{s_}.{b}.{('y0','y1','out')[k]}"]["gate"].numpy()
        v = np.ascontiguousarray((~gt).reshape(-1).astype(np.uint8)); idx = np.arange(v.size, dtype=np.int64)
        _lib.check(m.lib.seld_debug_set_relu_gates(m.ctx, 1, k, v.size, C.c_void_p(idx.ctypes.data), C.c_void_p(v.ctypes.data)), m.ctx)
    return f
def inj_pool(m):
    f = free[0]; val = np.where(f["gate"].numpy(), (f["pos"].numpy() + 1) % 20 + 1, 1)
    v = np.ascontiguousarray(val.reshape(-1).astype(np.uint8)); idx = np.arange(v.size, dtype=np.int64)
    _lib.check(m.lib.seld_debug_set_routing(m.ctx, 0, v.size, C.c_void_p(idx.ctypes.data), C.c_void_p(v.ctypes.data)), m.ctx)
for name, fn in (("complement of block 1 y0", inj_gate(0)), ("complement of block 1 y1", inj_gate(1)), ("complement of block 1 out", inj_gate(2)), ("rotated pool0", inj_pool)):
    g1 = grads_with(fn)
    print(f"{name}: max |dg| / max |g| = {np.abs(g1 - g0).max() / np.abs(g0).max():.3e}")
