"""Why does conv0.kernel's gradient differ from the fp64 oracle by more than 1e-4 at full size?  Hypothesis: not summation
error but ROUTING FLIPS — pooling windows whose two largest elements (or whose maximum and 0) lie within one fp32 rounding of
each other send a whole gradient element to a different pixel (or gate it differently) than the fp64 evaluation does.

Test: evaluate the fp64 oracle a second time with the FIRST block's MaxPool/ReLU routing replaced by the routing the GPU
recorded (argmax position per window + sign of the pooled value).  If the hypothesis holds, the HIP gradients agree with that
"GPU-routed fp64" evaluation to ~1e-5 while they differ from the free-running fp64 oracle by ~1e-3.

  python tools/diag_routing_flips.py [B=32] [T=3000]      (GPU box; ~35 GB of host memory at B=32)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g
from oracle import seldnet_oracle as O
from seld_amd import _lib, losses, models, train

import threading
import time


def _heartbeat():      # the fp64 oracle runs for minutes without output: gpurun takes 7 silent minutes for a hang
    t0 = time.time()
    while True:
        time.sleep(60)
        print(f"[heartbeat] {time.time() - t0:.0f} s", flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
cfg = g.SELDNET_CONFIG
spec = O.Spec.from_config(cfg)
tr, nt = O.variable_specs(spec)
w, st = O.random_weights(spec, 0)
x, ys, yd = O.synthetic_batch(B, T, seed=1234)
model = models.seldnet((B, T, 64, 7), cfg)
model.set_weights(w, st)
train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), train.Adam(1e-3), False)
g_gpu = model.get_grads().astype(np.float64)

# the routing the GPU's first block recorded (same kernel, same grid -> bit-identical to the model's own run)
lib = _lib.load()
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
wd32 = O.unflatten(torch.as_tensor(w), tr)
xd = torch.as_tensor(x).cuda()
k0, b0, g0 = (wd32[n].contiguous().cuda() for n in ("conv0.kernel", "conv0.bias", "bn0.gamma"))
Hp = T // 5
ze = torch.empty((B, Hp, 16, 64), device="cuda")
am = torch.empty((B, Hp, 16, 64), device="cuda", dtype=torch.uint8)
stt = torch.zeros(128, device="cuda")
assert lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(k0), ptr(b0), ptr(g0), None, ptr(ze), ptr(am), ptr(stt), B, T, 7) == 0
N = B * T * 64
s = stt.cpu().numpy().astype(np.float64)
mean = s[:64] / N
var = s[64:] / N - mean ** 2
sc = wd32["bn0.gamma"].numpy().astype(np.float64) / np.sqrt(var + O.BN_EPS)
sh = wd32["bn0.beta"].numpy().astype(np.float64) - mean * sc
gate_gpu = torch.as_tensor((ze.cpu().numpy().astype(np.float64) * sc + sh) > 0)
idx_gpu = am.cpu().to(torch.int64)
del xd, ze, am
torch.cuda.empty_cache()


def forward_routed(wd, sd, xt, idx, gate, record):
    """oracle.forward with block 0's pool/ReLU routing fixed to (idx, gate) when given; records the free routing otherwise."""
    h = xt
    for i in range(3):
        z = O.conv2d_same_nhwc(h, wd[f"conv{i}.kernel"], wd[f"conv{i}.bias"])
        y, _, _ = O.batchnorm(z, wd[f"bn{i}.gamma"], wd[f"bn{i}.beta"], sd[f"bn{i}.moving_mean"], sd[f"bn{i}.moving_variance"], True)
        if i == 0:
            Bz, H, Wd, Cc = y.shape
            yw = y.reshape(Bz, H // 5, 5, Wd // 4, 4, Cc).permute(0, 1, 3, 5, 2, 4).reshape(Bz, H // 5, Wd // 4, Cc, 20)
            if idx is None:
                top, a = yw.max(dim=-1)
                record["idx"], record["gate"] = a.detach(), (top > 0).detach()
                h = torch.relu(top)
            else:
                h = torch.where(gate, yw.gather(-1, idx.unsqueeze(-1)).squeeze(-1), torch.zeros((), dtype=y.dtype))
        else:
            h = O.maxpool_nhwc(torch.relu(y), spec.pools[i])
    Bz, S = h.shape[0], h.shape[1]
    h = h.reshape(Bz, S, -1)
    for i in range(2):
        h = O.bigru_mul(h, wd, f"gru{i}")
    outs = []
    for head, act in (("sed", torch.sigmoid), ("doa", torch.tanh)):
        a = h @ wd[f"{head}.dense0.kernel"][0] + wd[f"{head}.dense0.bias"]
        outs.append(act(a @ wd[f"{head}.out.kernel"] + wd[f"{head}.out.bias"]))
    return outs


def grads(idx, gate, record):
    fw = torch.as_tensor(w, dtype=torch.float64).clone().requires_grad_(True)
    wd, sd = O.unflatten(fw, tr), O.unflatten(torch.as_tensor(st, dtype=torch.float64), nt)
    sed, doa = forward_routed(wd, sd, torch.as_tensor(x, dtype=torch.float64), idx, gate, record)
    obj, _, _ = O.losses_and_objective(sed, doa, torch.as_tensor(ys, dtype=torch.float64), torch.as_tensor(yd, dtype=torch.float64), "MSE", (1.0, 1000.0))
    (gr,) = torch.autograd.grad(obj, fw)
    return gr.numpy()


rec = {}
g_free = grads(None, None, rec)
flip_arg = ((rec["idx"] != idx_gpu) & rec["gate"] & gate_gpu).sum().item()
flip_gate = (rec["gate"] != gate_gpu).sum().item()
print(f"B={B} T={T}: first-block pooling windows {idx_gpu.numel()}; argmax flips (both gates open) {flip_arg}; ReLU gate flips {flip_gate}")
g_rout = grads(idx_gpu, gate_gpu, rec)
print("%-28s %-22s %-22s %-22s" % ("variable", "HIP vs fp64 oracle", "HIP vs GPU-routed fp64", "routed fp64 vs free fp64"))
for n, off, shp in model.variables:
    k = int(np.prod(shp))
    if n.startswith("conv") and n.endswith("bias"):
        continue
    a, f, r = g_gpu[off:off + k], g_free[off:off + k], g_rout[off:off + k]
    den = np.abs(f).max()
    print("%-28s %-22.3e %-22.3e %-22.3e" % (n, np.abs(a - f).max() / den, np.abs(a - r).max() / den, np.abs(r - f).max() / den))
