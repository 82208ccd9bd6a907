"""Per-variable gradient error of one train step vs the fp64 oracle, B clips of T frames (default 2 x 3000)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as g
from oracle import seldnet_oracle as O
from seld_amd import losses, models, train

B, T = int(sys.argv[1]), int(sys.argv[2])
opts = sys.argv[3:]
cfg = g.SELDNET_CONFIG
spec = O.Spec.from_config(cfg)
w, st = O.random_weights(spec, 0)
x, ys, yd = O.synthetic_batch(B, T, seed=1234)
model = models.seldnet((B, T, 64, 7), cfg)
for o in opts:
    k, v = o.split("=")
    model.set_option(k, int(v))
model.set_weights(w, st)
ref = O.train_step(spec, w, st, x, ys, yd, doa_loss="MSE", loss_weight=(1.0, 1000.0), lr=1e-3, step=1, dtype=torch.float64)
train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), train.Adam(1e-3), False)
got = model.get_grads()
for n, off, sh in model.variables:
    k = int(np.prod(sh))
    a, b = got[off:off + k], ref["grad"][off:off + k]
    if n == "conv0.kernel":
        d = np.abs(a - b).reshape(sh)
        i = np.unravel_index(d.argmax(), sh)
        print("conv0.kernel worst element", i, "got", a.reshape(sh)[i], "ref", b.reshape(sh)[i], " per-out-channel max err:", np.round(d.max(axis=(0, 1, 2))[:16], 4))
        print("  per-tap max err:", np.round(d.max(axis=(2, 3)), 4).tolist())
    print("%-28s rel err %.3e   |ref|max %.3e" % (n, np.abs(a - b).max() / max(np.abs(b).max(), 1e-30), np.abs(b).max()))

# argmax flips of the first pooling layer between the GPU's fp32 z and the fp64 oracle's z
import ctypes as C
import torch.nn.functional as F
from seld_amd import _lib
lib = _lib.load()
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
tr = {n: (o, s) for n, o, s in model.variables}
def var(n):
    o, s = tr[n]
    return w[o:o + int(np.prod(s))].reshape(s)
k0, b0, g0, be0 = var("conv0.kernel"), var("conv0.bias"), var("bn0.gamma"), var("bn0.beta")
xd = x.cuda() if isinstance(x, torch.Tensor) else torch.as_tensor(x).cuda()
zd = torch.empty((B, T, 64, 64), device="cuda")
ze = torch.empty((B, T // 5, 16, 64), device="cuda")
am = torch.empty((B, T // 5, 16, 64), device="cuda", dtype=torch.uint8)
kd, bd, gd = torch.as_tensor(k0).cuda(), torch.as_tensor(b0).cuda(), torch.as_tensor(g0).cuda()
assert lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(kd), ptr(bd), ptr(gd), ptr(zd), ptr(ze), ptr(am), None, B, T, 7) == 0
z32 = zd.cpu().numpy().astype(np.float64)
xt = torch.as_tensor(np.asarray(xd.cpu()), dtype=torch.float64).permute(0, 3, 1, 2)
z64 = F.conv2d(xt, torch.as_tensor(k0, dtype=torch.float64).permute(3, 2, 0, 1), torch.as_tensor(b0, dtype=torch.float64), padding=1).permute(0, 2, 3, 1).numpy()
mean, var_ = z64.mean(axis=(0, 1, 2)), z64.var(axis=(0, 1, 2))
sc = g0 / np.sqrt(var_ + 1e-3); sh = be0 - mean * sc
def amax(z):
    y = (z * sc + sh).reshape(B, T // 5, 5, 16, 4, 64).transpose(0, 1, 3, 5, 2, 4).reshape(B, T // 5, 16, 64, 20)
    return y.argmax(-1), y.max(-1)
a32, m32 = amax(z32)
a64, m64 = amax(z64)
flips = (a32 != a64) & (m64 > 0)
print("max |z32 - z64| = %.3e;  pooling windows whose argmax differs between fp32 and fp64 z: %d of %d" % (np.abs(z32 - z64).max(), flips.sum(), flips.size))
