"""Where does the conv1 weight-gradient error come from at full clip length?  B=2, H=3000:
fused backward with (A) fp64 batch statistics, (B) the statistics the forward kernel produces."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib

lib = _lib.load()
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).cuda()
B, H, CIN = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 3000, 7
rng = np.random.default_rng(12)
x = rng.standard_normal((B, H, 64, CIN)).astype(np.float32)
w = (rng.standard_normal((3, 3, CIN, 64)) / np.sqrt(9 * CIN)).astype(np.float32)
b = rng.standard_normal(64).astype(np.float32) * 0.1
gamma = rng.uniform(0.5, 1.5, 64).astype(np.float32)
beta = rng.normal(0, 0.3, 64).astype(np.float32)
xd, wd, bd, gd, bed = dev(x), dev(w), dev(b), dev(gamma), dev(beta)
zd = torch.empty((B, H, 64, 64), device="cuda")
ze = torch.empty((B, H // 5, 16, 64), device="cuda")
am = torch.empty((B, H // 5, 16, 64), device="cuda", dtype=torch.uint8)
st = torch.zeros(128, device="cuda")
assert lib.seld_k_conv_first_fwd_pool(ptr(xd), ptr(wd), ptr(bd), ptr(gd), ptr(zd), ptr(ze), ptr(am), ptr(st), B, H, CIN) == 0
N = B * H * 64
s = st.cpu().numpy().astype(np.float64)
mean_gpu = s[:64] / N
var_gpu = s[64:] / N - mean_gpu ** 2
tw = torch.as_tensor(w, dtype=torch.float64).requires_grad_(True)
tb = torch.as_tensor(b, dtype=torch.float64).requires_grad_(True)
tg = torch.as_tensor(gamma, dtype=torch.float64).requires_grad_(True)
tbe = torch.as_tensor(beta, dtype=torch.float64).requires_grad_(True)
zt = F.conv2d(torch.as_tensor(x, dtype=torch.float64).permute(0, 3, 1, 2), tw.permute(3, 2, 0, 1), tb, padding=1).permute(0, 2, 3, 1)
mean = zt.mean(dim=(0, 1, 2))
var = ((zt - mean) ** 2).mean(dim=(0, 1, 2))
invstd = torch.rsqrt(var + 1e-3)
print("mean err %.3e  var rel err %.3e" % (np.abs(mean_gpu - mean.detach().numpy()).max(), np.abs(var_gpu / var.detach().numpy() - 1).max()))
zh = zd.cpu().numpy().astype(np.float64)
print("z err vs fp64 %.3e" % np.abs(zh - zt.detach().numpy()).max())
y = (zt - mean) * invstd * tg + tbe
p = F.max_pool2d(torch.relu(y).permute(0, 3, 1, 2), (5, 4), (5, 4)).permute(0, 2, 3, 1)
dp = rng.standard_normal(tuple(p.shape)).astype(np.float32)
gw, gb, gg, gbe = torch.autograd.grad(p, (tw, tb, tg, tbe), torch.as_tensor(dp, dtype=torch.float64))
dpd = dev(dp)
for tag, m, iv in (("A fp64 stats", mean.detach().numpy(), invstd.detach().numpy()), ("B gpu stats ", mean_gpu, 1.0 / np.sqrt(var_gpu + 1e-3))):
    nan = lambda *sh: torch.full(sh, float("nan"), device="cuda")
    dw, db, dg, dbe = nan(3, 3, CIN, 64), nan(64), nan(64), nan(64)
    md, ivd = dev(m), dev(iv)
    assert lib.seld_k_conv1_bwd_fused(ptr(xd), ptr(zd), ptr(dpd), ptr(md), ptr(ivd), ptr(gd), ptr(bed), ptr(dw), ptr(db),
                                      ptr(dg), ptr(dbe), B, H, CIN, 5, 4) == 0
    e = lambda a, r: np.abs(a.cpu().numpy() - r.numpy()).max() / np.abs(r.numpy()).max()
    print(tag, "dw rel err %.3e  dgamma %.3e  dbeta %.3e   |db|max %.3e  (|dw|max %.3e)" % (e(dw, gw), e(dg, gg), e(dbe, gbe), db.abs().max().item(), gw.abs().max().item()))
# how often do two pixels of a pooling window share the maximal fp32 y = fmaf(z, scale, shift)?
sc = (gamma.astype(np.float64) * invstd.detach().numpy()).astype(np.float32)
sh = (beta.astype(np.float64) - mean.detach().numpy() * sc).astype(np.float32)
y = (zd.cpu().numpy().astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)).astype(np.float32)
win = y.reshape(B, H // 5, 5, 16, 4, 64)
mx = win.max(axis=(2, 4), keepdims=True)
nties = ((win == mx).sum(axis=(2, 4)) > 1) & (mx[:, :, 0, :, 0, :] > 0)
print("pooling windows with a tied positive maximum: %d of %d" % (nties.sum(), nties.size))
