"""Time the first-layer conv(+pool) kernels through their seld_k_* entry points at the headline shape
(B=32, T=3000); with a -DCPOOL_TIMING build of the library, the per-phase cycle counts of conv_pool.hip.  Wall clock around a synchronous call: good to ~10 us."""
import ctypes as C
import sys
import time

import os

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from seld_amd import _lib

lib = _lib.load()
B, H, Cin = 32, 3000, 7
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, H, 64, Cin, device="cuda", generator=g)
w = torch.randn(3, 3, Cin, 64, device="cuda", generator=g) / (9 * Cin) ** 0.5
b = torch.randn(64, device="cuda", generator=g)
gamma = torch.ones(64, device="cuda")
z = torch.empty(B, H, 64, 64, device="cuda")
ze = torch.empty(B, H // 5, 16, 64, device="cuda")
am = torch.empty(B, H // 5, 16, 64, device="cuda", dtype=torch.uint8)
st = torch.zeros(128, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, n=30):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# clock ramp: keep the GPU busy for a while before the first measurement
a_ = torch.randn(4096, 4096, device="cuda")
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    (a_ @ a_).sum().item()


unfused = lambda: lib.seld_k_conv3x3_fwd(P(x), P(w), P(b), P(z), P(st), B, H, 64, Cin, 64)
print("conv_first_fwd (unfused)      %.4f ms" % timeit(unfused))
print("conv_first_fwd (unfused)      %.4f ms" % timeit(unfused))
for _ in range(2):
    t1 = timeit(lambda: lib.seld_k_conv_first_fwd_pool(P(x), P(w), P(b), P(gamma), P(z), P(ze), P(am), P(st), B, H, Cin))
    t2 = timeit(lambda: lib.seld_k_conv_first_fwd_pool(P(x), P(w), P(b), P(gamma), None, P(ze), None, P(st), B, H, Cin))
    print("conv_first_fwd_pool          %.4f ms (z stored)   %.4f ms (z not stored)" % (t1, t2))
print("conv_first_fwd (unfused)      %.4f ms" % timeit(unfused))
if os.environ.get("CPOOL_TIMING"):      # library built with -DCPOOL_TIMING: stats slots 0..5 = summed phase cycles of wave 0
    for zz in (z, None):
        lib.seld_k_conv_first_fwd_pool(P(x), P(w), P(b), P(gamma), P(zz) if zz is not None else None, P(ze), P(am) if zz is not None else None, P(st), B, H, Cin)
        v = st[:6].cpu().numpy() / 512 / 18.75
        print("z stored    " if zz is not None else "z not stored", "cycles per tile (wave 0 mean): top %.0f  phaseA %.0f  phaseB+drainA %.0f  commit %.0f  drainB %.0f  barrier %.0f  total %.0f" % (*v, v.sum()))
if os.environ.get("CPSB_TIMING"):       # library built with -DCPSB_TIMING (conv_pool_sb.hip): phase cycles of wave 0, z not stored
    lib.seld_k_conv_first_fwd_pool(P(x), P(w), P(b), P(gamma), None, P(ze), P(am), P(st), B, H, Cin)
    v = st[:4].cpu().numpy() / 512 / 18.75
    print("split-bf16, z not stored: cycles per tile (wave 0 mean): top %.0f  mfma loop %.0f  window reduction %.0f  barrier+commit+barrier %.0f  total %.0f" % (*v, v.sum()))
