"""mother_stage's products through the module operators (seld_m_gemm / seld_m_gemm_tn): rows of K = 63 / 927 / 7 / 103 floats (not 16-byte
aligned: scalar loads in gemm_f32 / gemm_tn) against the next multiple of 4."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def t_us(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, K, N, tag in [(422400, 63, 96, "b0.c1 3x3 7->96"), (422400, 64, 96, "  K padded"), (30720, 927, 96, "b1.c1 3x3 103->96"), (30720, 928, 96, "  K padded"),
                     (422400, 7, 96, "b0.p1_0 1x1 7->96"), (422400, 8, 96, "  K padded"), (30720, 103, 96, "b1.p1_0 1x1"), (30720, 104, 96, "  K padded")]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(K, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    z = torch.empty(M, N, device="cuda")
    dz = torch.randn(M, N, device="cuda")
    dcol = torch.empty(M, K, device="cuda")
    dw = torch.empty(K, N, device="cuda")
    db = torch.empty(N, device="cuda")
    slab = torch.empty(int(lib.seld_m_gemm_tn_scratch(K, N)), device="cuda")
    f = t_us(lambda: lib.seld_m_gemm(P(a), P(w), P(bias), P(z), M, N, K, 0, 0, None))
    d = t_us(lambda: lib.seld_m_gemm(P(dz), P(w), None, P(dcol), M, K, N, 1, 0, None))
    g = t_us(lambda: lib.seld_m_gemm_tn(P(a), P(dz), P(dw), P(db), P(slab), M, K, N, 0, 0, None))
    print(f"{tag:24s} M={M} K={K} N={N}: forward {f:7.1f} us, input gradient {d:7.1f} us, kernel gradient {g:7.1f} us", flush=True)
