import ctypes as C, os, sys, torch
sys.path.insert(0, '/root/repo')
from seld_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def run(M, N, K, dbg):
    a = torch.randn(M, K, device="cuda"); b = torch.randn(K, N, device="cuda"); c = torch.empty(M, N, device="cuda")
    lib.seld_k_set_option(b"gsb_dbg", dbg)
    f = lambda: lib.seld_k_gemm_sb(P(a), None, P(b), None, None, None, P(c), None, M, N, K, 0, 0, 0)
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    lib.seld_k_set_option(b"gsb_dbg", 0)
    return e0.elapsed_time(e1) / 20 * 1e3
for N, K in [(128, 1152), (256, 512), (128, 512), (128, 256)]:
    for M in (16384, 19200, 32768, 38400):
        print(f"N={N} K={K} M={M}: 16-wave {run(M, N, K, 8):7.1f} us ({(M + 127) // 128 * (N // 128)} workgroups), 4-wave {run(M, N, K, 4):7.1f} us", flush=True)
