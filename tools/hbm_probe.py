"""HBM write / read / copy rates at the size of the first block's pre-BN tensor (1.57 GB), via torch fills."""
import time
import torch

z = torch.empty(32, 3000, 64, 64, device="cuda")
y = torch.empty_like(z)
a_ = torch.randn(4096, 4096, device="cuda")
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    (a_ @ a_).sum().item()


def ev(fn, n=20):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


gb = z.numel() * 4 / 1e9
t = ev(lambda: z.fill_(1.0)); print("fill  %.3f ms  %.2f TB/s write" % (t, gb / t))
t = ev(lambda: z.sum()); print("sum   %.3f ms  %.2f TB/s read" % (t, gb / t))
t = ev(lambda: y.copy_(z)); print("copy  %.3f ms  %.2f TB/s read+write" % (t, 2 * gb / t))
