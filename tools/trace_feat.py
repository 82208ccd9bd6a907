"""Timeline of feat_dft_kernel inside one workgroup (diagnostic build):
    cd seld_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DFEAT_TRACE -c features.hip -o /tmp/feat_trace.o && \
        hipcc --offload-arch=gfx950 -shared -fPIC -o ../libseld_hip_trace.so $(ls *.o | grep -v '^features.o') /tmp/feat_trace.o
    SELD_HIP_LIB=$PWD/seld_amd/libseld_hip_trace.so python tools/trace_feat.py
Prints, for the 12 waves of workgroup (0, 0) and their frames 2 and 3, the cycles between stamps: channels 0..3 (load wait, window,
split, two matrix-core steps, power / intensity), intensity normalisation, the log-mel rows, the intensity rows, the store."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import _lib
from seld_amd import feature_extractor as FE

lib = _lib.load()
if not hasattr(lib, "seld_feat_trace_read"):
    print("(not a -DFEAT_TRACE build)")
    sys.exit(0)
rng = np.random.default_rng(0)
wavs = torch.as_tensor((rng.standard_normal((8, 4, 1440000)) * 0.1).astype(np.float32)).cuda()
fx = FE.FeatureExtractor(24000, "foa", 64, win_length=960, hop_length=480, n_fft=1024)
if len(sys.argv) > 1:
    fx.set_option("dbg", int(sys.argv[1]))          # ablation bits of the kernel under test (results are wrong then)
for _ in range(3):
    fx.batch(wavs)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5):
    fx.batch(wavs)
t1.record(); torch.cuda.synchronize()
print(f"dbg {sys.argv[1] if len(sys.argv) > 1 else 0}: {t0.elapsed_time(t1) / 5 * 1e3:.1f} us per 8-clip batch (trace build)")
buf = np.zeros((12, 2, 10), np.uint64)
assert lib.seld_feat_trace_read(C.c_void_p(buf.ctypes.data)) == 0
t = buf.astype(np.int64)
names = ["ch0", "ch1", "ch2", "ch3", "ivnorm", "logmel", "ivmel", "store"]
if os.environ.get("TRACE_BRIEF"):
    print("mean per phase:", " ".join(f"{n}={np.diff(t[:, :, :9], axis=2)[:, :, i].mean():.0f}" for i, n in enumerate(names)))
    sys.exit(0)
print("wave " + " ".join(f"{n:>7s}" for n in names) + "   frame   | next frame start - this start")
for w in range(12):
    for it in range(2):
        d = np.diff(t[w, it, :9])
        extra = f" | {int(t[w, 1, 0] - t[w, 0, 0])}" if it == 0 else ""
        print(f"w{w:<2d}{it} " + " ".join(f"{int(x):7d}" for x in d) + f" {int(t[w, it, 8] - t[w, it, 0]):7d}{extra}")
print("mean per phase:", " ".join(f"{n}={np.diff(t[:, :, :9], axis=2)[:, :, i].mean():.0f}" for i, n in enumerate(names)))
