#!/usr/bin/env python3
"""Where does the feature stage's time go?  Times a batch of 60-s clips (CLIPS, default 8) through the default kernel (foa, n_fft 1024:
feat_dft, the transform on the matrix cores — it carries no ablation hooks: they cost it 15 %) and through the radix-4 wave kernel
with parts of it switched off (seld_feat_set_option "dft" 0, "dbg": 1 no loads, 2 no FFT passes, 4 no mel projections; outputs
are then wrong)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seld_amd import feature_extractor as FE  # noqa: E402

n = 1440000
rng = np.random.default_rng(0)
clips = int(os.environ.get("CLIPS", "8"))
wav = torch.as_tensor((rng.standard_normal((clips, 4, n)) * 0.1).astype(np.float32)).cuda()
fx = FE.FeatureExtractor(24000, "foa", 64, win_length=960, hop_length=480, n_fft=1024)
for name, wk, dbg in [("feat_dft (default)", 2, 0), ("old workgroup kernel", 0, 0), ("wave kernel", 1, 0), ("  no loads", 1, 1), ("  no FFT passes", 1, 2), ("  no mel", 1, 4),
                      ("  no loads+FFT", 1, 3), ("  no FFT+mel", 1, 6), ("  nothing but bins/planes/stores", 1, 7),
                      ("  mel without LDS reads", 1, 8), ("  mel without stores", 1, 16), ("  mel without log10", 1, 32), ("  mel w/o reads+stores+log", 1, 56)]:
    fx.set_option("dft", 1 if wk == 2 else 0)
    fx.set_option("wave_kernel", 1 if wk else 0)
    fx.set_option("dbg", dbg)
    for _ in range(3):
        fx.batch(wav)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fx.batch(wav)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:36s} {e0.elapsed_time(e1) / 20 / clips * 1e3:8.1f} us per clip ({clips} clips per launch pair)")
