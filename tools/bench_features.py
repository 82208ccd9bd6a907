#!/usr/bin/env python3
"""Feature-stage throughput on one MI355X: 60 s FOA clips [4, 1 440 000] -> [3001, 64, 7]
(feature_extractor.py:294-301 parameters).  Prints one JSON line; not the headline metric."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seld_amd import feature_extractor as FE  # noqa: E402


def main():
    n, clips = 1440000, int(sys.argv[1]) if len(sys.argv) > 1 else 8      # clips per launch pair
    rng = np.random.default_rng(0)
    wavs = torch.as_tensor((rng.standard_normal((clips, 4, n)) * 0.1).astype(np.float32)).cuda()
    fx = FE.FeatureExtractor(24000, "foa", 64, win_length=960, hop_length=480, n_fft=1024)
    for _ in range(2):
        fx.batch(wavs)                      # the clips of a batch in one pair of launches (seld_feat_extract_batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        out = fx.batch(wavs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = dt / (reps * clips)
    bytes_clip = 4 * n * 4 + 3001 * 64 * 7 * 4
    print(json.dumps({"stage": f"feature_extractor foa 1024/960/480 -> [3001,64,7], {clips} clips per launch pair", "clips_per_s": round(1 / per, 1),
                      "ms_per_clip": round(per * 1e3, 4), "algorithmic_GBps": round(bytes_clip / per / 1e9, 1),
                      "algorithmic_bytes_per_clip": bytes_clip}))


if __name__ == "__main__":
    main()
