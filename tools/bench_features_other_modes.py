"""Feature-stage timing of the configurations the headline leg does not cover (mic at n_fft 1024, foa at the function-default n_fft 512),
8 clips of 60 s per launch pair; prints ms per clip."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from seld_amd import feature_extractor as FE
n, clips = 1440000, 8
rng = np.random.default_rng(0)
wavs = torch.as_tensor((rng.standard_normal((clips, 4, n)) * 0.1).astype(np.float32)).cuda()
for mode, kw in (("mic", dict(win_length=960, hop_length=480, n_fft=1024)), ("foa", dict(win_length=400, hop_length=256, n_fft=512))):
    fx = FE.FeatureExtractor(24000, mode, 64, **kw)
    for _ in range(2): fx.batch(wavs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fx.batch(wavs)
    torch.cuda.synchronize()
    print(mode, kw["n_fft"], "%.4f ms per clip (batch of 8)" % ((time.perf_counter() - t0) / 5 / clips * 1e3))
