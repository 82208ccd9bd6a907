import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import features_oracle as FO
from seld_amd import feature_extractor as FE
rng = np.random.default_rng(5)
n = 1440000
t = np.arange(n) / 24000.0
wav = (rng.standard_normal((4, n)) * 0.05).astype(np.float32)
wav[0] += (0.3 * np.sin(2 * np.pi * 440 * t)).astype(np.float32); wav[2] += (0.2 * np.sin(2 * np.pi * 1000 * t + 0.3)).astype(np.float32)
wav[:, n // 2: n // 2 + 48000] = 0.0          # a silent second: angle(0) = 0 bins
kw = dict(win_length=960, hop_length=480, n_fft=1024)
ref = FO.extract_features(wav, 24000, mode="mic", dtype=torch.float64, **kw)
for dft in (1, 0):
    fx = FE.FeatureExtractor(24000, "mic", 64, **kw)
    fx.set_option("dft", dft)
    got = fx(wav).cpu().numpy()
    e = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    print("dft", dft, got.shape, "log-mel %.3e" % e(got[..., :4], ref[..., :4]), "gcc %.3e" % e(got[..., 4:], ref[..., 4:]), "gcc max", np.abs(ref[..., 4:]).max())
