"""Diagnostic for an intermittent mismatch seen ONCE in test_batch_extraction_equals_clip_by_clip[mic] (64 of 32 640 elements, GCC channels):
repeat the batch / clip-by-clip comparison in one process and print where they differ."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from seld_amd import feature_extractor as FE


def _wav(n, seed=0, scale=0.1):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 24000.0
    base = rng.standard_normal((4, n)) * scale
    base[0] += 0.3 * np.sin(2 * np.pi * 440 * t)
    base[1] += 0.2 * np.sin(2 * np.pi * 440 * t + 0.4)
    return base.astype(np.float32)


kw = dict(win_length=960, hop_length=480, n_fft=1024)
fx = FE.FeatureExtractor(24000, "mic", 64, **kw)
wavs = np.stack([_wav(24000 + 77, seed=s, scale=sc) for s, sc in ((1, 0.1), (2, 0.001), (3, 0.05))])
ref_b = fx.batch(wavs).cpu().numpy()
ref_s = [fx(wavs[i]).cpu().numpy() for i in range(3)]
bad = 0
for it in range(200):
    b = fx.batch(wavs).cpu().numpy()
    s = [fx(wavs[i]).cpu().numpy() for i in range(3)]
    for i in range(3):
        for name, a, r in (("batch", b[i], ref_b[i]), ("single", s[i], ref_s[i]), ("batch-vs-single", b[i], s[i])):
            if not np.array_equal(a, r):
                idx = np.argwhere(a != r)
                bad += 1
                print(f"it {it} clip {i} {name}: {len(idx)} differ; frames {sorted(set(idx[:, 0]))} mels {idx[:, 1].min()}..{idx[:, 1].max()} channels {sorted(set(idx[:, 2]))} "
                      f"max |d| {np.abs(a - r).max():.3e}", flush=True)
print("iterations with a difference:", bad)
