"""CPU oracle for the label/unit-conversion helpers on the data boundary of the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/seldnet_oracle.py header).  Restates, in numpy:
  feature_extractor.py:238-271  cartesian_to_polar / polar_to_cartesian (data_utils.py:13-18 degree<->radian)
  feature_extractor.py:91-114   extract_labels: csv rows (frame, class, _, azi, ele) -> [frames, 4*n_classes]
  feature_extractor.py:117-149  preprocess_features_labels: pad/trim to 600 labels / 3000 frames
  transforms.py:117-119         split_total_labels_to_sed_doa
  data_loader.py:132-156        windowing [N*3000,F,C] -> [N*10, 300, F, C]
Pinned by the reference's own known-answer tables (feature_extractor_test.py:8-22): tests/test_oracle_pins.py.
"""
import numpy as np


def cartesian_to_polar(c):
    c = np.asarray(c, dtype=np.float64)
    if c.shape[-1] != 3:
        raise ValueError("only 3D cartesian coordinates are allowed")
    x, y, z = c[..., 0], c[..., 1], c[..., 2]
    azi = np.arctan2(y, x) / np.pi * 180
    ele = np.arctan2(z, np.sqrt(x ** 2 + y ** 2)) / np.pi * 180
    return np.stack([azi, ele, np.sqrt(x ** 2 + y ** 2 + z ** 2)], axis=-1)


def polar_to_cartesian(p):
    p = np.asarray(p, dtype=np.float64)
    azi, ele = p[..., 0] * np.pi / 180, p[..., 1] * np.pi / 180
    r = p[..., 2] if p.shape[-1] == 3 else 1
    return np.stack([r * np.cos(azi) * np.cos(ele), r * np.sin(azi) * np.cos(ele), r * np.sin(ele)], axis=-1)


def labels_from_rows(rows, n_classes=14, max_frames=None):
    """rows: iterable of (frame, class, azimuth_deg, elevation_deg) ints."""
    rows = np.asarray(list(rows), dtype=np.float64)
    cart = polar_to_cartesian(rows[:, 2:])
    n = int(rows[:, 0].max()) + 1
    if max_frames is not None:
        n = max(max_frames, n)
    out = np.zeros((n, 4, n_classes), np.float32)
    for (frame, cls), xyz in zip(rows[:, :2].astype(int), cart):
        out[frame, :, cls] = [1.0, *xyz]
    return out.reshape(-1, 4 * n_classes)


def preprocess_features_labels(features, labels, max_label_length=600, multiplier=5):
    def fit(a, n):
        if a.shape[0] < n:
            return np.pad(a, ((0, n - a.shape[0]),) + ((0, 0),) * (a.ndim - 1))
        return a[:n]
    return fit(features, max_label_length * multiplier), fit(labels, max_label_length)


def split_total_labels_to_sed_doa(y):
    n = y.shape[-1] // 4
    return y[..., :n], y[..., n:]


def window(features_list, labels_list, label_window_size=60):
    f = np.concatenate(features_list, 0)
    l = np.concatenate(labels_list, 0)
    f = f.reshape(l.shape[0], -1, *f.shape[1:])
    n = f.shape[0] // label_window_size
    f = f[:n * label_window_size].reshape(n, label_window_size * f.shape[1], *f.shape[2:])
    l = l[:n * label_window_size].reshape(n, label_window_size, -1)
    return f, l
