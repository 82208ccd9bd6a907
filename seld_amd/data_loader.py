"""Data boundary of the reference, host side (numpy): the `.npy` dataset layout, fold selection and
windowing of data_loader.py, without tf.data.

  load_seldnet_data            data_loader.py:58-92   glob *.npy, fold = 5th character of the file name
  seldnet_data_to_dataloader   data_loader.py:132-168 + data_loader():13-55
      concat files -> [labels, 5, F, C] -> windows of 60 labels (300 frames) -> repeat(loop_time) ->
      batch(batch_size, drop_remainder=False) -> [device_transforms: augmentation on the device batch, run by
      the step loop] -> split labels into (sed, doa) -> shuffle (train)
Batches are (x [b,300,F,C] float32, (sed [b,60,C], doa [b,60,3C])) numpy arrays; pinned host buffers
and the H2D copy belong to the caller (`train.trainstep` accepts numpy or device tensors)."""
from __future__ import annotations

import os
from glob import glob

import numpy as np

SPLITS = {'train': [1, 2, 3, 4], 'val': [5], 'test': [6]}


def load_seldnet_data(feat_path, label_path, mode='train', n_freq_bins=64):
    assert mode in SPLITS
    out = []
    for path, what in ((feat_path, 'feat_path'), (label_path, 'label_path')):
        if not os.path.exists(path):
            raise ValueError(f'no such {what} ({path}) exists')
        files = sorted(glob(os.path.join(path, '*.npy')))
        out.append([np.load(f).astype('float32') for f in files
                    if int(f[f.rfind(os.path.sep) + 5]) in SPLITS[mode]])
    features, labels = out
    if features and features[0].ndim == 2:
        features = [np.reshape(x, (x.shape[0], -1, n_freq_bins)).transpose(0, 2, 1) for x in features]
    return features, labels


def split_total_labels_to_sed_doa(x, y):
    """transforms.py:117-119"""
    n_classes = y.shape[-1] // 4
    return x, (y[..., :n_classes], y[..., n_classes:])


class SeldDataset:
    """Iterable of batches with the structure the reference's tf.data pipeline yields."""

    def __init__(self, x, y, batch_size, train, loop_time, shuffle_size, seed=None, device_transforms=None):
        self.x, self.y = x, y
        self.batch_size, self.train, self.loop_time, self.shuffle_size = batch_size, train, loop_time, shuffle_size
        self.rng = np.random.default_rng(seed)
        # augmentations f(x_dev, y_total_dev, rng) -> (x_dev, y_total_dev) run by the step loop on the device batch
        # (seld_amd.transforms); with any of them the labels stay unsplit until they have run
        self.device_transforms = list(device_transforms or [])

    def to_device(self, device):
        """Keep the whole windowed dataset resident in HBM (DCASE dev set: 600 clips x 5.4 MB = 3.2 GB of 288 GB): a
        batch is then a device-side row gather and no byte crosses PCIe per step.  Batches become torch tensors."""
        import torch
        self.x = torch.as_tensor(self.x).to(device)
        self.y = torch.as_tensor(self.y).to(device)
        return self

    def __len__(self):
        n = self.x.shape[0] * (self.loop_time if self.train else 1)
        return -(-n // self.batch_size)

    def _batches(self):
        reps = self.loop_time if self.train else 1
        idx = np.concatenate([np.arange(self.x.shape[0])] * reps)      # cache().repeat(loop_time)
        for i in range(0, idx.size, self.batch_size):                  # batch(drop_remainder=False)
            sel = idx[i:i + self.batch_size]
            if not isinstance(self.x, np.ndarray):                     # HBM-resident: gather on the device
                import torch
                sel = torch.as_tensor(sel, device=self.x.device)
            if self.device_transforms:
                yield self.x[sel], self.y[sel]                         # total labels [b,60,4C]: split after the transforms
            else:
                yield split_total_labels_to_sed_doa(self.x[sel], self.y[sel])

    def __iter__(self):
        if not self.train or not self.shuffle_size or self.shuffle_size <= 1:
            yield from self._batches()
            return
        buf = []                                                       # tf.data shuffle(buffer) of BATCHES
        for b in self._batches():
            buf.append(b)
            if len(buf) >= self.shuffle_size:
                yield buf.pop(int(self.rng.integers(len(buf))))
        while buf:
            yield buf.pop(int(self.rng.integers(len(buf))))

    def take(self, n):
        for i, b in enumerate(self):
            if i >= n:
                break
            yield b


def seldnet_data_to_dataloader(features, labels, train=True, label_window_size=60, drop_remainder=True,
                               shuffle_size=None, batch_size=32, loop_time=1, seed=None, **kwargs):
    if kwargs.get('sample_transforms') or kwargs.get('preprocessing') or kwargs.get('batch_transforms'):
        raise ValueError('host-side tf.data transforms are not taken: pass device_transforms (seld_amd.transforms)')
    device_transforms = kwargs.get('device_transforms')
    total_length = labels[0].shape[0]
    features = np.concatenate(features, axis=0)
    labels = np.concatenate(labels, axis=0)
    features = np.reshape(features, (labels.shape[0], -1, *features.shape[1:]))    # [labels, 5, F, C]
    n_samples = features.shape[0] // label_window_size
    if not drop_remainder and features.shape[0] % label_window_size:
        raise ValueError('drop_remainder=False with a ragged last window is not supported')
    x = features[:n_samples * label_window_size].reshape(n_samples, -1, *features.shape[2:])
    y = labels[:n_samples * label_window_size].reshape(n_samples, label_window_size, -1)
    if not train:
        batch_size = total_length // label_window_size          # one file per batch (data_loader.py:157-158)
    if train and shuffle_size is None:
        shuffle_size = n_samples // batch_size
    return SeldDataset(np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32), batch_size, train,
                       loop_time, shuffle_size, seed, device_transforms)
