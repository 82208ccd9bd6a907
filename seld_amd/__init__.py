"""seld_amd — MI355X-native SELDnet train/inference hot path (HIP kernels behind a C ABI).

Mirrors the interface of the reference IRIS-AUDIO/SELD for this path:
    models.seldnet(input_shape, model_config)      (reference models.py:18-32)
    models.seldnet_v1(input_shape, model_config)   (reference models.py:36-52)
    train.trainstep / train.teststep               (reference train.py:22-44)
    losses.MMSE, losses.MSE, losses.BinaryCrossentropy   (reference losses.py:4-13, train.py:311-320)
PyTorch is used for device buffers, streams and torch.distributed only.
"""
from . import _lib  # noqa: F401
from . import losses, metrics, models, parallel, train  # noqa: F401

__all__ = ["models", "losses", "train"]
