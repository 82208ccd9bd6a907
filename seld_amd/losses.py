"""Loss selectors mirroring the reference (losses.py:4-13; train.py:311-320).

The arithmetic is fused into the HIP loss kernel; these objects only select it, so that
`trainstep(model, x, y, sed_loss, doa_loss, loss_weight, optimizer, agc)` keeps the reference's
argument list (train.py:22)."""
from __future__ import annotations

from . import _lib


class _Loss:
    name = ""

    def __repr__(self):
        return f"<seld_amd.losses.{self.name}>"


class BinaryCrossentropy(_Loss):
    """tf.keras.losses.BinaryCrossentropy() (train.py:312-313)."""
    name = "BinaryCrossentropy"


class _MSE(_Loss):
    """tf.keras.losses.MSE *function*: mean over the last axis -> [B,S]; the resulting non-scalar
    `loss` is summed by tape.gradient (train.py:29-31)."""
    name = "MSE"
    code = _lib.SELD_DOA_MSE


class _MMSE(_Loss):
    """losses.MMSE (losses.py:4-13): masked MSE, scalar."""
    name = "MMSE"
    code = _lib.SELD_DOA_MMSE


class _MAE(_MSE):
    """tf.keras.losses.MAE *function* (`--doa_loss MAE`, params.py:16-17): mean |y - p| over the last axis -> [B,S], summed by
    tape.gradient like MSE."""
    name = "MAE"
    code = _lib.SELD_DOA_MAE


class _MSLE(_MSE):
    """tf.keras.losses.MSLE *function* (`--doa_loss MSLE`): mean over the last axis of (log(max(p, 1e-7) + 1) - log(max(y, 1e-7) + 1))^2."""
    name = "MSLE"
    code = _lib.SELD_DOA_MSLE


MSE = _MSE()
MMSE = _MMSE()
MAE = _MAE()
MSLE = _MSLE()


def get_doa_loss(name: str):
    """`getattr(tf.keras.losses, config.doa_loss)` / `getattr(losses, ...)` (train.py:317-320)."""
    table = {"MSE": MSE, "MMSE": MMSE, "MAE": MAE, "MSLE": MSLE}
    if name not in table:
        raise ValueError(f"doa_loss {name!r} has no MI355X kernel (built: {sorted(table)})")
    return table[name]
