"""Device augmentation (seld_aug_mask / seld_aug_gather_sign behind seld_amd.transforms) against the numpy oracle
on the same random draws: pure data movement and sign flips, so the comparison is bit-exact."""
import numpy as np
import pytest
import torch

from oracle import transforms_oracle as TO

pytestmark = pytest.mark.gpu


def test_mask_time_and_frequency(seld_lib):
    from seld_amd import transforms as T
    rng = np.random.default_rng(0)
    B, Tn, F, Cc = 5, 300, 64, 7
    x = rng.standard_normal((B, Tn, F, Cc)).astype(np.float32)
    ts, to = T.draw_mask(rng, B * 3, 100, 24)
    fs, fo = T.draw_mask(rng, B * 3, 64, 16)
    xd = torch.as_tensor(x).cuda()
    T.mask(xd, -3, draws=(ts, to))
    T.mask(xd, -2, draws=(fs, fo))
    ref = np.stack([TO.mask(TO.mask(x[b], -3, ts[3 * b:3 * b + 3], to[3 * b:3 * b + 3]), -2, fs[3 * b:3 * b + 3], fo[3 * b:3 * b + 3])
                    for b in range(B)])
    np.testing.assert_array_equal(xd.cpu().numpy(), ref)
    # the generator path draws within the reference's ranges and zeroes something
    xg = torch.as_tensor(x).cuda()
    T.mask(xg, -3, max_mask_size=24, rng=np.random.default_rng(1))
    frac = float((xg == 0).float().mean())
    assert 0.0 < frac < 0.24
    with pytest.raises(ValueError):
        T.mask(torch.zeros(2, 250, 64, 7, device="cuda"), -3)             # T % period != 0 (transforms.py:39-40)
    with pytest.raises(ValueError):
        T.mask(torch.zeros(2, 300, 64, 7), -3)                            # host tensor: no CPU path


@pytest.mark.parametrize("B,Tn", [(4, 20), (32, 300)])
def test_foa_intensity_vec_aug(seld_lib, B, Tn):
    from seld_amd import transforms as T
    rng = np.random.default_rng(2)
    x = rng.standard_normal((B, Tn, 64, 7)).astype(np.float32)
    y = rng.standard_normal((B, Tn // 5, 48)).astype(np.float32)
    flip, p = rng.integers(0, 2, (B, 3)), 2 * rng.integers(0, 2, B)
    xd, yd = torch.as_tensor(x).cuda(), torch.as_tensor(y).cuda()
    xo, yo = T.foa_intensity_vec_aug(xd, yd, draws=(flip, p))
    xr, yr = TO.foa_intensity_vec_aug(x, y, flip, p)
    np.testing.assert_array_equal(xo.cpu().numpy(), xr)
    np.testing.assert_array_equal(yo.cpu().numpy(), yr)
    _, (sed, doa) = T.split_total_labels_to_sed_doa(xo, yo)
    assert sed.shape[-1] == 12 and doa.shape[-1] == 36


def test_acs_aug(seld_lib):
    from seld_amd import transforms as T
    rng = np.random.default_rng(4)
    B = 16
    x = rng.standard_normal((B, 30, 64, 17)).astype(np.float32)
    y = rng.standard_normal((B, 6, 56)).astype(np.float32)
    idx = np.concatenate([np.arange(8), rng.integers(0, 8, 8)])
    xd, yd = torch.as_tensor(x).cuda(), torch.as_tensor(y).cuda()
    xo, yo = T.acs_aug(xd, yd, draws=idx)
    xr, yr = TO.acs_aug(x, y, idx)
    np.testing.assert_array_equal(xo.cpu().numpy(), xr)
    np.testing.assert_array_equal(yo.cpu().numpy(), yr)
    with pytest.raises(ValueError):
        T.acs_aug(torch.zeros(2, 10, 64, 7, device="cuda"), torch.zeros(2, 2, 48, device="cuda"))
