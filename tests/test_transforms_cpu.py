"""Augmentation (reference transforms.py): the oracle's own invariants and the host-side index tables of
seld_amd.transforms against it.  No GPU: the tables are applied with numpy here; the device kernels are compared
with the oracle in tests/test_transforms_gpu.py."""
import numpy as np
import pytest

from oracle import transforms_oracle as TO
from seld_amd import transforms as T


def _apply(t, src, sgn, R, inner):
    """numpy statement of seld_aug_gather_sign: out[b,o,r,i] = sgn[b,r] * in[b,o,src[b,r],i]"""
    B = t.shape[0]
    v = t.reshape(B, -1, R, inner)
    out = np.empty_like(v)
    for b in range(B):
        out[b] = (v[b][:, src[b], :] * sgn[b][None, :, None]).astype(np.float32)
    return out.reshape(t.shape)


def test_mic_gcc_perm_matches_oracle_and_identity():
    perms = np.array([c[0] for c in TO.CHANNEL_LIST.tolist()])
    np.testing.assert_array_equal(T.mic_gcc_perm(perms), TO.mic_gcc_perm(perms))
    np.testing.assert_array_equal(T.mic_gcc_perm(np.array([[0, 1, 2, 3]]))[0], np.arange(6))
    # a gcc pair (a, b) of the permuted array is the pair (perm[a], perm[b]) of the original one
    np.testing.assert_array_equal(TO.mic_gcc_perm(np.array([[1, 0, 3, 2]]))[0], [0, 4, 3, 2, 1, 5])


def test_mic_gcc_perm_reference_known_answer_table():
    """The reference's own 6-row known-answer table for mic_gcc_perm (transforms_test.py:64-73, data only): both the
    product's host function and the oracle must reproduce every row."""
    mic_perm = np.array([[1, 3, 0, 2], [3, 1, 2, 0], [1, 0, 3, 2], [2, 0, 3, 1], [0, 2, 1, 3], [3, 2, 1, 0]], np.int32)
    res = np.array([[4, 0, 3, 2, 5, 1],
                    [4, 5, 2, 3, 0, 1],
                    [0, 4, 3, 2, 1, 5],
                    [1, 5, 3, 2, 0, 4],
                    [1, 0, 2, 3, 5, 4],
                    [5, 4, 2, 3, 1, 0]], np.int32)
    np.testing.assert_array_equal(TO.mic_gcc_perm(mic_perm), res)
    np.testing.assert_array_equal(T.mic_gcc_perm(mic_perm), res)


def test_channel_list_is_the_references_table():
    np.testing.assert_array_equal(np.array(T.channel_list), TO.CHANNEL_LIST)
    assert TO.CHANNEL_LIST.shape == (8, 2, 4)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_foa_tables_reproduce_the_oracle(seed):
    rng = np.random.default_rng(seed)
    B, Tn, F, S, nc = 6, 20, 8, 4, 12
    x = rng.standard_normal((B, Tn, F, 7)).astype(np.float32)
    y = rng.standard_normal((B, S, 4 * nc)).astype(np.float32)
    flip, p = rng.integers(0, 2, (B, 3)), 2 * rng.integers(0, 2, B)
    xr, yr = TO.foa_intensity_vec_aug(x, y, flip, p)
    xs, xg, ys, yg = T.foa_tables(flip, p)
    np.testing.assert_array_equal(_apply(x, xs, xg, 7, 1), xr)
    np.testing.assert_array_equal(_apply(y, ys, yg, 4, nc), yr)


def test_foa_aug_keeps_features_and_labels_consistent():
    """The intensity vector points at the source: if the IV channels carry the label's cartesian vector, they still do
    after the augmentation (same permutation and signs on both, transforms.py:95-105); activity and W stay put."""
    rng = np.random.default_rng(3)
    B, S, nc = 8, 5, 1
    y = rng.standard_normal((B, S, 4 * nc)).astype(np.float32)
    x = np.zeros((B, S, 1, 7), np.float32)
    x[..., 0, 0] = y[..., 0]
    x[..., 0, 4:7] = y[..., 1:4]
    x[..., 0, 1:4] = rng.standard_normal((B, S, 3))
    flip, p = rng.integers(0, 2, (B, 3)), 2 * rng.integers(0, 2, B)
    xr, yr = TO.foa_intensity_vec_aug(x, y, flip, p)
    np.testing.assert_array_equal(xr[..., 0, 4:7], yr[..., 1:4])
    np.testing.assert_array_equal(xr[..., 0, 0], y[..., 0])
    np.testing.assert_array_equal(yr[..., 0], y[..., 0])
    np.testing.assert_allclose(np.linalg.norm(yr[..., 1:4], axis=-1), np.linalg.norm(y[..., 1:4], axis=-1), rtol=1e-6)
    # no flip, no swap = identity
    x0, y0 = TO.foa_intensity_vec_aug(x, y, np.zeros((B, 3), int), np.zeros(B, int))
    np.testing.assert_array_equal(x0, x)
    np.testing.assert_array_equal(y0, y)


@pytest.mark.parametrize("seed", [0, 5])
def test_acs_tables_reproduce_the_oracle(seed):
    rng = np.random.default_rng(seed)
    B, Tn, F, S, nc = 8, 10, 4, 3, 14
    x = rng.standard_normal((B, Tn, F, 17)).astype(np.float32)
    y = rng.standard_normal((B, S, 4 * nc)).astype(np.float32)
    idx = np.arange(8) if seed == 0 else rng.integers(0, 8, B)          # seed 0: every row of the table once
    xr, yr = TO.acs_aug(x, y, idx)
    xs, xg, ys, yg = T.acs_tables(idx)
    np.testing.assert_array_equal(_apply(x, xs, xg, 17, 1), xr)
    np.testing.assert_array_equal(_apply(y, ys, yg, 4, nc), yr)
    i2 = np.flatnonzero(idx == 2)                                        # row 2 of the table is the identity
    np.testing.assert_array_equal(xr[i2], x[i2])
    np.testing.assert_array_equal(yr[i2], y[i2])


def test_mask_oracle_and_draws():
    rng = np.random.default_rng(7)
    spec = rng.standard_normal((300, 64, 7)).astype(np.float32) + 5.0      # no zeros in the input
    size, offset = T.draw_mask(rng, 3, 100, 24)
    assert size.min() >= 0 and size.max() < 24 and (offset >= 0).all() and (offset + size <= 100).all()
    out = TO.mask(spec, -3, size, offset)
    for s in range(3):
        seg = out[100 * s:100 * (s + 1)]
        dead = np.flatnonzero((seg == 0).all(axis=(1, 2)))
        np.testing.assert_array_equal(dead, np.arange(offset[s], offset[s] + size[s]))
    fs, fo = T.draw_mask(rng, 3, 64, 16)
    out = TO.mask(spec, -2, fs, fo)
    for s in range(3):
        dead = np.flatnonzero((out[100 * s:100 * (s + 1)] == 0).all(axis=(0, 2)))
        np.testing.assert_array_equal(dead, np.arange(fo[s], fo[s] + fs[s]))
    with pytest.raises(ValueError):
        TO.mask(spec[:250], -3, size, offset)
    with pytest.raises(ValueError):
        T.draw_mask(rng, 1000, 16, 17)           # a mask as long as the axis leaves no room for the offset draw
