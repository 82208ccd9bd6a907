"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference's batch augmentations, written to follow
/root/reference/transforms.py step by step, with the random draws passed in (the reference draws them with
tf.random.uniform; values, not streams, are what can be compared).  PARITY UNPINNED: the reference holds no
fixtures for these functions; the restatement is anchored on the code lines cited per function.

Only tests/ may import this module."""
import numpy as np

# transforms.py:146-155 (arXiv 2101.02919 table 1): [[mic channel order], [signed 1-based foa channel order]]
CHANNEL_LIST = np.array([
    [[1, 3, 0, 2], [0, -3, -2, 1]],
    [[3, 1, 2, 0], [0, -3, 2, -1]],
    [[0, 1, 2, 3], [0, 1, 2, 3]],
    [[1, 0, 3, 2], [0, -1, -2, 3]],
    [[2, 0, 3, 1], [0, 3, -2, -1]],
    [[0, 2, 1, 3], [0, 3, 2, 1]],
    [[3, 2, 1, 0], [0, -1, 2, -3]],
    [[2, 3, 0, 1], [0, 1, -2, -3]],
])


def mask(specs, axis, size, offset, period=100):
    """transforms.py:6-44 with n_mask = 1.  specs [T,F,C] (one sample); size/offset: one draw per period-frame
    segment (arrays [T // period]).  axis -3 = time within the segment, -2 = frequency."""
    specs = np.array(specs, copy=True)
    T = specs.shape[0]
    if T % period != 0:
        raise ValueError("(spec time length / period)' rest must be 0")          # transforms.py:39-40
    seg = specs.reshape(T // period, period, *specs.shape[1:])                    # tf.signal.frame(specs, period, period, axis=0)
    total = seg.shape[1:][axis]                                                    # tf.shape(specs[:period])[axis]
    for s in range(seg.shape[0]):
        m = np.concatenate([np.ones(offset[s]), np.zeros(size[s]), np.ones(total - size[s] - offset[s])]).astype(specs.dtype)
        shape = [1] * (seg.ndim - 1)
        shape[axis] = total
        seg[s] = seg[s] * m.reshape(shape)
    return seg.reshape(specs.shape)


def foa_intensity_vec_aug(x, y, flip, p):
    """transforms.py:73-114.  x [B,T,F,7], y [B,S,4*nc]; flip int [B,3] in {0,1}; p int [B] in {0,2} (the reference's
    `2 * uniform(maxval=2)`)."""
    x, y = np.array(x, copy=True), np.array(y, copy=True)
    B = x.shape[0]
    nc = y.shape[-1] // 4
    y = y.reshape(*y.shape[:-1], 4, nc)
    iv = x[..., -3:]
    cart = y[..., -3:, :]
    f = flip.astype(np.float32)
    iv = (1 - 2 * f.reshape(B, 1, 1, 3)) * iv
    cart = (1 - 2 * f.reshape(B, 1, 3, 1)) * cart
    perm = np.stack([p, np.ones_like(p), 2 - p], axis=-1)                         # [B,3]
    check = (perm != np.array([[0, 1, 2]])).sum(-1, keepdims=True)
    feat_perm = (perm + check) % 3
    b = np.arange(B)[:, None]
    iv = iv.transpose(0, 3, 1, 2)[b, feat_perm].transpose(0, 2, 3, 1)             # gather(axis=-1, batch_dims=1)
    cart = cart.transpose(0, 2, 1, 3)[b, feat_perm].transpose(0, 2, 1, 3)         # gather(axis=-2, batch_dims=1)
    foa = x[..., 1:4].transpose(0, 3, 1, 2)[b, perm].transpose(0, 2, 3, 1)
    x = np.concatenate([x[..., :1], foa, iv], axis=-1)
    y = np.concatenate([y[..., :-3, :], cart], axis=-2)
    return x.astype(np.float32), y.reshape(*y.shape[:-2], 4 * nc).astype(np.float32)


def mic_gcc_perm(mic_perm):
    """transforms.py:122-140: position of the pair (mic_perm[a], mic_perm[b]) in the ordered pair list, per gcc pair (a, b)."""
    pairs = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
    decode = np.array([[0, 0, 1, 2], [0, 0, 3, 4], [1, 3, 0, 5], [2, 4, 5, 0]])
    mic_perm = np.asarray(mic_perm)
    return decode[mic_perm[:, pairs[:, 0]], mic_perm[:, pairs[:, 1]]]


def acs_aug(x, y, idx):
    """transforms.py:159-207.  x [B,T,F,17] (1+3 foa, 3 iv, 4 mic, 6 gcc), y [B,S,4*nc], idx int [B] in [0,8)."""
    x, y = np.array(x, copy=True), np.array(y, copy=True)
    B = x.shape[0]
    nc = y.shape[-1] // 4
    y = y.reshape(*y.shape[:-1], 4, nc)
    iv = x[..., 4:7]
    cart = y[..., -3:, :]
    flip = CHANNEL_LIST[idx]                                                        # [B,2,4]
    foa_flip = flip[:, 1, 1:]
    foa_sign = np.sign(foa_flip)
    foa_perm = foa_sign * foa_flip - 1
    check = (foa_perm != np.array([0, 1, 2])).sum(-1, keepdims=True)
    feat_perm = (foa_perm + check) % 3
    b = np.arange(B)[:, None]
    sgn = foa_sign.astype(np.float32)
    foa_x = x[..., 1:4].transpose(0, 3, 1, 2)[b, foa_perm].transpose(0, 2, 3, 1)
    iv = iv.transpose(0, 3, 1, 2)[b, feat_perm].transpose(0, 2, 3, 1) * sgn[:, None, None, :]
    cart = cart.transpose(0, 2, 1, 3)[b, feat_perm].transpose(0, 2, 1, 3) * sgn[:, None, :, None]
    mic_flip = flip[:, 0, :]
    gcc = x[..., 11:].transpose(0, 3, 1, 2)[b, mic_gcc_perm(mic_flip)].transpose(0, 2, 3, 1)
    mic_x = x[..., 7:11].transpose(0, 3, 1, 2)[b, mic_flip].transpose(0, 2, 3, 1)
    x = np.concatenate([x[..., :1], foa_x, iv, mic_x, gcc], axis=-1)
    y = np.concatenate([y[..., :-3, :], cart], axis=-2)
    return x.astype(np.float32), y.reshape(*y.shape[:-2], 4 * nc).astype(np.float32)
