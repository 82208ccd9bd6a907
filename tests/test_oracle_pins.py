"""Pins the CPU oracle (a) against every known answer the reference's own tests hold for this path
(parameter counts, output shapes, polar<->cartesian tables — copied here as DATA),
(b) against independent torch.nn implementations of the same ops, (c) against the committed golden
fixtures.  Numeric values of the model path are otherwise unpinned by the reference (see the header
of oracle/seldnet_oracle.py and DESIGN.md)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from oracle import labels_oracle as L
from oracle import seldnet_oracle as O


@pytest.fixture(scope="module")
def spec(seldnet_config):
    return O.Spec.from_config(seldnet_config)


# ---------------------------------------------------------------- (a) reference known answers
def test_param_counts(spec):
    # complexity.py:356-390,442-479 formulas; complexity_test.py:205-221,292-307 values
    assert O.param_count(spec) == 513840
    tr, nt = O.variable_specs(spec)
    per = {n: int(np.prod(s)) for n, s in tr}
    assert per["conv0.kernel"] + per["conv0.bias"] == 9 * 7 * 64 + 64
    assert per["conv1.kernel"] + per["conv1.bias"] == 9 * 64 * 64 + 64
    gru_layer = sum(v for k, v in per.items() if k.startswith("gru0."))
    assert gru_layer == 198144 == 2 * 3 * 128 * (128 + 128 + 2)      # gru_complexity: 3u(in+u+2), bi
    assert per["sed.dense0.kernel"] + per["sed.dense0.bias"] == (128 + 1) * 128   # linear_complexity
    assert per["sed.out.kernel"] + per["sed.out.bias"] == (128 + 1) * 12
    assert per["doa.out.kernel"] + per["doa.out.bias"] == (128 + 1) * 36
    assert sum(int(np.prod(s)) for _, s in nt) == 6 * 64


def test_gru_complexity_known_answer():
    # complexity_test.py:292-299: gru_complexity([32,100,20], 30, bi=True) -> params 9360
    s = O.Spec(in_ch=1, n_freq=20, filters=[1], pools=[(1, 1)], gru_units=[30], sed_units=[], doa_units=[], n_classes=1)
    tr, _ = O.variable_specs(s)
    assert sum(int(np.prod(sh)) for n, sh in tr if n.startswith("gru0.")) == 9360


def test_conv2d_complexity_known_answer():
    # complexity_test.py:205-213: conv2d_complexity([32,32,3], 16, 3) -> params 448
    assert 3 * 3 * 3 * 16 + 16 == 448


def test_output_shapes(spec):
    # modules_test.py:202-258 pattern (zeros in, shapes out); models.py:18-32 heads
    w, st = O.random_weights(spec, 0)
    x = np.zeros((2, 50, 64, 7), np.float32)
    r = O.test_step(spec, w, st, x, np.zeros((2, 10, 12), np.float32), np.zeros((2, 10, 36), np.float32))
    assert r["sed"].shape == (2, 10, 12) and r["doa"].shape == (2, 10, 36) and r["dloss"].shape == (2, 10)


CART = [[0, 0, 1], [0, -1, 0], [1, 0, 0], [-2, 2, 0], [0, 0, 0]]
POLAR = [[0, 90, 1], [-90, 0, 1], [0, 0, 1], [135, 0, np.sqrt(8)], [0, 0, 0]]


def test_polar_cartesian_tables():
    # feature_extractor_test.py:8-22,36-46
    np.testing.assert_allclose(L.cartesian_to_polar(CART), POLAR, atol=1e-6)
    np.testing.assert_allclose(L.polar_to_cartesian(POLAR), CART, atol=1e-6)


def test_label_layout_and_windowing():
    # feature_extractor.py:91-149, transforms.py:117-119, data_loader.py:132-156
    lab = L.labels_from_rows([(0, 1, 0, 0), (3, 5, 90, 0)], n_classes=12)
    assert lab.shape == (4, 48)
    sed, doa = L.split_total_labels_to_sed_doa(lab)
    assert sed.shape == (4, 12) and doa.shape == (4, 36) and sed[0, 1] == 1 and sed[3, 5] == 1
    np.testing.assert_allclose(doa[0, [1, 13, 25]], [1, 0, 0], atol=1e-7)      # x|y|z blocks of 12
    np.testing.assert_allclose(doa[3, [5, 17, 29]], [0, 1, 0], atol=1e-7)
    f, l = L.preprocess_features_labels(np.ones((3001, 64, 7), np.float32), lab)
    assert f.shape == (3000, 64, 7) and l.shape == (600, 48) and l[4:].sum() == 0
    fw, lw = L.window([f, f], [l, l])
    assert fw.shape == (20, 300, 64, 7) and lw.shape == (20, 60, 48)


# ---------------------------------------------------------------- (b) independent implementations
def test_gru_matches_torch_nn_gru():
    """torch.nn.GRU has the same reset_after formulation; gate order r|z|n vs Keras z|r|h."""
    rng = np.random.default_rng(0)
    B, S, I, u = 3, 17, 128, 128
    x = torch.tensor(rng.standard_normal((B, S, I)), dtype=torch.float64)
    k = torch.tensor(rng.standard_normal((I, 3 * u)) / np.sqrt(I))
    U = torch.tensor(rng.standard_normal((u, 3 * u)) / np.sqrt(u))
    b = torch.tensor(rng.standard_normal((2, 3 * u)) * 0.1)
    perm = torch.cat([torch.arange(u, 2 * u), torch.arange(0, u), torch.arange(2 * u, 3 * u)])  # z|r|h -> r|z|n
    g = torch.nn.GRU(I, u, batch_first=True).double()
    with torch.no_grad():
        g.weight_ih_l0.copy_(k[:, perm].T); g.weight_hh_l0.copy_(U[:, perm].T)
        g.bias_ih_l0.copy_(b[0, perm]); g.bias_hh_l0.copy_(b[1, perm])
    ref, _ = g(x)
    np.testing.assert_allclose(O.gru_direction(x, k, U, b, False).numpy(), ref.detach().numpy(), atol=1e-12)
    refb, _ = g(torch.flip(x, [1]))
    np.testing.assert_allclose(O.gru_direction(x, k, U, b, True).numpy(), torch.flip(refb, [1]).detach().numpy(), atol=1e-12)


def test_batchnorm_matches_torch():
    rng = np.random.default_rng(1)
    z = torch.tensor(rng.standard_normal((2, 5, 4, 64)) * 2 + 0.5)
    g, be = torch.tensor(rng.uniform(0.5, 1.5, 64)), torch.tensor(rng.normal(0, 0.2, 64))
    mm, mv = torch.zeros(64, dtype=torch.float64), torch.ones(64, dtype=torch.float64)
    y, nm, nv = O.batchnorm(z, g, be, mm, mv, True)
    rm, rv = mm.clone(), mv.clone()
    ref = torch.nn.functional.batch_norm(z.permute(0, 3, 1, 2), rm, rv, g, be, True, 1 - O.BN_MOMENTUM, O.BN_EPS).permute(0, 2, 3, 1)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), atol=1e-12)
    np.testing.assert_allclose(nm.numpy(), rm.numpy(), atol=1e-12)   # torch also uses the unbiased var for the running stat
    np.testing.assert_allclose(nv.numpy(), rv.numpy(), atol=1e-12)


def test_mmse_and_bce_numpy():
    rng = np.random.default_rng(2)
    _, ys, yd = O.synthetic_batch(2, 50, seed=3)
    p = rng.uniform(0.01, 0.99, ys.shape)
    d = rng.uniform(-1, 1, yd.shape)
    mask = np.round((yd.reshape(2, 10, 3, 12) ** 2).sum(2))
    mask3 = np.concatenate([mask] * 3, -1)
    np.testing.assert_allclose(O.mmse(torch.tensor(yd, dtype=torch.float64), torch.tensor(d)).item(),
                               (((yd - d) ** 2) * mask3).sum() / mask3.sum(), rtol=1e-12)
    pc = np.clip(p, 1e-7, 1 - 1e-7)
    np.testing.assert_allclose(O.bce(torch.tensor(ys, dtype=torch.float64), torch.tensor(p)).item(),
                               (-(ys * np.log(pc + 1e-7) + (1 - ys) * np.log(1 - pc + 1e-7))).mean(), rtol=1e-12)


def test_mse_quirk_is_sum_of_rows(spec):
    """train.py:29-31 with the Keras MSE function: gradient of SUM_{b,s}(w0*bce + w1*mse[b,s])."""
    sed = torch.rand(2, 10, 12, dtype=torch.float64, requires_grad=True)
    doa = torch.rand(2, 10, 36, dtype=torch.float64, requires_grad=True)
    _, ys, yd = O.synthetic_batch(2, 50, seed=5)
    obj, sl, dl = O.losses_and_objective(sed, doa, torch.tensor(ys, dtype=torch.float64), torch.tensor(yd, dtype=torch.float64), "MSE", (1.0, 1000.0))
    assert dl.shape == (2, 10)
    np.testing.assert_allclose(obj.item(), 20 * sl.item() + 1000 * dl.sum().item(), rtol=1e-12)


# ---------------------------------------------------------------- (c) golden fixtures
@pytest.mark.parametrize("name", ["b2_t50_mse", "b2_t50_mmse", "b3_t100_mse"])
def test_golden(spec, name):
    z = np.load(os.path.join(ROOT, "tests", "golden", f"seldnet_{name}.npz"))
    B, T, dl = (int(v) for v in z["meta"])
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(B, T, seed=1234)
    r = O.train_step(spec, w, st, x, ys, yd, doa_loss=["MSE", "MMSE"][dl], loss_weight=(1.0, 1000.0), lr=1e-3, step=1)  # fp32 oracle
    tol = dict(rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(r["sed"], z["train_sed"], **tol)
    np.testing.assert_allclose(r["doa"], z["train_doa"], **tol)
    np.testing.assert_allclose(r["dloss"], z["train_dloss"], **tol)
    g = r["grad"][::997]
    assert np.abs(g - z["grad_sample"]).max() <= 1e-4 * np.abs(z["grad_sample"]).max()
    np.testing.assert_allclose(r["new_state"], z["new_state"], **tol)


def test_frame_and_overlap_average_semantics():
    """evaluator.py:16-50 on the clip geometry: [3000,64,7] -> 541 windows of 300 -> 600 label frames."""
    from oracle import infer_oracle as IO
    x = np.arange(3000, dtype=np.float32)[:, None]
    f = IO.frame(x, 300, 5)
    assert f.shape == (541, 300, 1) and f[7, 0, 0] == 35 and f[540, 299, 0] == 2999
    y = np.ones((541, 60, 3))
    y[:, :, 1] = np.arange(541)[:, None]
    out = IO.overlap_average(y)
    assert out.shape == (600, 3)
    np.testing.assert_allclose(out[:, 0], 1.0)
    np.testing.assert_allclose(out[0, 1], 0.0)
    np.testing.assert_allclose(out[100, 1], np.mean(np.arange(41, 101)))


def test_metrics_oracle_hand_worked_case():
    """metrics.py:60-154 on a case small enough to work by hand: 1 clip, 1 block of 2 frames, 3 classes.
    class 0: active + detected, prediction 10 deg off -> TP;  class 1: active, missed -> FN;
    class 2: inactive, predicted -> FP.  => S = min(FP=1, FN=1) = 1, D = I = 0, ER = 1/2, F = 1/2... """
    from oracle import metrics_oracle as MO
    m = MO.SELDMetrics(doa_threshold=20, block_size=10, n_classes=3)
    sed_t = np.array([[[1, 1, 0], [1, 0, 0]]], float)
    sed_p = np.array([[[0.9, 0.1, 0.8], [0.7, 0.2, 0.1]]], float)
    v = lambda az: [np.cos(np.deg2rad(az)), np.sin(np.deg2rad(az)), 0.0]
    doa_t = np.zeros((1, 2, 3, 3)); doa_p = np.zeros((1, 2, 3, 3))
    for f in range(2):
        doa_t[0, f, :, 0] = v(30); doa_p[0, f, :, 0] = v(40)
    doa_t[0, 0, :, 1] = v(100)
    doa_p[0, 0, :, 2] = v(-50)
    m.update_states((sed_t, doa_t.reshape(1, 2, 9)), (sed_p, doa_p.reshape(1, 2, 9)))
    assert (m.TP, m.FP, m.FN, m.TN) == (1, 1, 1, 0)
    assert (m.S, m.D, m.I, m.Nref, m.Nsys, m.DE_TP) == (1, 0, 0, 2, 2, 1)
    np.testing.assert_allclose(m.total_DE, 10.0, atol=1e-9)
    ER, F, DE, DE_F = m.result()
    np.testing.assert_allclose([ER, F, DE, DE_F], [0.5, 0.5, 10.0, 0.5], atol=1e-9)
    np.testing.assert_allclose(MO.calculate_seld_score((ER, F, DE, DE_F)), (0.5 + 0.5 + 10 / 180 + 0.5) / 4)


def test_philox_known_answers_and_dropout_uniforms():
    """The dropout draws (oracle philox_uniform = loss_adam.hip::dropout_kernel) are Philox4x32-10: the three known-answer vectors
    Random123 publishes (kat_vectors: philox4x32 10), and the counter layout (element / 4, layer, step, 0) the kernel uses."""
    kat = (((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)))
    for ctr, key, want in kat:
        got = O.philox4x32_10(*[np.array([c], np.uint64) for c in ctr], *key)
        assert tuple(int(v[0]) for v in got) == want
    seed = 0x5e1d5e1d5e1d5e1d
    u = O.philox_uniform(10, seed, 3, 7)
    w = O.philox4x32_10(np.array([1], np.uint64), np.array([3], np.uint64), np.array([7], np.uint64), np.array([0], np.uint64), seed, seed >> 32)
    assert u.shape == (10,) and u[4] == (int(w[0][0]) >> 8) * 2.0 ** -24 and u[7] == (int(w[3][0]) >> 8) * 2.0 ** -24
    big = O.philox_uniform(200000, seed, 0, 0)
    assert 0.0 <= big.min() and big.max() < 1.0 and abs(big.mean() - 0.5) < 5e-3 and abs((big >= 0.25).mean() - 0.75) < 5e-3


def test_conv1d_same_matches_torch_conv1d_and_tensorflow_padding():
    """simple_dense_block's Conv1D(units, kernel_size, padding='same') (modules.py:370-372): odd kernels against torch's conv1d
    ('same' is symmetric then), an even kernel against TensorFlow's rule (the extra frame of padding goes behind)."""
    rng = np.random.default_rng(0)
    a = torch.as_tensor(rng.standard_normal((2, 9, 4)))
    for ks in (1, 3, 5):
        k = torch.as_tensor(rng.standard_normal((ks, 4, 6)))
        b = torch.as_tensor(rng.standard_normal(6))
        want = torch.nn.functional.conv1d(a.transpose(1, 2), k.permute(2, 1, 0), b, padding=ks // 2).transpose(1, 2)
        assert torch.allclose(O.conv1d_same(a, k, b), want, atol=1e-12)
    k = torch.as_tensor(rng.standard_normal((2, 4, 6)))
    got = O.conv1d_same(a, k, torch.zeros(6, dtype=torch.float64))
    want = a @ k[0] + torch.cat([a[:, 1:], torch.zeros(2, 1, 4, dtype=torch.float64)], dim=1) @ k[1]      # pad 0 in front, 1 behind
    assert torch.allclose(got, want, atol=1e-12)


def test_seldnet_v1_output_coupling(spec):
    """models.seldnet_v1 (models.py:36-52): the same network with doa_out = tanh(doa * Concatenate([sed] * 3))."""
    import copy
    w, st = O.random_weights(spec, 0)
    x, ys, yd = O.synthetic_batch(2, 50)
    a = O.test_step(spec, w, st, x, ys, yd, "MSE", dtype=torch.float64)
    sp1 = copy.deepcopy(spec)
    sp1.output_coupling = True
    b = O.test_step(sp1, w, st, x, ys, yd, "MSE", dtype=torch.float64)
    np.testing.assert_array_equal(a["sed"], b["sed"])
    np.testing.assert_allclose(b["doa"], np.tanh(a["doa"] * np.concatenate([a["sed"]] * 3, axis=-1)), rtol=0, atol=1e-15)


def test_gru_dropout_masks_follow_keras_implementation_2():
    """Keras GRUCell.call, implementation 2 (the default; modules.py:312-314 passes dropout = recurrent_dropout = rate): `inputs * dp_mask[0]`
    ahead of the kernel product, `h_tm1 = h_tm1 * rec_dp_mask[0]` ahead of the recurrent product — and, h_tm1 being reassigned, of the blend.
    Properties of the restatement: an all-ones mask changes nothing; a zero in the input mask = that kernel row zeroed; a zero in the state
    mask at unit k = recurrent row k zeroed AND unit k's output reduced to (1 - z) * hh (its own previous state never comes back); the
    masks scale by 1 / (1 - rate) and hold for the whole sequence."""
    from oracle import seldnet_oracle as O
    rng = np.random.default_rng(5)
    B, S, I, u = 2, 7, 6, 4
    x = torch.as_tensor(rng.standard_normal((B, S, I)))
    k = torch.as_tensor(rng.standard_normal((I, 3 * u)) * 0.5)
    r = torch.as_tensor(rng.standard_normal((u, 3 * u)) * 0.5)
    b = torch.as_tensor(rng.standard_normal((2, 3 * u)) * 0.1)
    for rev in (False, True):
        base = O.gru_direction(x, k, r, b, rev)
        same = O.gru_direction(x, k, r, b, rev, in_mask=torch.ones(B, I, dtype=x.dtype), rec_mask=torch.ones(B, u, dtype=x.dtype))
        np.testing.assert_allclose(same.numpy(), base.numpy(), rtol=0, atol=0)
        im = torch.ones(B, I, dtype=x.dtype); im[:, 2] = 0.0
        k0 = k.clone(); k0[2] = 0.0
        np.testing.assert_allclose(O.gru_direction(x, k, r, b, rev, in_mask=im).numpy(), O.gru_direction(x, k0, r, b, rev).numpy(), rtol=1e-12, atol=1e-12)
        rm = torch.ones(B, u, dtype=x.dtype); rm[:, 1] = 0.0
        got = O.gru_direction(x, k, r, b, rev, rec_mask=rm)
        r0 = r.clone(); r0[1] = 0.0
        other = O.gru_direction(x, k, r0, b, rev)              # unit 1 unseen by the recurrent product, but still blended with its own past
        keep = [0, 2, 3]
        np.testing.assert_allclose(got.numpy()[..., keep], other.numpy()[..., keep], rtol=1e-12, atol=1e-12)      # the other units see unit 1 through the recurrent product alone
        assert np.abs(got.numpy()[..., 1] - other.numpy()[..., 1]).max() > 1e-3                                  # unit 1 itself lost its own past in the blend
        # unit 1 by hand: z, hh from the masked state; h = (1 - z) * hh
        h = torch.zeros(B, u, dtype=x.dtype)
        gx = x @ k + b[0]
        for t in (range(S - 1, -1, -1) if rev else range(S)):
            hm = h * rm
            gh = hm @ r + b[1]
            z = torch.sigmoid(gx[:, t, :u] + gh[:, :u]); rr = torch.sigmoid(gx[:, t, u:2 * u] + gh[:, u:2 * u])
            hh = torch.tanh(gx[:, t, 2 * u:] + rr * gh[:, 2 * u:])
            h = z * hm + (1 - z) * hh
            np.testing.assert_allclose(got[:, t].numpy(), h.numpy(), rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(got[:, t, 1].numpy(), ((1 - z) * hh)[:, 1].numpy(), rtol=1e-12, atol=1e-12)
    m = O.dropout_mask((3, 8), 0.25, 0x5e1d5e1d5e1d5e1d, 96, 7, torch.float64).numpy()
    assert set(np.unique(m)) <= {0.0, 1.0 / 0.75} and 0 < (m == 0).sum() < m.size
    u_ = O.philox_uniform(24, 0x5e1d5e1d5e1d5e1d, 96, 7).reshape(3, 8)
    np.testing.assert_array_equal(m == 0, u_ < np.float32(0.25))


def test_first_and_second_block_dropout_in_the_oracle_forward(spec):
    """FIRST_ARGS / SECOND_ARGS dropout_rate reach the oracle's forward only in training; the draws are those of the given step."""
    import copy
    from oracle import seldnet_oracle as O
    sp = copy.deepcopy(spec)
    sp.conv_dropout, sp.gru_dropout = 0.2, 0.3
    w, st = O.random_weights(sp, 0)
    tr, nt = O.variable_specs(sp)
    wd, sd = O.unflatten(torch.as_tensor(w), tr), O.unflatten(torch.as_tensor(st), nt)
    x, _, _ = O.synthetic_batch(2, 40)
    xt = torch.as_tensor(x)
    with torch.no_grad():
        e0 = O.forward(spec, wd, sd, xt, training=False)[0]
        e1 = O.forward(sp, wd, sd, xt, training=False)[0]
        t0 = O.forward(spec, wd, sd, xt, training=True)[0]
        t1 = O.forward(sp, wd, sd, xt, training=True, dropout_step=3)[0]
        t1b = O.forward(sp, wd, sd, xt, training=True, dropout_step=3)[0]
        t2 = O.forward(sp, wd, sd, xt, training=True, dropout_step=4)[0]
    np.testing.assert_array_equal(e0.numpy(), e1.numpy())
    np.testing.assert_array_equal(t1.numpy(), t1b.numpy())
    assert np.abs(t0.numpy() - t1.numpy()).max() > 1e-5 and np.abs(t1.numpy() - t2.numpy()).max() > 1e-5


def test_fixture_margins_follow_the_rule():
    """every decision tensor of the four full-size fixtures — the MaxPool(ReLU) routings as well as the ReLU gates — lists its
    near-ties by ONE rule (tests/golden/make_golden_blocks.margin_rule: 8 x the fp32 oracle's own error on the value the decision is taken
    on, clamped to [1e-5, 2e-3]), and the fixtures carry the rule's input (`err32`) beside its output (`eps`): a change of fp32 summation
    order in a kernel re-rolls only decisions the rule already lists.  Also: the fp32 oracle itself takes no decision outside a list."""
    import importlib.util
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sp = importlib.util.spec_from_file_location("make_golden_blocks", os.path.join(here, "make_golden_blocks.py"))
    mb = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mb)
    assert mb.margin_rule(0.0) == 1e-5 and mb.margin_rule(1.0) == 2e-3 and abs(mb.margin_rule(1e-5) - 8e-5) < 1e-18
    n = 0
    for name in ("seldnet_full_b32_t3000_mse", "seldnet_full_b32_t3000_mmse", "xception_gru_full_b32_t3000_mse", "resnet50_gru_full_b16_t3000_mse"):
        z = np.load(os.path.join(here, name + ".npz"))
        keys = sorted({k.split(".")[1] for k in z.files if k.startswith("dec.") and k.endswith(".eps")})
        assert keys, name
        assert keys == sorted({k.split(".")[1] for k in z.files if k.startswith("dec.") and k.endswith(".near")}), f"{name}: a decision tensor without a stored margin"
        for k in keys:
            eps, err = float(z[f"dec.{k}.eps"]), float(z[f"dec.{k}.err32"])
            assert eps == mb.margin_rule(err), (name, k, eps, err)
            assert z[f"dec.{k}.near"].shape == z[f"dec.{k}.near_val"].shape
            n += 1
        if "fp32_flips" in z.files:      # [decisions the fp32 oracle takes differently, largest fp64 margin among them, how many outside eps]
            assert (z["fp32_flips"][:, 2] == 0).all(), f"{name}: the fp32 oracle flips a decision outside its list"
    assert n == 3 + 3 + 26 + 49


def test_oracle_given_its_own_decisions_is_itself(xception_config):
    """`routing=` (GIVEN decisions: the first block's MaxPool(ReLU) routing, xception_block's unit gates and exit routing — what the round-5
    fixtures use to evaluate the fp32 oracle ON the fp64 decisions) fed with the free-running evaluation's own decisions reproduces that
    evaluation bit for bit; with one gate complemented it does not."""
    import copy
    from oracle import seldnet_oracle as O
    cfg = copy.deepcopy(xception_config)
    cfg["FIRST_ARGS"]["block_num"] = 2
    spec = O.Spec.from_config(cfg)
    w, st = O.random_weights(spec, 3)
    x, ys, yd = O.synthetic_batch(2, 50, seed=5)
    kw = dict(doa_loss="MSE", loss_weight=(1.0, 1000.0), lr=1e-3, step=1, dtype=torch.float64)
    rec = {}
    free = O.train_step(spec, w, st, x, ys, yd, record_routing=rec, **kw)
    routing = {k: ((v["pos"], v["gate"]) if "pos" in v else v["gate"]) for k, v in rec.items()}
    assert set(routing) == {0, "exit"} | {f"xc{b}.{u}.in" for b in range(2) for u in range(3)}
    given = O.train_step(spec, w, st, x, ys, yd, routing=routing, **kw)
    assert np.array_equal(given["grad"], free["grad"]) and np.array_equal(given["sed"], free["sed"])
    routing["xc1.0.in"] = ~routing["xc1.0.in"]
    other = O.train_step(spec, w, st, x, ys, yd, routing=routing, **kw)
    assert np.abs(other["grad"] - free["grad"]).max() > 1e-3 * np.abs(free["grad"]).max()
