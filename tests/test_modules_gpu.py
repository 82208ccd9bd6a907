"""mother_block / mother_stage on the device (seld_amd/modules.py: module operators through the C ABI, seld_m_*) against the fp64 oracle
restatement (oracle/modules_oracle.py), on the reference's own test configurations (modules_test.py:8-28, 154-200: 3 / 6 / 8 / 11
channels, strides (2,2) and (1,2), with and without squeeze-excite) and as FIRST block of models.seldnet in a train step."""
import copy

import numpy as np
import pytest
import torch

from helpers import check
from test_modules_cpu import BLOCK, BLOCK_SE, STAGE

pytestmark = pytest.mark.gpu


def _block_case(cfgs, in_shape, seed):
    """the blocks alone: forward output, and for a random upstream gradient the gradients of the input and of every variable"""
    from oracle import modules_oracle as M
    from seld_amd import modules
    B = in_shape[0]
    dev = torch.device("cuda", torch.cuda.current_device())
    rt = modules._Rt(dev)
    shape, blocks = tuple(in_shape[1:]), []
    for d, c in enumerate(cfgs):
        blocks.append(modules.MotherBlock(rt, c, shape, f"mb{d}", B))
        shape = blocks[-1].out_shape
    rt.finalize()
    tr, nt = [], []
    sh = tuple(in_shape[1:])
    for d, c in enumerate(cfgs):
        t, n, sh = M.mother_block_plan(c, sh, f"mb{d}")
        tr += t
        nt += n
    assert [(n, s) for n, _, s in rt.variables] == tr and [(n, s) for n, _, s in rt.state_variables] == nt and sh == shape
    rng = np.random.default_rng(seed)
    w = np.concatenate([rng.standard_normal(int(np.prod(s))) * (0.3 if n.endswith("kernel") else 0.1) + (1.0 if n.endswith("gamma") else 0.0) for n, s in tr])
    st = np.concatenate([(rng.standard_normal(int(np.prod(s))) * 0.1 if n.endswith("mean") else 1.0 + rng.random(int(np.prod(s)))) for n, s in nt])
    x = rng.standard_normal(in_shape)
    dy = rng.standard_normal((B,) + shape)
    rt.params[:rt.n_params].copy_(torch.as_tensor(w.astype(np.float32)))
    rt.state[:rt.n_state].copy_(torch.as_tensor(st.astype(np.float32)))
    xd = torch.as_tensor(x.astype(np.float32)).cuda()
    h = xd
    for blk in blocks:
        h = blk.forward(h, B, True)
    out = h.cpu().numpy().copy()
    g = torch.as_tensor(dy.astype(np.float32)).cuda()
    for blk in reversed(blocks):
        g = blk.backward(g, None, B)
    dx = g.cpu().numpy().copy()
    grads = rt.grads[:rt.n_params].cpu().numpy().copy()
    new_state = rt.state[:rt.n_state].cpu().numpy().copy()
    # ---- oracle
    from oracle import seldnet_oracle as O
    fw = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wd = O.unflatten(fw, tr)
    sd = O.unflatten(torch.tensor(st, dtype=torch.float64), nt)
    ns = {}
    ht = xt
    for d, c in enumerate(cfgs):
        ht = M.mother_block_forward(c, wd, sd, ns, ht, True, f"mb{d}")
    gw, gx = torch.autograd.grad((ht * torch.tensor(dy)).sum(), (fw, xt))
    check("mother forward", out, ht.detach().numpy())
    check("mother input gradient", dx, gx.numpy())
    off = 0
    for n, s in tr:
        k = int(np.prod(s))
        r = gw.numpy()[off:off + k]
        if n.endswith("bias") and ".c" in n or ".p" in n and n.endswith("bias"):
            # a bias in front of training-mode BatchNormalization: exactly 0 in exact arithmetic, rounding noise on both sides
            assert np.abs(grads[off:off + k]).max() <= 1e-3 * np.abs(gw.numpy()).max(), n
        else:
            check(f"mother grad {n}", grads[off:off + k], r)
        off += k
    check("mother BN moving statistics", new_state, np.concatenate([ns[n].detach().numpy().reshape(-1) for n, _ in nt]))


@pytest.mark.parametrize("name", ["stage", "block", "block_se"])
def test_mother_blocks_on_the_reference_test_configurations(name):
    """modules_test.py's own configurations and input shape (32 x 32 x 3), batch 8: forward, input gradient, every variable's gradient and
    the BatchNorm moving statistics against the fp64 oracle at 1e-4"""
    from oracle import modules_oracle as M
    cfgs = {"stage": M.stage_configs(STAGE), "block": [BLOCK], "block_se": [BLOCK_SE]}[name]
    _block_case(cfgs, (8, 32, 32, 3), seed=3)


def test_mother_block_every_branch_kind():
    """a configuration that exercises what the reference's tests do not: identity skips, projected skips on layers 0 / 1 / 2, a skipped first
    layer (alias) feeding a concatenating third layer, tanh / sigmoid activations"""
    full = {'filters0': 8, 'filters1': 8, 'filters2': 8, 'kernel_size0': 3, 'kernel_size1': 1, 'kernel_size2': 3, 'connect0': [1],
            'connect1': [1, 1], 'connect2': [1, 1, 1], 'strides': [1, 1], 'activation': 'tanh', 'squeeze_ratio': 0.25, 'se_activation': 'relu'}
    _block_case([full], (4, 10, 12, 5), seed=5)
    strided = dict(full, strides=[2, 3], connect2=[1, 0, 1], activation='swish')
    _block_case([strided], (4, 10, 12, 8), seed=6)
    # layers whose output reaches nothing (the reference's checks accept them, Keras' functional model drops them): no variables, no
    # computation, and the backward pass no longer aborts
    dead1 = dict(full, filters2=0, kernel_size2=0, connect2=[1, 1, 0], squeeze_ratio=0)          # second layer dead, output = concat(input, first layer)
    _block_case([dead1], (4, 10, 12, 5), seed=8)
    dead0 = dict(full, filters1=0, kernel_size1=0, connect1=[1, 0], connect2=[1, 0, 0])          # first layer dead, third layer on the input alone
    _block_case([dead0], (4, 10, 12, 8), seed=9)
    alias = {'filters0': 0, 'filters1': 6, 'filters2': 0, 'kernel_size0': 0, 'kernel_size1': 3, 'kernel_size2': 0, 'connect0': [1],
             'connect1': [1, 0], 'connect2': [1, 0, 1], 'strides': [1, 3], 'activation': 'relu'}      # model_config/SS5.json's BLOCK0 pattern
    _block_case([alias, dict(alias, strides=[1, 1])], (4, 12, 9, 4), seed=7)


@pytest.mark.parametrize("doa_loss", ["MSE", "MMSE"])
def test_train_step_with_a_mother_stage_first_block(seldnet_config, doa_loss):
    """models.seldnet with FIRST = mother_stage (the only conv FIRST-stage block the reference snapshot defines: modules.py:15-43,
    184-298), strides (5, 4) taking [T, 64, 7] to the label rate: one test step and one train step (train.py:22-44) against the fp64
    oracle — outputs, losses, every variable's gradient, BatchNorm state, the post-Adam weights — at 1e-4"""
    from oracle import modules_oracle as M
    from oracle import seldnet_oracle as O
    from seld_amd import losses, models, train
    cfg = copy.deepcopy(seldnet_config)
    cfg["FIRST"] = "mother_stage"
    cfg["FIRST_ARGS"] = {'depth': 2, 'filters0': 16, 'filters1': 24, 'filters2': 0, 'kernel_size0': 3, 'kernel_size1': 3, 'kernel_size2': 0,
                         'connect0': [1], 'connect1': [0, 1], 'connect2': [1, 0, 1], 'strides': [5, 4], 'activation': 'relu',
                         'squeeze_ratio': 0.5, 'se_activation': 'swish'}
    B, T = 3, 100
    in_shape = (B, T, 64, 7)
    tr, nt = M.variable_specs(cfg, in_shape)
    w, st = M.random_weights(cfg, in_shape, seed=11)
    x, ys, yd = O.synthetic_batch(B, T, seed=23)
    model = models.seldnet(in_shape, cfg)
    assert [(n, s) for n, _, s in model.variables] == tr and [(n, s) for n, _, s in model.state_variables] == nt
    model.set_weights(w, st)
    fw = torch.tensor(w, dtype=torch.float64)
    sed_t, doa_t, _ = M.forward(cfg, O.unflatten(fw, tr), O.unflatten(torch.tensor(st, dtype=torch.float64), nt), torch.tensor(x, dtype=torch.float64), False)
    y_t, sl_t, dl_t = train.teststep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.get_doa_loss(doa_loss))
    check("mother_stage model teststep sed", y_t[0].cpu().numpy(), sed_t.numpy())
    check("mother_stage model teststep doa", y_t[1].cpu().numpy(), doa_t.numpy())
    ref = M.train_step(cfg, in_shape, w, st, x, ys, yd, doa_loss=doa_loss, loss_weight=(1.0, 1000.0), lr=1e-3, step=1)
    y_p, sl, dl = train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.get_doa_loss(doa_loss), (1.0, 1000.0), train.Adam(1e-3))
    check("mother_stage model trainstep sed", y_p[0].cpu().numpy(), ref["sed"])
    check("mother_stage model trainstep doa", y_p[1].cpu().numpy(), ref["doa"])
    check("mother_stage model sloss", sl.cpu().numpy(), ref["sloss"])
    check("mother_stage model dloss", dl.cpu().numpy(), ref["dloss"])
    g = model.get_grads()
    for n, off, sh in model.variables:
        k = int(np.prod(sh))
        r = ref["grad"][off:off + k]
        if np.abs(r).max() < 1e-9 * np.abs(ref["grad"]).max():      # conv biases in front of training-mode BatchNormalization
            assert np.abs(g[off:off + k]).max() <= 1e-3 * np.abs(ref["grad"]).max(), n
            continue
        check(f"mother_stage model grad {n}", g[off:off + k], r)
    w1, st1 = model.get_weights()
    check("mother_stage model BN state", st1, ref["new_state"])
    big = np.abs(ref["grad"]) > 1e-3 * np.abs(ref["grad"]).max()
    assert np.abs(w1 - ref["new_w"])[big].max() <= 2e-3 * 1e-3 + 1e-7       # Adam's first step moves a weight by lr g / (|g| + eps)


STAGE_FIRST = {'depth': 2, 'filters0': 16, 'filters1': 24, 'filters2': 0, 'kernel_size0': 3, 'kernel_size1': 3, 'kernel_size2': 0,
               'connect0': [1], 'connect1': [0, 1], 'connect2': [1, 0, 1], 'strides': [5, 4], 'activation': 'relu',
               'squeeze_ratio': 0.5, 'se_activation': 'swish'}


def test_composed_model_runs_any_batch_up_to_the_one_it_was_built_for(seldnet_config):
    """train.main builds ONE model for max(train, val, test batch) and every loader ends on a ragged batch, so a
    composed model must take 1 <= B <= Bmax like SeldNet does.  A model built for 5 clips, given 3 (train step) and 1 (test step), gives
    bit for bit what models built for exactly 3 / 1 clips give; 6 clips are refused; the sliding-window inference path
    (evaluator.ensemble_outputs) takes a composed model and equals the forward on hand-framed windows."""
    from oracle import modules_oracle as M
    from oracle import seldnet_oracle as O
    from seld_amd import evaluator, losses, models, train
    cfg = copy.deepcopy(seldnet_config)
    cfg["FIRST"], cfg["FIRST_ARGS"] = "mother_stage", STAGE_FIRST
    T = 50
    w, st = M.random_weights(cfg, (5, T, 64, 7), seed=3)
    x, ys, yd = O.synthetic_batch(5, T, seed=9)
    args = (losses.BinaryCrossentropy(), losses.MSE)

    def run(Bm, B):
        m = models.seldnet((Bm, T, 64, 7), cfg)
        m.set_weights(w, st)
        yt, _, dt = train.teststep(m, x[:1], (ys[:1], yd[:1]), *args)
        yp, sl, dl = train.trainstep(m, x[:B], (ys[:B], yd[:B]), *args, (1.0, 1000.0), train.Adam(1e-3))
        torch.cuda.synchronize()
        return [t.cpu().numpy().copy() for t in (yt[0], yt[1], dt, yp[0], yp[1], sl, dl)] + [m.get_grads(), m.get_weights()[0]], m

    big, m5 = run(5, 3)
    exact, _ = run(3, 3)
    assert big[3].shape == (3, T // 5, 12) and big[0].shape == (1, T // 5, 12)
    for a, b in zip(big, exact):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        m5(np.zeros((6, T, 64, 7), np.float32))
    # evaluator.ensemble_outputs on a composed model: windows of T frames, step 5, batches of 4 (the last one ragged)
    clip = torch.as_tensor(np.random.default_rng(1).standard_normal((T + 45, 64, 7)).astype(np.float32)).cuda()
    (sed, doa), = evaluator.ensemble_outputs(m5, [clip], win_size=T, step_size=5, batch_size=4)
    wins = torch.stack([clip[5 * i:5 * i + T] for i in range(10)])
    s_ref = torch.cat([m5(wins[i:i + 5], training=False)[0].clone() for i in (0, 5)])
    L = T // 5
    acc, cnt = torch.zeros((9 + L, 12), device="cuda"), torch.zeros((9 + L, 1), device="cuda")
    for i in range(10):
        acc[i:i + L] += s_ref[i]
        cnt[i:i + L] += 1
    check("composed ensemble_outputs sed", sed.cpu().numpy(), (acc / cnt).cpu().numpy())
    assert tuple(doa.shape) == (9 + L, 36) and bool(torch.isfinite(doa).all())


def test_main_loop_with_a_mother_stage_first_block(tmp_path, seldnet_config, monkeypatch):
    """reference train.main (train.py:264-390) with FIRST = mother_stage: the epoch loop's train / val / test passes run batches smaller
    than the one the model was built for (one file per val / test batch, ragged last batches) through the composed model; the loss
    decreases, the best weights are saved and --resume loads them."""
    import json
    from seld_amd import params, train
    root = tmp_path / "DCASE2021" / "feat_label"
    feat, lab = root / "foa_dev_norm", root / "foa_dev_label"
    feat.mkdir(parents=True), lab.mkdir(parents=True)
    rng = np.random.default_rng(0)
    for fold in range(1, 7):
        name = f"fold{fold}_room1_mix000.npy"
        np.save(feat / name, rng.standard_normal((3000, 64, 7)).astype(np.float32))
        sed = (rng.random((600, 12)) < 0.1).astype(np.float32)
        vec = rng.standard_normal((600, 3, 12)); vec /= np.linalg.norm(vec, axis=1, keepdims=True)
        np.save(lab / name, np.concatenate([sed, (vec * sed[:, None, :]).reshape(600, 36)], -1).astype(np.float32))
    cfg = copy.deepcopy(seldnet_config)
    cfg["FIRST"], cfg["FIRST_ARGS"] = "mother_stage", dict(STAGE_FIRST, filters0=8, filters1=8)
    mcd = tmp_path / "model_config"
    mcd.mkdir()
    (mcd / "seldnet.json").write_text(json.dumps(cfg))
    monkeypatch.chdir(tmp_path)
    config, mc = params.get_param(["--name", "ms", "--abspath", str(tmp_path) + "/", "--batch", "16", "--loop_time", "1", "--epoch", "2", "--lr", "0.001"],
                                  model_config_dir=str(mcd))
    ds = [train.get_dataset(config, m) for m in ("train", "val", "test")]
    sizes = {int(np.asarray(b[0].shape[0] if hasattr(b[0], "shape") else len(b[0]))) for d in ds for b in d}
    assert len(sizes) > 1, "the three passes must present more than one batch size for this test to mean anything"
    model, hist = train.main((config, mc))
    assert type(model).__name__ == "ComposedSeldNet" and len(hist) == 2
    assert all(np.isfinite(h["score"]) and np.isfinite(h["test"][0]) for h in hist) and hist[-1]["train"][1] < hist[0]["train"][1]
    saved = list((tmp_path / "saved_model" / config.name).glob("bestscore_*.npz"))
    assert len(saved) == 1
    config_r, _ = params.get_param(["--name", "ms", "--abspath", str(tmp_path) + "/", "--batch", "16", "--loop_time", "1", "--epoch", "1", "--resume"],
                                   model_config_dir=str(mcd))
    _, h2 = train.main((config_r, mc))
    assert h2[0]["train"][1] < hist[0]["train"][1]
