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
