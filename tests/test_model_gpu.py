"""End-to-end parity of the HIP train/test step (through seld_amd -> C ABI) against the CPU oracle
on the same seeded inputs.  Tolerance 1e-4 relative (north_star), written per check."""
import numpy as np
import pytest
import torch

from helpers import check

pytestmark = pytest.mark.gpu


def _setup(seldnet_config, B, T, seed=0):
    from oracle import seldnet_oracle as O
    from seld_amd import models
    spec = O.Spec.from_config(seldnet_config)
    w, st = O.random_weights(spec, seed)
    x, ys, yd = O.synthetic_batch(B, T, seed=1234)
    model = models.seldnet((B, T, 64, 7), seldnet_config)
    assert model.n_params == O.param_count(spec) == 513840
    # the C-side layout and the oracle's must agree variable by variable
    tr, nt = O.variable_specs(spec)
    assert [(n, s) for n, _, s in model.variables] == tr
    assert [(n, s) for n, _, s in model.state_variables] == nt
    model.set_weights(w, st)
    return O, spec, model, w, st, x, ys, yd


def _per_var(model, name, got, ref, tol=1e-4, skip_conv_bias=True):
    worst = 0.0
    for n, off, sh in model.variables:
        k = int(np.prod(sh))
        if skip_conv_bias and n.startswith("conv") and n.endswith("bias"):
            # gradient of a bias in front of training-mode BatchNorm is exactly 0 in exact arithmetic;
            # both sides hold rounding noise only -> compare against the scale of the kernel gradient
            assert np.abs(got[off:off + k]).max() <= 1e-3 * max(1.0, np.abs(ref).max()), n
            continue
        worst = max(worst, check(f"{name} {n}", got[off:off + k], ref[off:off + k], tol))
    return worst


@pytest.mark.parametrize("B,T", [(2, 50), (3, 100)])
def test_forward_inference(seldnet_config, B, T):
    O, spec, model, w, st, x, ys, yd = _setup(seldnet_config, B, T)
    from seld_amd import losses, train
    ref = O.test_step(spec, w, st, x, ys, yd, "MSE", dtype=torch.float64)
    y_p, sl, dl = train.teststep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.MSE)
    check("teststep sed", y_p[0].cpu().numpy(), ref["sed"])
    check("teststep doa", y_p[1].cpu().numpy(), ref["doa"])
    check("teststep sloss", sl.cpu().numpy(), ref["sloss"])
    check("teststep dloss", dl.cpu().numpy(), ref["dloss"])
    sed, doa = model(x, training=False)
    check("model(x) sed", sed.cpu().numpy(), ref["sed"])


@pytest.mark.parametrize("B,T,doa_loss", [(2, 50, "MSE"), (2, 50, "MMSE"), (3, 100, "MSE")])
def test_train_step(seldnet_config, B, T, doa_loss):
    O, spec, model, w, st, x, ys, yd = _setup(seldnet_config, B, T)
    from seld_amd import losses, train
    ref = O.train_step(spec, w, st, x, ys, yd, doa_loss=doa_loss, loss_weight=(1.0, 1000.0), lr=1e-3, step=1, dtype=torch.float64)
    opt = train.Adam(1e-3)
    y_p, sl, dl = train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.get_doa_loss(doa_loss), (1.0, 1000.0), opt, False)
    check("trainstep sed", y_p[0].cpu().numpy(), ref["sed"])
    check("trainstep doa", y_p[1].cpu().numpy(), ref["doa"])
    check("trainstep sloss", sl.cpu().numpy(), ref["sloss"])
    check("trainstep dloss", dl.cpu().numpy(), ref["dloss"])
    g = model.get_grads()
    _per_var(model, "grad", g, ref["grad"])
    w1, st1 = model.get_weights()
    check("BN moving stats", st1, ref["new_state"])
    # Adam's first step is lr*sign(g)-like: compare the update where |g| is well above rounding noise
    upd, rupd = w1 - w, ref["new_w"] - w
    mask = np.abs(ref["grad"]) > 1e-3 * np.abs(ref["grad"]).max()
    check("adam update (|g| above noise)", upd[mask], rupd[mask], tol=2e-3)


def test_two_steps_and_short_batch(seldnet_config):
    """Second Adam step (bias correction t=2, non-zero slots) and a batch smaller than the ctx's."""
    O, spec, model, w, st, x, ys, yd = _setup(seldnet_config, 3, 50)
    from seld_amd import losses, train
    opt = train.Adam(1e-3)
    r1 = O.train_step(spec, w, st, x, ys, yd, doa_loss="MSE", step=1, dtype=torch.float64)
    train.trainstep(model, x, (ys, yd), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), opt)
    x2, ys2, yd2 = O.synthetic_batch(2, 50, seed=77)
    r2 = O.train_step(spec, r1["new_w"], r1["new_state"], x2, ys2, yd2, doa_loss="MSE", step=2, m=r1["m"], v=r1["v"], dtype=torch.float64)
    y_p, sl, dl = train.trainstep(model, x2, (ys2, yd2), losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), opt)
    # weights entering step 2 differ by Adam's first-step noise on near-zero gradients: looser bound
    check("step2 sed", y_p[0].cpu().numpy(), r2["sed"], tol=5e-3)
    check("step2 dloss", dl.cpu().numpy(), r2["dloss"], tol=5e-3)
