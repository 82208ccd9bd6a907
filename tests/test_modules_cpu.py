"""mother_block / mother_stage (reference modules.py:15-43, 184-298): the oracle restatement against the reference's OWN known answers for
these blocks — the output shapes of modules_test.py:8-28 and :154-200 — and the configuration errors of modules.py:202-222, which the
product side (seld_amd.modules.check_mother_config) must raise identically.  No GPU."""
import copy

import numpy as np
import pytest
import torch

from oracle import modules_oracle as M

# modules_test.py:9-21 (test_mother_stage): exp_input_shape 32,32,32,3 -> exp_output_shape 32,16,16,6
STAGE = {'depth': 2, 'filters0': 3, 'filters1': 3, 'filters2': 6, 'kernel_size0': 1, 'kernel_size1': 3, 'kernel_size2': 1,
         'connect0': [0], 'connect1': [0, 0], 'connect2': [1, 0, 0], 'strides': [2, 2], 'activation': 'relu'}
# modules_test.py:155-166 (test_mother_block): 32,32,32,3 -> 32,32,16,11; :178-191 the same with squeeze_ratio 0.5, se_activation swish
BLOCK = {'filters0': 6, 'filters1': 8, 'filters2': 0, 'kernel_size0': 3, 'kernel_size1': 3, 'kernel_size2': 0, 'connect0': [0],
         'connect1': [0, 1], 'connect2': [1, 0, 1], 'strides': [1, 2], 'activation': 'relu'}
BLOCK_SE = dict(BLOCK, squeeze_ratio=0.5, se_activation='swish')


def _run(cfgs, in_shape, seed=0):
    shape, tr, nt = in_shape[1:], [], []
    for d, c in enumerate(cfgs):
        t, n, shape = M.mother_block_plan(c, shape, f"mb{d}")
        tr += t
        nt += n
    rng = np.random.default_rng(seed)
    w = {n: torch.tensor(rng.standard_normal(s) * 0.1) for n, s in tr}
    st = {n: torch.tensor(np.abs(rng.standard_normal(s)) + 0.5) for n, s in nt}
    h = torch.tensor(rng.standard_normal(in_shape))
    new = {}
    for d, c in enumerate(cfgs):
        h = M.mother_block_forward(c, w, st, new, h, True, f"mb{d}")
    return tuple(h.shape), (in_shape[0],) + tuple(shape), tr, nt, new


@pytest.mark.parametrize("cfgs,expect", [(M.stage_configs(STAGE), (16, 16, 6)), ([BLOCK], (32, 16, 11)), ([BLOCK_SE], (32, 16, 11))])
def test_output_shapes_are_the_reference_tests_expectations(cfgs, expect):
    got, planned, tr, nt, new = _run(cfgs, (4, 32, 32, 3))
    assert got == (4,) + expect == planned
    assert set(new) == {n for n, _ in nt}          # every BatchNormalization updated its moving statistics


def test_stage_applies_the_strides_in_its_first_block_only():
    cfgs = M.stage_configs(STAGE)
    assert len(cfgs) == 2 and tuple(cfgs[0]["strides"]) == (2, 2) and tuple(cfgs[1]["strides"]) == (1, 1)      # modules.py:38-42


def test_variable_layout_follows_the_layer_creation_order():
    tr, nt, out = M.mother_block_plan(BLOCK_SE, (32, 32, 3), "mb0")
    names = [n for n, _ in tr]
    # layer 0: Conv2D, BatchNormalization; layer 1: Conv2D, BN, then the projection of outputs[1] (6 -> 8 channels at stride (1,2));
    # layer 2 skipped: the strided 1x1 of outputs[0] for the concatenation; then the two squeeze-excite convolutions
    assert names == ["mb0.c0.kernel", "mb0.c0.bias", "mb0.bn0.gamma", "mb0.bn0.beta", "mb0.c1.kernel", "mb0.c1.bias", "mb0.bn1.gamma", "mb0.bn1.beta",
                     "mb0.p1_1.kernel", "mb0.p1_1.bias", "mb0.pbn1_1.gamma", "mb0.pbn1_1.beta", "mb0.s2_0.kernel", "mb0.s2_0.bias",
                     "mb0.se0.kernel", "mb0.se0.bias", "mb0.se1.kernel", "mb0.se1.bias"]
    shapes = dict(tr)
    assert shapes["mb0.c0.kernel"] == (3, 3, 3, 6) and shapes["mb0.p1_1.kernel"] == (1, 1, 6, 8) and shapes["mb0.s2_0.kernel"] == (1, 1, 3, 3)
    assert shapes["mb0.se0.kernel"] == (1, 1, 11, 5) and shapes["mb0.se1.kernel"] == (1, 1, 5, 11) and out == (32, 16, 11)      # int(0.5 * 11) = 5


BAD = [({"filters0": 0, "kernel_size0": 3}, "0\\) skipped layer"), ({"filters1": 8, "kernel_size1": 0}, "1\\) skipped layer"),
       ({"filters0": 0, "kernel_size0": 0, "connect0": [1], "connect1": [0, 1]}, "cannot link skipped layer \\(first layer\\)"),
       ({"filters1": 0, "kernel_size1": 0, "strides": [1, 1], "connect1": [1, 0], "connect2": [0, 0, 1]}, "cannot link skipped layer \\(second layer\\)"),
       ({"filters0": 0, "kernel_size0": 0, "connect0": [0], "connect1": [1, 0], "connect2": [1, 0, 0]}, "zero inputs to the second layer"),
       ({"filters1": 0, "kernel_size1": 0, "strides": [1, 1], "connect1": [0, 0], "connect2": [1, 0, 0]}, "zero inputs to the third layer"),
       ({"connect2": [0, 0, 0]}, "zero inputs to the final output"),
       ({"filters1": 0, "kernel_size1": 0, "connect1": [1, 0], "connect2": [1, 0, 0]}, "the second layer must be active")]


@pytest.mark.parametrize("patch,msg", BAD)
def test_configuration_errors_of_the_reference(patch, msg):
    """modules.py:202-222: the oracle and the product's host side refuse the same configurations with the same messages"""
    from seld_amd import modules
    cfg = copy.deepcopy(BLOCK)
    cfg.update(patch)
    with pytest.raises(ValueError, match=msg):
        M.check_mother_config(cfg)
    with pytest.raises(ValueError, match=msg):
        modules.check_mother_config(cfg)
    modules.check_mother_config(BLOCK)
    modules.check_mother_config(BLOCK_SE)


def test_same_padding_is_tensorflows():
    """Conv2D(k, 'same', strides): ceil(in / stride) outputs, the odd padding element at the END (TensorFlow), checked on a delta image"""
    x = torch.zeros(1, 5, 6, 1, dtype=torch.float64)
    x[0, 0, 0, 0] = 1.0
    k = torch.arange(9, dtype=torch.float64).reshape(3, 3, 1, 1)
    y = M.conv2d_same(x, k, None, (2, 2))
    assert tuple(y.shape) == (1, 3, 3, 1)
    # H = 5, stride 2: pad_total = (3-1)*2 + 3 - 5 = 2 -> 1 before; W = 6: pad_total = (3-1)*2+3-6 = 1 -> 0 before, 1 after
    # output (0,0) sees input rows -1..1, cols 0..2: the delta at (0,0) meets kernel tap (1,0) = 3
    assert float(y[0, 0, 0, 0]) == 3.0 and float(y.abs().sum()) == 3.0


def test_layers_that_reach_nothing_have_no_variables():
    """The reference's configuration checks (modules.py:202-222) accept blocks in which a layer's output reaches neither a later layer nor the
    output; its Keras functional model then does not contain that layer.  The oracle's plan and the model's (same rule, tests/test_modules_gpu.py
    runs both) create no variables for it."""
    from oracle import modules_oracle as M
    from seld_amd import modules
    base = {'filters0': 8, 'filters1': 8, 'filters2': 8, 'kernel_size0': 3, 'kernel_size1': 1, 'kernel_size2': 3, 'connect0': [1], 'connect1': [1, 1],
            'connect2': [1, 1, 1], 'strides': [1, 1], 'activation': 'relu'}
    assert M.dead_layers(base) == (False, False) == modules.dead_layers(base)
    d1 = dict(base, filters2=0, kernel_size2=0, connect2=[1, 1, 0])
    assert M.dead_layers(d1) == (False, True) == modules.dead_layers(d1)
    tr, nt, out = M.mother_block_plan(d1, (10, 12, 5), "mb0")
    assert not any(".c1." in n or ".bn1." in n or ".p1_" in n for n, _ in tr) and any(".c0." in n for n, _ in tr) and out == (10, 12, 13)
    d0 = dict(base, filters1=0, kernel_size1=0, connect1=[1, 0], connect2=[1, 0, 0])
    assert M.dead_layers(d0) == (True, False) == modules.dead_layers(d0)
    tr, nt, out = M.mother_block_plan(d0, (10, 12, 8), "mb0")
    assert [n.split(".")[1] for n, _ in tr] == ["c2", "c2", "bn2", "bn2"] and out == (10, 12, 8)
    both = dict(base, filters2=0, kernel_size2=0, connect2=[1, 0, 0])
    assert M.dead_layers(both) == (True, True) and M.mother_block_plan(both, (10, 12, 5), "mb0")[0] == []
    for c in (base, d1, d0, both):
        modules.check_mother_config(c)      # accepted, as the reference accepts them
