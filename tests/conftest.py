import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def seld_lib():
    """libseld_hip.so, built in-tree if absent (hipcc cross-compiles gfx950 without a GPU)."""
    from seld_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


@pytest.fixture(scope="session")
def seldnet_config():
    """model_config/seldnet.json of the reference, restated as data (the reference tree does not
    travel to the GPU box); n_classes = 12 as train.py:306-307 forces."""
    return {
        "FIRST": "simple_conv_block",
        "FIRST_ARGS": {"filters": [64, 64, 64], "pool_size": [[5, 4], [1, 4], [1, 2]], "dropout_rate": 0.0},
        "SECOND": "bidirectional_GRU_block",
        "SECOND_ARGS": {"units": [128, 128], "dropout_rate": 0.0},
        "SED": "simple_dense_block",
        "SED_ARGS": {"units": [128], "n_classes": 14, "activation": "sigmoid", "name": "sed_out"},
        "DOA": "simple_dense_block",
        "DOA_ARGS": {"units": [128], "n_classes": 42, "activation": "tanh", "name": "doa_out"},
        "n_classes": 12,
    }


@pytest.fixture(scope="session")
def xception_config(seldnet_config):
    """model_config/xception_gru.json of the reference, restated as data (n_classes = 12 as train.py:306-307 forces)."""
    import copy
    cfg = copy.deepcopy(seldnet_config)
    cfg["FIRST"] = "xception_block"
    cfg["FIRST_ARGS"] = {"filters": 32, "block_num": 8, "kernel_regularizer": {"l1": 0, "l2": 1e-3}}
    return cfg


@pytest.fixture(scope="session")
def resnet50_config(seldnet_config):
    """model_config/resnet50_gru.json of the reference, restated as data (n_classes = 12 as train.py:306-307 forces)."""
    import copy
    cfg = copy.deepcopy(seldnet_config)
    cfg["FIRST"] = "resnet50_block"
    cfg["FIRST_ARGS"] = {"filters": 32, "block_num": [3, 4, 6, 3], "kernel_regularizer": {"l1": 0, "l2": 1e-3}}
    return cfg
