"""CPU-side checks of the C-ABI boundary: the library builds, loads without a GPU, and exports
exactly the symbols include/seld_hip.h declares; the ctypes table covers all of them."""
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "seld_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(seld_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(seld_lib):
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(seld_lib, n), f"libseld_hip.so does not export {n}"


def test_ctypes_table_matches_header():
    from seld_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_create_fails_loudly_without_gpu_or_bad_args(seld_lib):
    import ctypes as C
    import torch
    from seld_amd import _lib
    a = _lib.Arch()
    ctx = C.c_void_p()
    rc = seld_lib.seld_create(C.byref(a), 2, 50, 0, 0, C.byref(ctx))
    assert rc != 0 and not ctx.value
    assert seld_lib.seld_last_error(None)
    if not torch.cuda.is_available():
        from seld_amd import models
        import pytest
        with pytest.raises(_lib.SeldLibraryError):
            models.seldnet((2, 50, 64, 7), {"FIRST": "simple_conv_block", "FIRST_ARGS": {"filters": [64, 64, 64], "pool_size": [[5, 4], [1, 4], [1, 2]]},
                                            "SECOND": "bidirectional_GRU_block", "SECOND_ARGS": {"units": [128, 128]},
                                            "SED": "simple_dense_block", "SED_ARGS": {"units": [128]},
                                            "DOA": "simple_dense_block", "DOA_ARGS": {"units": [128]}})


def test_unsupported_blocks_raise_value_error():
    import pytest
    from seld_amd import models
    with pytest.raises(ValueError):
        models._arch_from_config({"FIRST": "mother_block", "SECOND": "bidirectional_GRU_block", "SED": "simple_dense_block",
                                  "DOA": "simple_dense_block"}, 7, 64)
    with pytest.raises(ValueError):      # xception_block kernels exist for the JSON's width only (spec/XCEPTION_BLOCK.md)
        models._arch_from_config({"FIRST": "xception_block", "FIRST_ARGS": {"filters": 48, "block_num": 8},
                                  "SECOND": "bidirectional_GRU_block", "SECOND_ARGS": {"units": [128, 128]}, "SED": "simple_dense_block",
                                  "SED_ARGS": {"units": [128]}, "DOA": "simple_dense_block", "DOA_ARGS": {"units": [128]}}, 7, 64)
