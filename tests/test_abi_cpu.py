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
    with pytest.raises(ValueError):      # a fused ctx has no mother_block (models.seldnet composes one from module operators: seld_amd/modules.py)
        models._arch_from_config({"FIRST": "mother_block", "SECOND": "bidirectional_GRU_block", "SED": "simple_dense_block",
                                  "DOA": "simple_dense_block"}, 7, 64)
    with pytest.raises(ValueError):      # blocks of the reference that nothing here implements
        models._arch_from_config({"FIRST": "conformer_encoder_stage", "SECOND": "bidirectional_GRU_block", "SED": "simple_dense_block",
                                  "DOA": "simple_dense_block"}, 7, 64)
    with pytest.raises(ValueError):      # xception_block kernels exist for the JSON's width only (spec/XCEPTION_BLOCK.md)
        models._arch_from_config({"FIRST": "xception_block", "FIRST_ARGS": {"filters": 48, "block_num": 8},
                                  "SECOND": "bidirectional_GRU_block", "SECOND_ARGS": {"units": [128, 128]}, "SED": "simple_dense_block",
                                  "SED_ARGS": {"units": [128]}, "DOA": "simple_dense_block", "DOA_ARGS": {"units": [128]}}, 7, 64)


def test_documented_binding_stub_matches_header_ctypes_and_library(seld_lib):
    """INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would copy.  Its struct declarations are
    extracted from the document and EXECUTED, then compared field by field with seld_amd/_lib.py's (the binding the tests run on),
    with the field list of `seld_arch` / `seld_loss_cfg` parsed from include/seld_hip.h, and with the sizes the compiled library
    reports (seld_abi_sizes) — so header, ctypes, document and binary cannot drift apart again (round 2's stub ended at n_classes)."""
    import ctypes as C
    from seld_amd import _lib
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", doc, flags=re.S)
    stub = next(b for b in blocks if "class Arch(C.Structure)" in b)
    classes = re.findall(r"(class (?:Arch|LossCfg)\(C\.Structure\):.*?\n)(?=\n|class |def )", stub, flags=re.S)
    assert len(classes) == 2
    ns = {"C": C}
    exec("".join(classes), ns)

    def fields(cls):
        return [(n, C.sizeof(t), getattr(t, "_length_", 1)) for n, t in cls._fields_]

    assert fields(ns["Arch"]) == fields(_lib.Arch)
    assert fields(ns["LossCfg"]) == fields(_lib.LossCfg)
    # the header's struct bodies: name and array length of every member, in order
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "seld_hip.h")).read(), flags=re.S)
    for cname, cls in (("seld_arch", _lib.Arch), ("seld_loss_cfg", _lib.LossCfg)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        members = []
        for _, decls in re.findall(r"\b(int32_t|float)\s+([^;]+);", body):          # `float w_sed, w_doa;` declares two
            for d in decls.split(","):
                m = re.fullmatch(r"\s*(\w+)(?:\[(\w+)\])?\s*", d)
                members.append((m.group(1), m.group(2)))
        consts = {"SELD_MAX_LAYERS": 4}
        want = [(n, 4 * int(consts.get(k, k or 1)), int(consts.get(k, k or 1))) for n, k in members]
        assert want == fields(cls), cname
    sizes = (C.c_int32 * 2)()
    assert seld_lib.seld_abi_sizes(sizes, 2) == 2
    assert (sizes[0], sizes[1]) == (C.sizeof(ns["Arch"]), C.sizeof(ns["LossCfg"])) == (C.sizeof(_lib.Arch), C.sizeof(_lib.LossCfg))
