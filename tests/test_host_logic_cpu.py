"""Host-side mirrors of the reference's flag surface and data boundary (no GPU needed)."""
import json
import os

import numpy as np
import pytest

from oracle import labels_oracle as L


def test_get_param_defaults_match_reference(tmp_path, seldnet_config):
    """params.py:10-51 flag names and defaults; params.py:53-66 model_config resolution and run name."""
    from seld_amd import params
    d = tmp_path / "model_config"
    d.mkdir()
    (d / "seldnet.json").write_text(json.dumps(seldnet_config))
    config, mc = params.get_param(["--name", "x"], model_config_dir=str(d))
    assert (config.lr, config.decay, config.batch, config.agc, config.epoch) == (0.001, 0.5, 256, False, 1000)
    assert (config.loss_weight, config.lr_patience, config.patience, config.loop_time) == ("1,1000", 80, 100, 5)
    assert (config.doa_loss, config.model, config.sed_loss, config.lad_doa_thresh) == ("MSE", "seldnet", "BCE", 20)
    assert config.name == "seldnet_seldnet_MSE_x_v_0" and mc["FIRST"] == "simple_conv_block"
    with pytest.raises(ValueError):
        params.get_param(["--name", "x", "--model", "nope"], model_config_dir=str(d))
    for dl in ("MAE", "MSLE", "MMSE"):        # every --doa_loss choice of params.py:16-17 has a kernel
        assert params.get_param(["--name", "x", "--doa_loss", dl], model_config_dir=str(d))[0].doa_loss == dl
    with pytest.raises(ValueError):           # FOCAL: the reference builds a non-callable Focal_Loss (train.py:314-315, losses.py:38-48)
        params.get_param(["--name", "x", "--sed_loss", "FOCAL"], model_config_dir=str(d))
    assert params.get_param(["--name", "x", "--agc", "False"], model_config_dir=str(d))[0].agc is True   # type=bool quirk
    # augmentation flags: masks and foa swapping run on the device; time-domain mixing is refused loudly
    cfg = params.get_param(["--name", "x", "--use_tfm", "--use_acs"], model_config_dir=str(d))[0]
    assert cfg.use_tfm and cfg.use_acs and (cfg.time_mask_size, cfg.freq_mask_size) == (24, 16)
    with pytest.raises(ValueError):
        params.get_param(["--name", "x", "--use_tdm"], model_config_dir=str(d))


def _write_dataset(root, n_per_fold=1):
    feat, lab = root / "foa_dev_norm", root / "foa_dev_label"
    feat.mkdir(parents=True), lab.mkdir(parents=True)
    rng = np.random.default_rng(0)
    for fold in range(1, 7):
        for k in range(n_per_fold):
            name = f"fold{fold}_room1_mix{k:03d}.npy"           # 5th character = fold digit (data_loader.py:72-80)
            np.save(feat / name, rng.standard_normal((3000, 64, 7)).astype(np.float32))
            y = np.zeros((600, 48), np.float32)
            y[:, fold] = 1.0
            np.save(lab / name, y)
    return str(feat), str(lab)


def test_load_and_window(tmp_path):
    from seld_amd import data_loader as dl
    feat, lab = _write_dataset(tmp_path, 2)
    x, y = dl.load_seldnet_data(feat, lab, mode="train")
    assert len(x) == 8 and x[0].shape == (3000, 64, 7) and y[0].shape == (600, 48)
    assert len(dl.load_seldnet_data(feat, lab, mode="val")[0]) == 2
    with pytest.raises(ValueError):
        dl.load_seldnet_data(str(tmp_path / "missing"), lab)
    ds = dl.seldnet_data_to_dataloader(x, y, train=True, batch_size=32, loop_time=2, seed=0)
    batches = list(ds)
    assert len(batches) == len(ds) == 5                       # 80 windows * 2 loops / 32
    xb, (sed, doa) = batches[0]
    assert xb.shape == (32, 300, 64, 7) and sed.shape == (32, 60, 12) and doa.shape == (32, 60, 36)
    assert sum(b[0].shape[0] for b in batches) == 160
    # windowing agrees with the oracle restatement of data_loader.py:132-156
    fw, lw = L.window(x, y)
    ds1 = dl.seldnet_data_to_dataloader(x, y, train=False)
    xb, (sed, doa) = next(iter(ds1))
    assert xb.shape == (10, 300, 64, 7)                        # eval: one file per batch
    np.testing.assert_array_equal(xb, fw[:10])
    np.testing.assert_array_equal(np.concatenate([sed, doa], -1), lw[:10])


def test_dataset_with_device_transforms_keeps_labels_unsplit(tmp_path):
    """With augmentations attached the loader yields the total label tensor [b,60,4C] (they act on it before the
    sed/doa split, train.py:162-165); host-side tf.data style transforms are refused."""
    from seld_amd import data_loader as dl
    feat, lab = _write_dataset(tmp_path, 1)
    x, y = dl.load_seldnet_data(feat, lab, mode="train")
    ds = dl.seldnet_data_to_dataloader(x, y, train=True, batch_size=8, loop_time=1, seed=0, device_transforms=[lambda a, b, r: (a, b)])
    xb, yb = next(iter(ds))
    assert xb.shape == (8, 300, 64, 7) and yb.shape == (8, 60, 48) and len(ds.device_transforms) == 1
    with pytest.raises(ValueError):
        dl.seldnet_data_to_dataloader(x, y, sample_transforms=[lambda a, b: (a, b)])


def test_synthetic_batch_matches_the_oracles_generator():
    """bench.py draws its batch from seld_amd.synthetic (the product side may not import oracle/); the oracle keeps its own copy
    for the tests: both must produce the same arrays from a seed."""
    from oracle.seldnet_oracle import synthetic_batch as oracle_batch
    from seld_amd.synthetic import synthetic_batch
    for got, want in zip(synthetic_batch(2, 50, seed=7), oracle_batch(2, 50, seed=7)):
        assert got.dtype == want.dtype and np.array_equal(got, want)


def test_first_and_second_block_dropout_rates_reach_the_arch(seldnet_config):
    """FIRST_ARGS / SECOND_ARGS `dropout_rate` (model_config/seldnet.json:7,13; modules.py:306, 312-314) -> seld_arch.conv_dropout / gru_dropout;
    out-of-range rates and a Dropout the other FIRST blocks' specs do not have are refused."""
    import copy
    from seld_amd import models
    a = models._arch_from_config(seldnet_config, 7, 64)
    assert (a.conv_dropout, a.gru_dropout) == (0.0, 0.0)
    cfg = copy.deepcopy(seldnet_config)
    cfg["FIRST_ARGS"]["dropout_rate"], cfg["SECOND_ARGS"]["dropout_rate"] = 0.25, 0.5
    a = models._arch_from_config(cfg, 7, 64)
    assert (a.conv_dropout, a.gru_dropout) == (0.25, 0.5)
    for key, val in (("FIRST_ARGS", 1.0), ("FIRST_ARGS", -0.5), ("SECOND_ARGS", 1.0), ("SECOND_ARGS", -0.1)):
        bad = copy.deepcopy(seldnet_config)
        bad[key]["dropout_rate"] = val
        with pytest.raises(ValueError):
            models._arch_from_config(bad, 7, 64)
    xc = copy.deepcopy(seldnet_config)
    xc["FIRST"], xc["FIRST_ARGS"] = "xception_block", {"filters": 32, "block_num": 8, "dropout_rate": 0.1}
    with pytest.raises(ValueError, match="no Dropout"):
        models._arch_from_config(xc, 7, 64)
    xc["FIRST_ARGS"]["dropout_rate"] = 0.0
    xc["SECOND_ARGS"]["dropout_rate"] = 0.2
    assert abs(models._arch_from_config(xc, 7, 64).gru_dropout - 0.2) < 1e-7


def test_head_args_the_kernels_do_not_implement_are_rejected(seldnet_config):
    """simple_dense_block honours dense_activation / kernel_size / dropout_rate (modules.py:350-376): the HIP heads take
    dense_activation None / linear / relu / tanh / sigmoid (config_sampler.py:216-218 samples None and relu), kernel_size 1 .. 15 and
    dropout_rate in [0, 1); any other value must raise instead of silently training a different network."""
    import copy
    from seld_amd import models
    a = models._arch_from_config(seldnet_config, 7, 64)
    assert (a.n_sed_dense, a.n_doa_dense, a.n_classes) == (1, 1, 12)
    assert (a.sed_kernel_size, a.doa_kernel_size, a.sed_dropout, a.doa_dropout, a.output_coupling) == (1, 1, 0.0, 0.0, 0)
    ok = copy.deepcopy(seldnet_config)
    ok["SED_ARGS"].update(dense_activation="linear", kernel_size=1, dropout_rate=0, kernel_regularizer={"l1": 0.0, "l2": 1e-3})
    models._arch_from_config(ok, 7, 64)      # the regulariser only feeds model.losses, which train.trainstep never adds
    for head in ("SED_ARGS", "DOA_ARGS"):
        good = copy.deepcopy(seldnet_config)
        good[head].update(dense_activation="relu", kernel_size=3, dropout_rate=0.25)
        ar = models._arch_from_config(good, 7, 64, output_coupling=True)
        assert (ar.sed_dense_act, ar.doa_dense_act) == ((3, 0) if head == "SED_ARGS" else (0, 3))
        assert (ar.sed_kernel_size, ar.doa_kernel_size) == ((3, 1) if head == "SED_ARGS" else (1, 3))
        assert (ar.sed_dropout, ar.doa_dropout) == ((0.25, 0.0) if head == "SED_ARGS" else (0.0, 0.25)) and ar.output_coupling == 1
        for key, val in (("dense_activation", "swish"), ("kernel_size", 0), ("kernel_size", 16), ("dropout_rate", 1.0), ("dropout_rate", -0.1)):
            bad = copy.deepcopy(seldnet_config)
            bad[head][key] = val
            with pytest.raises(ValueError):
                models._arch_from_config(bad, 7, 64)


def test_stage_wrappers_and_identity_heads_map_to_the_blocks_they_build(seldnet_config):
    """bidirectional_GRU_stage (modules.py:46-61), simple_dense_stage (modules.py:86-103: `activation` -> dense_activation) and
    identity_block (modules.py:639-642) as SECOND / SED / DOA: the arch the wrappers' blocks have."""
    import copy
    from seld_amd import models
    cfg = copy.deepcopy(seldnet_config)
    cfg["SECOND"], cfg["SECOND_ARGS"] = "bidirectional_GRU_stage", {"depth": 2, "units": 128, "dropout_rate": 0.0}
    cfg["SED"], cfg["SED_ARGS"] = "simple_dense_stage", {"depth": 2, "units": 64, "activation": "relu"}
    cfg["DOA"], cfg["DOA_ARGS"] = "identity_block", {}
    a = models._arch_from_config(cfg, 7, 64)
    assert (a.n_gru, list(a.gru_units)[:2]) == (2, [128, 128])
    assert (a.n_sed_dense, list(a.sed_units)[:2], a.sed_dense_act) == (2, [64, 64], 3)
    assert (a.n_doa_dense, a.doa_dense_act) == (0, 0)
    assert cfg["SED"] == "simple_dense_stage"                      # the caller's dict is left alone
    bad = copy.deepcopy(cfg)
    bad["SECOND"] = "RNN_stage"
    with pytest.raises(ValueError):
        models._arch_from_config(bad, 7, 64)


def test_keras_h5_name_mapping_on_a_hand_built_name_list():
    """tools/keras_h5_to_npz.py (reference seams train.py:372-380 / :322-331 / evaluator.py:57): the Keras-variable -> seld_amd
    mapping is a pure function of names and shapes, tested here on the names Keras gives model_config/seldnet.json's layers
    (h5py, which the converter's file IO needs, is absent from this image)."""
    import importlib.util
    from conftest import ROOT
    spec_ = importlib.util.spec_from_file_location("keras_h5_to_npz", os.path.join(ROOT, "tools", "keras_h5_to_npz.py"))
    K = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(K)
    kv = []
    # a checkpoint taken in a session where earlier models had been built: suffixes do not start at 0 and the file order is alphabetical
    for i, cin in zip((4, 5, 6), (7, 64, 64)):
        kv += [(f"conv2d_{i}/kernel:0", (3, 3, cin, 64)), (f"conv2d_{i}/bias:0", (64,))]
        kv += [(f"batch_normalization_{i}/{leaf}:0", (64,)) for leaf in ("gamma", "beta", "moving_mean", "moving_variance")]
    for b, (fw, bw) in (("bidirectional_2", ("gru_cell_7", "gru_cell_8")), ("bidirectional_3", ("gru_cell_10", "gru_cell_11"))):
        for tag, cell in (("forward_gru_" + b[-1], fw), ("backward_gru_" + b[-1], bw)):
            kv += [(f"{b}/{tag}/{cell}/kernel:0", (128, 384)), (f"{b}/{tag}/{cell}/recurrent_kernel:0", (128, 384)),
                   (f"{b}/{tag}/{cell}/bias:0", (2, 384))]
    kv += [("conv1d_2/kernel:0", (1, 128, 128)), ("conv1d_2/bias:0", (128,)), ("conv1d_3/kernel:0", (1, 128, 128)), ("conv1d_3/bias:0", (128,))]
    kv += [("sed_out/kernel:0", (128, 12)), ("sed_out/bias:0", (12,)), ("doa_out/kernel:0", (128, 36)), ("doa_out/bias:0", (36,))]
    kv = sorted(kv)
    m = K.map_keras_variables(kv)
    ours = K.our_variable_shapes()
    K.check_shapes(m, kv, ours)
    # the converter's variable list is the oracle's (and so the C library's) variable list
    from oracle import seldnet_oracle as O
    from __graft_entry__ import SELDNET_CONFIG
    tr, nt = O.variable_specs(O.Spec.from_config(SELDNET_CONFIG))
    assert {n: tuple(s) for n, s in tr + nt} == ours and sum(int(np.prod(s)) for _, s in tr) == 513840
    assert m["conv0.kernel"] == "conv2d_4/kernel:0" and m["bn2.moving_variance"] == "batch_normalization_6/moving_variance:0"
    assert m["gru1.bwd.recurrent_kernel"] == "bidirectional_3/backward_gru_3/gru_cell_11/recurrent_kernel:0"
    assert m["sed.dense0.kernel"] == "conv1d_2/kernel:0" and m["doa.dense0.bias"] == "conv1d_3/bias:0"
    assert m["doa.out.kernel"] == "doa_out/kernel:0" and len(m) == len(ours)
    with pytest.raises(ValueError):                       # a layer missing
        K.map_keras_variables([v for v in kv if not v[0].startswith("conv2d_6")])
    bad = [(n, (3, 3, 10, 64) if n == "conv2d_4/kernel:0" else s) for n, s in kv]
    with pytest.raises(ValueError):                       # a mic-feature checkpoint into a foa model
        K.check_shapes(K.map_keras_variables(bad), bad, ours)


def test_seldnet_v1_json_is_the_same_network(seldnet_config):
    """model_config/seldnet_v1.json differs from seldnet.json only in DOA_ARGS lacking 'activation' — a key nothing reads:
    simple_dense_block takes 'dense_activation' (modules.py:356) and models.seldnet hard-codes sigmoid / tanh on the output layers
    (models.py:27-30).  Both JSONs must map to the same architecture."""
    import copy
    import ctypes as C
    from seld_amd import models
    v1 = copy.deepcopy(seldnet_config)
    del v1["DOA_ARGS"]["activation"]
    a, b = models._arch_from_config(seldnet_config, 7, 64), models._arch_from_config(v1, 7, 64)
    assert bytes(C.string_at(C.addressof(a), C.sizeof(a))) == bytes(C.string_at(C.addressof(b), C.sizeof(b)))


def test_bench_last_line_is_a_compact_record():
    """The driver parses bench.py's LAST stdout line out of an 8 KB tail (round 3's single 28 KB line left `parsed: null`).  The
    formatter, run on a canned FULL record of that size (round 3's own line), must give a line under 3 KB that round-trips through
    json and keeps the contract's fields, the dominant kernel's roofline and cpu_baseline."""
    import json
    import bench
    from conftest import ROOT
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_f_bench_11475clips.json")))
    assert len(json.dumps(full)) > 20000
    full["config"]["dp_backend"] = "none"
    line = bench.compact_record(full, "gpurun_out/bench_detail.json")
    assert len(line) < 3072 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert d[k] == full[k], k
    assert d["config"]["workload"] == full["config"]["workload"] and d["config"]["dp_backend"] == "none"
    assert d["roofline"]["kernel"] == full["roofline"]["kernel"] and d["roofline"]["frac"] == full["roofline"]["frac"]
    assert set(("bound", "achieved", "peak", "unit", "traffic")) <= set(d["roofline"])
    assert d["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and d["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["sample"]
    assert d["xception_gru_clips_s"] == full["configs"]["xception_gru"]["value"]
    assert d["resnet50_gru_clips_s"] == full["configs"]["resnet50_gru"]["value"]
    assert d["seldnet_bf16_clips_s"] == full["configs"]["seldnet_bf16"]["value"]
    # a pathological record (very long free-text fields) still fits: optional fields go first
    full["config"]["workload"] = "w" * 2500
    full["cpu_baseline"]["sample"] = "s" * 2500
    assert len(bench.compact_record(full, "x" * 200)) < 3072


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` with no outer launcher: bench.py becomes the launcher — N fresh rank processes with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set on 127.0.0.1, rank 0's stdout last on the launcher's stdout, a failing rank's exit
    code propagated and the lingering ranks stopped.  SELD_BENCH_LAUNCH_PROBE makes every rank report what it was started with and leave
    before anything touches a GPU (the full two-rank run on a GPU is tests/test_dp_gpu.py::test_bench_two_ranks_rehearsal[self-...])."""
    import json
    import subprocess
    import sys
    import time
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, env=dict(env, SELD_BENCH_LAUNCH_PROBE="ok"), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["rank"] == 0 and out[0]["world"] == 2 and out[0]["local"] == 0, "only rank 0 writes to the launcher's stdout"
    assert out[0]["master"].startswith("127.0.0.1:") and out[0]["argv"] == cmd[2:]
    other = [json.loads(l.split("] ", 1)[1]) for l in r.stderr.splitlines() if l.startswith("[rank 1] {")]
    assert len(other) == 1 and other[0]["rank"] == 1 and other[0]["local"] == 1 and other[0]["master"] == out[0]["master"]
    # under an outer launcher (WORLD_SIZE set) nothing is spawned: the process IS the rank
    r = subprocess.run(cmd, env=dict(env, SELD_BENCH_LAUNCH_PROBE="ok", WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["rank"] == 1
    # a failing rank: its code comes back, and the rank that would have lingered for a minute is stopped
    t0 = time.time()
    r = subprocess.run(cmd, env=dict(env, SELD_BENCH_LAUNCH_PROBE="fail1"), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 3 and time.time() - t0 < 45, (r.returncode, time.time() - t0)


def test_stored_traffic_figures_are_dropped_when_their_kernel_source_changes(tmp_path):
    """bench.py's `roofline.traffic` comes from a STORED counter pass (profiles/traffic.json).  A figure survives only while the source
    of its kernel (and common.h) hashes to what tools/pmc_traffic.py recorded beside it."""
    import hashlib
    import json
    import bench
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    for f, body in (("gru.hip", b"gru v1"), ("conv_sb.hip", b"conv v1"), ("common.h", b"common")):
        (csrc / f).write_bytes(body)
    h = lambda b: hashlib.sha256(b).hexdigest()[:16]
    tab = {"_provenance": "test", "_source_hashes": {"gru.hip": h(b"gru v1"), "conv_sb.hip": h(b"conv v1"), "common.h": h(b"common")},
           "gru_fwd": {"hbm_bytes_per_launch": 1}, "conv64_fwd_dgrad_W4": {"hbm_bytes_per_launch": 2}, "unknown_group": {"hbm_bytes_per_launch": 3}}
    p = tmp_path / "traffic.json"
    p.write_text(json.dumps(tab))
    got = bench.load_traffic(str(p), str(csrc))
    assert got["gru_fwd"]["hbm_bytes_per_launch"] == 1 and "conv64_fwd_dgrad_W4" in got and got["_dropped"] == ["unknown_group"]
    (csrc / "gru.hip").write_bytes(b"gru v2")
    got = bench.load_traffic(str(p), str(csrc))
    assert "gru_fwd" not in got and "conv64_fwd_dgrad_W4" in got and "gru_fwd" in got["_dropped"]
    (csrc / "common.h").write_bytes(b"common v2")
    assert [k for k in bench.load_traffic(str(p), str(csrc)) if not k.startswith("_")] == []
    del tab["_source_hashes"]
    p.write_text(json.dumps(tab))
    assert [k for k in bench.load_traffic(str(p), str(csrc)) if not k.startswith("_")] == []
    # the committed table is consistent with bench.py's group -> source map
    real = json.load(open(os.path.join(bench.ROOT, "profiles", "traffic.json")))
    assert all(k in bench.TRAFFIC_SOURCE for k in real if not k.startswith("_"))
