#!/usr/bin/env python3
"""bench.py — train-step clips/s of the SELDnet hot path on N MI355X (one process per GPU).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N --steps K --warmup W        (no launcher: bench.py starts its own N rank processes, spawn_ranks)

A "step" is one reference train.trainstep (train.py:22-36: forward(training) + BCE/MSE losses +
gradients + Adam) on one synthetic batch of B clips of [T=3000, F=64, C=7] per GPU (weak scaling:
B per GPU is fixed).  Rank 0 prints ONE JSON line (metric of BASELINE.json) that also carries
  roofline      the dominant kernel's achieved rate, from HIP events recorded by the library on its
                own stream during the timed region, against the gfx950 peak of its bound
  cpu_baseline  the CPU oracle's train step (PyTorch-CPU restatement of the reference semantics, NOT
                the reference's TensorFlow) timed on the host cores, N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from __graft_entry__ import SELDNET_CONFIG

METRIC = "train-step clips/sec (7ch×3000×64) seldnet.json at 1/2/4/8 MI355X"
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16)
# fp32 products computed as 6 bf16 MFMA products of exactly split operands (conv_sb / conv_pool_sb / conv_wgrad_sb /
# gemm_sb): the ceiling for the fp32-equivalent FLOP count is a sixth of the bf16 peak
PEAK_SPLIT_BF16_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6
SPLIT_BF16_GROUPS = {"conv1_fwd", "conv2_fwd", "conv2_dgrad", "conv2_wgrad", "conv3_fwd", "conv3_dgrad", "conv3_wgrad",
                     "gru_inproj_gemm", "gru_bwd_gemms",
                     # resnet50_block: 87 % of the stages' FLOP (stages 2-3, the 128-column products of stages 0-1, stage 1's 3x3) run on
                     # the split-bf16 kernels, the 32- / 64-column rest on the f32 MFMA: the whole group is priced against the split peak
                     "rn_stages_fwd", "rn_stages_bwd"}   # with the default options (seld_set_option)
# backward-only products (input / kernel gradients: they feed no MaxPool / ReLU decision) run FOUR of the six products by default (option
# "bwd_four_products", conv_sb.hip): their ceiling is a quarter of the bf16 peak — the harder bar, and the one these groups are priced against
PEAK_SPLIT4_BF16_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 4
FOUR_PRODUCT_GROUPS = {"conv2_dgrad", "conv2_wgrad", "conv3_dgrad", "conv3_wgrad", "gru_bwd_gemms", "rn_stages_bwd", "rn_products_dgrad"}
PEAK_HBM_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak
# GRU recurrence (gru.hip): S dependent steps per launch on the 2B CUs that hold a (clip, direction) each.  The HARDWARE floor of a step is
# its mat-vec at the CU's fp32 rate: 128 x 384 multiply-adds / (128 FMA per clock per CU: 4 SIMD-32 x 32 lanes) = 384 cycles, forward and
# backward alike (MI355X_MICROARCH.md: 157.3 TFLOP/s = 256 CUs x 2.4 GHz x 256 FLOP/clk).  Where the measured cycles go beyond that is
# profiles/r03_gru_experiments.txt (in-kernel timeline, instruction latencies, the rejected forms).
GRU_FMA_PER_STEP = 128 * 384
CU_FMA_PER_CLK = 128
GRU_GROUPS = ("gru_fwd", "gru_bwd")


def kernel_work(name, B, T, F=64, C=7):
    """Algorithmic work per LAUNCH of a timed kernel group: (bound, amount, unit).
    FLOP = 2*MAC with MAC counts of SURVEY.md §8 (complexity.py formulas); bytes = the tensors the
    op must read+write once (fp32)."""
    S = T // 5
    px1, px2, px3 = B * T * F, B * S * 16, B * S * 4
    rows = B * S
    mac = {"conv1": px1 * 64 * 9 * C, "conv2": px2 * 64 * 576, "conv3": px3 * 64 * 576}
    table = {
        "conv1_fwd": ("mfma", 2 * mac["conv1"]),
        # first block's kernel gradient without the pre-BN tensor (conv_gram.hip): the dense product runs as the patch Gram
        # matrix on the side stream; what is timed here is the sparse gather-accumulate over x, p, dp (fp32) and amax (u8)
        "conv1_wgrad": ("hbm", 4 * (px1 * C + 2 * (px1 * 64 // 20)) + px1 * 64 // 20),
        "conv2_fwd": ("mfma", 2 * mac["conv2"]), "conv2_wgrad": ("mfma", 2 * mac["conv2"]), "conv2_dgrad": ("mfma", 2 * mac["conv2"]),
        "conv3_fwd": ("mfma", 2 * mac["conv3"]), "conv3_wgrad": ("mfma", 2 * mac["conv3"]), "conv3_dgrad": ("mfma", 2 * mac["conv3"]),
        # first block: the conv epilogue already reduced the (5,4) windows; what is left reads zext and writes p
        "pool1_fwd": ("hbm", 4 * 2 * (px1 * 64 // 20)), "pool2_fwd": ("hbm", 4 * (px2 * 64 + px2 * 64 // 4)),
        "pool3_fwd": ("hbm", 4 * (px3 * 64 + px3 * 64 // 2)),
        "pool1_bwd_reduce": ("hbm", 4 * 2 * (px1 * 64 // 20)),   # pooled-only statistics pass: reads p and dp
        "pool1_bwd_dz": ("hbm", 4 * (2 * px1 * 64 + px1 * 64 // 20)),
        # (the sums of blocks 2 / 3 read the POOLED tensors only, like block 1's: x-hat at the argmax is recovered from p — bn_pool.hip; counting z
        # here overstated the group's rate 2.5x / 1.5x and put pool2_bwd_reduce above the HBM peak on a fast box)
        "pool2_bwd_reduce": ("hbm", 4 * 2 * (px2 * 64 // 4)), "pool2_bwd_dz": ("hbm", 4 * (2 * px2 * 64 + px2 * 64 // 4)),
        "pool3_bwd_reduce": ("hbm", 4 * 2 * (px3 * 64 // 2)), "pool3_bwd_dz": ("hbm", 4 * (2 * px3 * 64 + px3 * 64 // 2)),
        # GRU recurrence (both directions of one layer): read gx [rows,384] + write h [rows,128] + saved gates [rows,512]
        "gru_fwd": ("hbm", 2 * 4 * rows * (384 + 128 + 512)),
        "gru_bwd": ("hbm", 2 * 4 * rows * (128 + 128 + 512 + 128 + 768)),
        "gru_inproj_gemm": ("mfma", 2 * 2 * rows * 128 * 384),
        "gru_bwd_gemms": ("mfma", 2 * 2 * rows * 128 * 384),   # main stream: the two input-gradient GEMMs of a layer
        # the default build computes both heads as ONE 48-column product with W1 W2 pre-multiplied (heads_fused): what is timed on the main
        # stream is rows x 128 x 48 forward and the same again for the input gradient (the weight gradients run on the side stream)
        "heads_fwd": ("mfma", 2 * rows * 128 * 48), "heads_bwd": ("mfma", 2 * rows * 128 * 48),
        "adam": ("hbm", 4 * 7 * N_PARAMS),
        # xception_block middle flow (spec/XCEPTION_BLOCK.md), per launch group on [B,S,16,64]: depthwise 3x3 = read + write the
        # tensor; pointwise 64 x 64 product; BatchNorm passes
        "xc_depthwise_fwd": ("hbm", 4 * 2 * px2 * 64), "xc_pointwise_fwd": ("mfma", 2 * px2 * 64 * 64),
        # fused unit forward (default): read the unit's input once, write the depthwise output and z
        "xc_unit_fwd": ("hbm", 4 * 3 * px2 * 64),
        # per scope (one per unit): BatchNorm finalisation (tiny) and, for a module's last unit only, the apply + residual pass (3 tensors):
        # 8 of 24 scopes move data -> a third of 3 tensors on average
        "xc_bn_fwd": ("hbm", 4 * 1 * px2 * 64),
        # backward of a unit on the main stream: BatchNorm' sums (read gY, z), then xc_pw_bwd (dz formed on load: reads gY, z, the depthwise
        # output; writes the gradient w.r.t. the depthwise output; two 64 x 64 products on the f32 MFMA), then the depthwise input gradient
        # (reads that gradient, the unit's input for the ReLU mask, the residual gradient; writes the input gradient)
        # (round 5, the default: that pass also leaves the depthwise kernel gradient's slabs and, for 16 of the 24 units, the previous BatchNormalization's
        # backward sums — a scope of xc_bn_bwd then only folds partials: 8 of 24 scopes read gY and z -> a third of 2 tensors on average; the
        # depthwise scope reads the gradient and the unit's input, writes the input gradient, 8 of 24 also read the residual gradient)
        "xc_bn_bwd": ("hbm", 4 * 2 * px2 * 64 / 3),
        "xc_pointwise_bwd": ("hbm", 4 * 4 * px2 * 64), "xc_depthwise_bwd": ("hbm", 4 * (3 + 1 / 3) * px2 * 64),
    }
    if name.startswith("rn_") and name not in ("rn_stages_fwd", "rn_stages_bwd"):
        # level-3 groups of resnet50_block, per SCOPE sums are not meaningful (one scope per launch of very different sizes): work per STEP
        mac, bn_f, bn_b, cin, wbins = 0, 0, 0, 64, 16
        for s_, nb in enumerate(RESNET_BLOCKS):
            w = 32 * 2 ** s_
            for b in range(nb):
                if b == 0 and s_ > 0:
                    wbins //= 2
                px = B * S * wbins
                proj = b == 0
                mac += px * (cin * w + 9 * w * w + w * 4 * w + (cin * 4 * w if proj else 0))
                chans = [w, w, 4 * w] + ([4 * w] if proj else [])
                # forward per BatchNormalization: statistics read z, apply reads z and writes y; the block output also reads the shortcut
                bn_f += sum(4 * px * ch * 3 for ch in chans) + 4 * px * 4 * w
                # backward per BatchNormalization: sums read dy and z, dz reads dy and z and writes dz
                bn_b += sum(4 * px * ch * 5 for ch in chans)
                cin = 4 * w
        per_step = {"rn_products_fwd": ("mfma", 2 * mac), "rn_products_dgrad": ("mfma", 2 * mac), "rn_bn_fwd": ("hbm", bn_f),
                    "rn_bn_bwd": ("hbm", bn_b)}
        return per_step.get(name)
    if name in ("rn_stages_fwd", "rn_stages_bwd"):
        # resnet50_block (spec/RESNET50_BLOCK.md): the products of every bottleneck (1x1 reduce, 3x3, 1x1 expand, projection shortcut);
        # the backward pass runs each twice (kernel gradient + input gradient).  The group's time also holds the BatchNorm passes,
        # im2col and (backward) whatever the side stream's kernel gradients make the main stream wait for
        mac, cin, wbins = 0, 64, 16
        for s_, nb in enumerate(RESNET_BLOCKS):
            w = 32 * 2 ** s_
            for b in range(nb):
                if b == 0 and s_ > 0:
                    wbins //= 2
                px = B * S * wbins
                mac += px * (cin * w + 9 * w * w + w * 4 * w + (cin * 4 * w if b == 0 else 0))
                cin = 4 * w
        return ("mfma", (2 if name == "rn_stages_fwd" else 4) * mac)
    return table.get(name)


RESNET_BLOCKS = [3, 4, 6, 3]      # model_config/resnet50_gru.json:5
N_PARAMS = 513840                 # seldnet.json; main() sets the timed model's count


# profiles/traffic.json group -> the kernel source it was measured on (tools/pmc_traffic.py stores every source's hash beside the figures)
TRAFFIC_SOURCE = {"conv1_fwd": "conv_pool_sb.hip", "conv1_fwd_f32": "conv_pool.hip", "conv1_wgrad": "conv_gram.hip", "conv1_gram": "conv_gram.hip",
                  "conv1_wgrad_fused": "conv.hip", "gru_fwd": "gru.hip", "gru_bwd": "gru.hip", "conv64_fwd_dgrad_W16": "conv_sb.hip",
                  "conv64_fwd_dgrad_W4": "conv_sb.hip", "conv64_fwd_dgrad_W16_sbr": "conv_sb.hip", "conv64_fwd_dgrad_W4_sbr": "conv_sb.hip",
                  "conv2_wgrad": "conv_wgrad_sb.hip", "conv3_wgrad": "conv_wgrad_sb.hip", "conv2_wgrad_f32": "conv.hip", "conv3_wgrad_f32": "conv.hip",
                  "pool1_fwd": "conv_pool.hip", "gemm": "gemm.hip", "gemm_tn": "gemm.hip", "gemm_sb_4wave": "gemm_sb.hip", "gemm_sb_16wave": "gemm_sb.hip",
                  "feat_dft": "features.hip", "feat_frame": "features.hip", "feat_frame_workgroup": "features.hip", "feat_topdb": "features.hip"}


def load_traffic(path=None, csrc=None):
    """profiles/traffic.json (a STORED rocprofv3 --pmc pass: FETCH_SIZE / WRITE_SIZE cannot be collected inside a timed run) with every
    figure whose kernel source (or common.h) has changed since that pass REMOVED: a stale byte count is worse than `traffic: null`.
    A table without `_source_hashes` (collected before the hashes existed) cannot be checked and is dropped whole."""
    import hashlib
    path = path or os.path.join(ROOT, "profiles", "traffic.json")
    csrc = csrc or os.path.join(ROOT, "seld_amd", "csrc")
    if not os.path.exists(path):
        return {}
    tab = json.load(open(path))
    stored = tab.get("_source_hashes")
    if not isinstance(stored, dict):
        return {"_dropped": sorted(k for k in tab if not k.startswith("_")), "_provenance": str(tab.get("_provenance", "")) + " (no source hashes: dropped)"}

    def cur(f):
        fp = os.path.join(csrc, f)
        return hashlib.sha256(open(fp, "rb").read()).hexdigest()[:16] if os.path.exists(fp) else None

    common_ok = cur("common.h") == stored.get("common.h")
    out, dropped = {}, []
    for k, v in tab.items():
        if k.startswith("_"):
            out[k] = v
            continue
        src = TRAFFIC_SOURCE.get(k)
        if src and common_ok and cur(src) == stored.get(src):
            out[k] = v
        else:
            dropped.append(k)
    out["_dropped"] = dropped
    return out


def host_cores():
    """CPU threads this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(B_gpu, T, steps, warmup, budget_s, model_config=None):
    """the oracle's train step (PyTorch-CPU fp32 restatement of the reference semantics, NOT the
    reference's TensorFlow) on the host cores: `warmup` warm-ups, median of `steps` steps, at B=2 (BASELINE configs[0],
    the reference's own CPU-runnable case) and at the GPU batch.  The GPU-batch leg times fewer steps (>= 3) when `steps`
    of them would exceed `budget_s`, so that the default bench still finishes in minutes; what was run is in `sample`."""
    from oracle import seldnet_oracle as O
    spec = O.Spec.from_config(model_config or SELDNET_CONFIG)
    w, st = O.random_weights(spec, 0)
    cores = host_cores()
    torch.set_num_threads(cores)

    def leg(B, n_warm, n_steps, budget):
        x, ys, yd = O.synthetic_batch(B, T)
        t0 = time.perf_counter()
        O.train_step(spec, w, st, x, ys, yd)
        first = time.perf_counter() - t0
        if budget is not None:
            n_steps = max(3, min(n_steps, int(budget / max(first, 1e-3)) - n_warm))
            n_warm = max(1, min(n_warm, int(0.25 * budget / max(first, 1e-3))))
        for _ in range(n_warm - 1):
            O.train_step(spec, w, st, x, ys, yd)
        ts = []
        for _ in range(n_steps):
            t0 = time.perf_counter()
            O.train_step(spec, w, st, x, ys, yd)
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        return {"value": round(B / med, 3), "unit": "clips/s", "cores": cores, "kind": "port",
                "sample": f"oracle train_step, B={B} clips of [T={T},64,7], median of {n_steps} steps after {n_warm} warm-ups "
                          f"({med:.2f} s/step, {cores} threads); PyTorch-CPU restatement of the reference semantics, not the reference's TF"}

    small = leg(2, warmup, steps, None)
    big = leg(B_gpu, warmup, steps, budget_s) if B_gpu != 2 else small
    big["b2"] = small         # configs[0]: the plumbing case
    return big


def box_peaks(lib, dev_index):
    """peaks read on the box, printed next to the constants the roofline fractions use.  Derived from
    hipGetDeviceProperties (CU count, engine / memory clocks, bus width): fp32 MFMA = CUs x 4 SIMDs x 64 FLOP/clk x clock; dense
    bf16 MFMA = 16 x that; HBM = 2 x memory clock x bus width / 8."""
    import ctypes as C
    p = torch.cuda.get_device_properties(dev_index)
    out = {"name": p.name, "gcn_arch": getattr(p, "gcnArchName", None), "total_memory_GB": round(p.total_memory / 1e9, 1)}
    cu, clk, mclk, bus = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    if lib.seld_device_clocks(dev_index, C.byref(cu), C.byref(clk), C.byref(mclk), C.byref(bus)) == 0:
        out.update({"compute_units": cu.value, "engine_clock_MHz": clk.value / 1e3, "memory_clock_MHz": mclk.value / 1e3,
                    "memory_bus_bits": bus.value,
                    "derived_f32_mfma_TFLOPS": round(cu.value * 4 * 64 * clk.value * 1e3 / 1e12, 1),
                    "derived_bf16_mfma_TFLOPS": round(16 * cu.value * 4 * 64 * clk.value * 1e3 / 1e12, 1),
                    "derived_hbm_GBps": round(2 * mclk.value * 1e3 * bus.value / 8 / 1e9, 1)})
    out["constants_used"] = {"f32_mfma_TFLOPS": PEAK_F32_MFMA_TFLOPS, "bf16_mfma_TFLOPS": PEAK_BF16_MFMA_TFLOPS,
                             "split_bf16_TFLOPS": round(PEAK_SPLIT_BF16_TFLOPS, 1), "hbm_GBps": PEAK_HBM_GBPS,
                             "source": "/opt/skills/guides/MI355X_MICROARCH.md"}
    return out


def features_leg(dev, clips=8, reps=6):
    """The on-device feature stage (feature_extractor.extract_features, feature_extractor.py:53-88) on 60-s FOA clips
    [4, 1 440 000] -> [3001, 64, 7]: clips/s and the achieved ALGORITHMIC HBM rate (23 040 000 B of wav in +
    5 376 000 B of features out = 28 416 000 B per clip) of the extraction launches, timed with HIP events on their stream.  The
    clips go through `FeatureExtractor.batch` (one pair of launches for the batch of `clips` resident clips, each clip clamped by its
    own maximum) — what a loader preprocessing a list of files does; the clip-by-clip rate is reported beside it."""
    from seld_amd import feature_extractor as FE
    n = 1440000
    rng = np.random.default_rng(0)
    wavs = torch.as_tensor((rng.standard_normal((clips, 4, n)) * 0.1).astype(np.float32)).to(dev)
    fx = FE.FeatureExtractor(24000, "foa", 64, win_length=960, hop_length=480, n_fft=1024, device=dev.index)
    st = torch.cuda.current_stream(dev)

    def timed(fn, per_call):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.synchronize()
        e0.record(st)           # the extraction launches go to torch's current stream (FeatureExtractor.__call__ / .batch)
        for _ in range(reps):
            out = fn()
        e1.record(st)
        st.synchronize()
        return e0.elapsed_time(e1) / 1e3 / (reps * per_call), out

    per, out = timed(lambda: fx.batch(wavs), clips)
    per1, out1 = timed(lambda: [fx(wavs[i]) for i in range(clips)][-1], clips)
    bytes_clip = 4 * n * 4 + 3000 * 64 * 7 * 4
    assert tuple(out.shape) == (clips, 3001, 64, 7) and bool(torch.isfinite(out).all()) and bool(torch.equal(out[-1], out1))
    ach = bytes_clip / per / 1e9
    # the other mode of extract_features (feature_extractor.py:196-214: four log-mel channels + six GCC-PHAT pairs -> [3001,64,10]), same clips
    fm = FE.FeatureExtractor(24000, "mic", 64, win_length=960, hop_length=480, n_fft=1024, device=dev.index)
    per_m, out_m = timed(lambda: fm.batch(wavs), clips)
    bytes_mic = 4 * n * 4 + 3000 * 64 * 10 * 4
    assert tuple(out_m.shape) == (clips, 3001, 64, 10) and bool(torch.isfinite(out_m).all())
    mic = {"clips_per_s": round(1 / per_m, 1), "ms_per_clip": round(per_m * 1e3, 4), "algorithmic_bytes_per_clip": bytes_mic,
           "achieved_GBps": round(bytes_mic / per_m / 1e9, 1)}
    return {"stage": f"feature_extractor foa n_fft 1024 / win 960 / hop 480 -> [3001,64,7], 60-s clips resident in HBM, {clips} clips per launch pair", "mic_mode": mic,
            "clips_per_s": round(1 / per, 1), "ms_per_clip": round(per * 1e3, 4), "algorithmic_bytes_per_clip": bytes_clip,
            "clip_by_clip": {"clips_per_s": round(1 / per1, 1), "ms_per_clip": round(per1 * 1e3, 4)},
            "roofline": {"kernel": "feat_frame", "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(ach / PEAK_HBM_GBPS, 4), "traffic": None}}


def inference_leg(dev, local, files=4, reps=3, batch=271):
    """The reference's inference path (trainv2.ensemble_outputs, trainv2.py:158-192 = evaluator.py:16-50) on normalised 60-s feature files
    [3000,64,7] resident in HBM: 541 windows of 300 frames (step 5) through seldnet.json's forward in batches, overlap-averaged to
    [600, 12 | 36] — files/s and windows/s, timed with HIP events on the stream the calls go to."""
    from seld_amd import evaluator, models
    model = models.seldnet((batch, 300, 64, 7), model_config_of("seldnet"), device=local)
    rng = np.random.default_rng(1)
    xs = [torch.as_tensor(rng.standard_normal((3000, 64, 7)).astype(np.float32)).to(dev) for _ in range(files)]
    st = torch.cuda.current_stream(dev)
    for _ in range(2):
        out = evaluator.ensemble_outputs(model, xs, win_size=300, step_size=5, batch_size=batch)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.synchronize()
    e0.record(st)
    for _ in range(reps):
        out = evaluator.ensemble_outputs(model, xs, win_size=300, step_size=5, batch_size=batch)
    e1.record(st)
    st.synchronize()
    per = e0.elapsed_time(e1) / 1e3 / (reps * files)
    assert tuple(out[0][0].shape) == (600, 12) and tuple(out[0][1].shape) == (600, 36) and bool(torch.isfinite(out[-1][1]).all())
    model.close()
    return {"stage": f"ensemble_outputs: 60-s file [3000,64,7] -> 541 windows x 300 frames (step 5), batches of {batch}, overlap average -> [600,12|36]",
            "files_per_s": round(1 / per, 1), "ms_per_file": round(per * 1e3, 3), "windows_per_s": round(541 / per, 1),
            "clip_equivalents_per_s": round(54.1 / per, 1)}


MOTHER_STAGE_ARGS = {      # model_config/SS5.json's BLOCK0_ARGS (the reference's mother_stage configuration) — with the time stride of SS5's separate
    # first_pool_size [5, 2] carried by the stage's own strides, because models.seldnet has no pooling in front of FIRST: [1, 3] -> [5, 3]
    "depth": 2, "filters0": 0, "filters1": 96, "filters2": 0, "kernel_size0": 0, "kernel_size1": 3, "kernel_size2": 0,
    "connect0": [1], "connect1": [1, 0], "connect2": [1, 0, 1], "strides": [5, 3]}


def mother_stage_leg(dev, local, B=32, T=3000, steps=10, warmup=3):
    """models.seldnet with FIRST = mother_stage (reference modules.py:15-43, 184-298, the only conv FIRST-stage block the snapshot
    defines; SS5.json's BLOCK0 arguments) as a timed sub-record: the composed path (seld_amd/modules.py -> seld_m_* module operators, asynchronous
    on one stream), train.trainstep on B clips of [T,64,7].  Phases from HIP events on the stream the operators are launched on; the dominant
    phase is priced against the fp32 MFMA peak (its convolutions are im2col + v_mfma_f32_32x32x2_f32 products)."""
    from seld_amd import losses, models, train
    from seld_amd.synthetic import synthetic_batch
    cfg = model_config_of("seldnet")
    cfg["FIRST"], cfg["FIRST_ARGS"] = "mother_stage", dict(MOTHER_STAGE_ARGS)
    model = models.seldnet((B, T, 64, 7), cfg, device=local)
    x, ys, yd = (torch.as_tensor(a).to(dev) for a in synthetic_batch(B, T, seed=1234))
    opt = train.Adam(1e-3)
    args = (losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0), opt)
    for _ in range(warmup):
        train.trainstep(model, x, (ys, yd), *args)
    model.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        _, sl, _ = train.trainstep(model, x, (ys, yd), *args)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    assert np.isfinite(float(sl.item()))
    ph = {k: round(v / steps, 4) for k, v in model.phase_ms().items()}
    macs = model.conv_macs() * B
    dom = max(("first_fwd", "first_bwd"), key=lambda k: ph.get(k, 0.0))
    flop = 2 * macs * (1 if dom == "first_fwd" else 2)
    ach = flop / (ph[dom] / 1e3) / 1e12
    return {"value": round(B * steps / el, 2), "unit": "clips/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps, "warmup": warmup,
            "config": {"workload": f"models.seldnet with FIRST = mother_stage (reference modules.py:15-43, 184-298; SS5.json BLOCK0_ARGS, strides [5,3]), "
                                   f"train step, {B} clips of [T={T},64,7], composed from seld_m_* module operators on one stream",
                       "n_params": int(model.n_params)},
            "phase_ms_per_step": ph,
            "roofline": {"kernel": dom + " (im2col + gemm_f32 products, BatchNorm / activation / concatenation passes)", "bound": "mfma",
                         "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": None, "ms_per_step": ph[dom]}}


def model_config_of(name):
    import copy
    cfg = copy.deepcopy(SELDNET_CONFIG)
    if name == "xception_gru":        # model_config/xception_gru.json:2-11
        cfg["FIRST"] = "xception_block"
        cfg["FIRST_ARGS"] = {"filters": 32, "block_num": 8, "kernel_regularizer": {"l1": 0, "l2": 1e-3}}
    if name == "resnet50_gru":        # model_config/resnet50_gru.json:2-11
        cfg["FIRST"] = "resnet50_block"
        cfg["FIRST_ARGS"] = {"filters": 32, "block_num": list(RESNET_BLOCKS), "kernel_regularizer": {"l1": 0, "l2": 1e-3}}
    return cfg


def workload_text(name, B, T, with_features):
    return (f"model_config/{name}.json train step (fwd+BCE/MSE+bwd+Adam), {B} clips/GPU of [T={T},F=64,C=7], n_classes=12"
            + (" (FIRST block per spec/XCEPTION_BLOCK.md: absent from the reference snapshot)" if name == "xception_gru" else "")
            + (" (FIRST block per spec/RESNET50_BLOCK.md: absent from the reference snapshot)" if name == "resnet50_gru" else "")
            + (", features extracted on the device from 60-s FOA waveforms, normalised with statistics fitted on the device, inside every step"
               if with_features else ""))


def read_timers(model):
    import ctypes as C
    kernels = []
    for i in range(model.lib.seld_profile_count(model.ctx)):
        name = C.create_string_buffer(64)
        n, ms = C.c_int64(), C.c_double()
        model.lib.seld_profile_get(model.ctx, i, name, 64, C.byref(n), C.byref(ms))
        kernels.append((name.value.decode(), int(n.value), float(ms.value)))
    return kernels


def rooflines(kernels, B, T, steps, opts, traffic_tab, valu_clock_mhz, per_step_groups=False, bf16=False):
    """Per timed kernel group: achieved rate = algorithmic work / HIP-event time, against the gfx950 peak of its bound.  `per_step_groups`:
    the group's work table is per STEP (the block models' level-3 scopes wrap launches of very different sizes), else per launch."""
    per_kernel, breakdown = {}, {}

    def roof(name, n, ms):
        work = kernel_work(name, B, T)
        if not work or n == 0:
            return None
        bound, amount = work
        per_unit_s = (ms / steps if (per_step_groups and name.startswith("rn_") and name not in ("rn_stages_fwd", "rn_stages_bwd")) else ms / n) / 1e3
        path = None
        if bound == "mfma":
            split = name in SPLIT_BF16_GROUPS and not opts
            ach, unit = amount / per_unit_s / 1e12, "TFLOP/s"
            peak = round(PEAK_SPLIT_BF16_TFLOPS, 1) if split else PEAK_F32_MFMA_TFLOPS
            path = "fp32-equivalent FLOP on 6 bf16 MFMA products of exactly split operands" if split else "f32-input MFMA"
            if split and name in FOUR_PRODUCT_GROUPS:
                peak, path = round(PEAK_SPLIT4_BF16_TFLOPS, 1), "fp32-equivalent FLOP on 4 bf16 MFMA products (hi*hi, hi*mid, mid*hi, mid*mid; backward-only)"
            if bf16 and split:
                peak, path = PEAK_BF16_MFMA_TFLOPS, "one bf16 MFMA product per product (operands rounded to nearest bf16), dense bf16 peak"
        else:
            ach, peak, unit = amount / per_unit_s / 1e9, PEAK_HBM_GBPS, "GB/s"
        extra = {}
        if name in GRU_GROUPS and valu_clock_mhz:
            # the recurrence is a serial chain on 2B of the 256 CUs: HBM (what north_star asks to see) is not what bounds it.  Against the
            # fp32 rate of the CUs it occupies:
            cyc = per_unit_s * valu_clock_mhz * 1e6 / (T // 5)
            floor = GRU_FMA_PER_STEP / CU_FMA_PER_CLK
            extra = {"cus_used": 2 * B, "cycles_per_step": round(cyc, 1), "fp32_floor_cycles_per_step": floor,
                     "frac_of_fp32_peak_of_used_cus": round(floor / cyc, 4),
                     "clock_MHz": round(valu_clock_mhz), "phase_split": "profiles/r03_gru_experiments.txt (in-kernel timeline: the two waves of a SIMD "
                     "queue; gate tail = a chain of ~20 dependent VALU ops at 8.3-17.6 cycles each)"}
        tr = (traffic_tab.get(name) or {}).get("hbm_bytes_per_launch") if isinstance(traffic_tab.get(name), dict) else None
        if tr is not None:      # a STORED counter pass (another run of the build named in the table), never this run's
            extra["traffic_source"] = "profiles/traffic.json: " + str(traffic_tab.get("_provenance", ""))[:90]
        elif name in traffic_tab.get("_dropped", ()):
            extra["traffic_source"] = "profiles/traffic.json: stored figure dropped (kernel source changed since the counter pass)"
        return {"kernel": name, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
                "frac": round(ach / peak, 4), "traffic": tr,
                "avg_launch_ms": round(ms / n, 4), "launches_per_step": n // steps, "ms_per_step": round(ms / steps, 4),
                **({"mfma_path": path} if path else {}), **extra}

    for name, n, ms in kernels:
        breakdown[name] = round(ms / steps, 4)
        r = roof(name, n, ms)
        if r:
            per_kernel[name] = r
    return per_kernel, breakdown


def run_workload(name, B, T, steps, warmup, world, rank, local, dev, *, opts=(), timing_level=1, with_features=False, profile_level=0,
                 allreduce_ablation=False, dtype="float32"):
    """One BASELINE workload on this rank's GPU: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize
    (max over ranks), with the library's HIP-event timers at `timing_level` inside the timed region.  `profile_level` > timing_level:
    a SECOND pass of `steps` steps with the finer per-kernel scopes (hundreds of event pairs per step for the block models: they cost a few
    percent, so they never run in the pass that is timed for `value`)."""
    import ctypes as C
    from seld_amd import losses, models, train
    from seld_amd.synthetic import synthetic_batch      # oracle/ is imported by the cpu_baseline leg only
    dist = torch.distributed
    global N_PARAMS
    model = models.seldnet((B, T, 64, 7), model_config_of(name), device=local, dtype=dtype)
    N_PARAMS = model.n_params
    for kv in opts:
        key, _, val = kv.partition("=")
        model.set_option(key, int(val))
    if world > 1:
        # RCCL process group: the LIBRARY owns the communicator, the communication stream and the two gradient buckets (seld_dp_*);
        # over gloo (the one-GPU rehearsal) the torch.distributed path stays
        from seld_amd import parallel
        # SELD_DP=torch selects round 2's torch.distributed collectives on the library's buckets ON EVERY RANK (an environment
        # variable, the same for all ranks of a torch.distributed.run job).  There is no silent per-rank fallback: init_library_dp
        # raises on ALL ranks if RCCL cannot be bound / initialised on any of them, and the job ends.
        if os.environ.get("SELD_DP", "library") != "torch":
            dp_backend = "library-rccl" if parallel.init_library_dp(model) else f"torch-{dist.get_backend()}"
        else:
            dp_backend = f"torch-{dist.get_backend()}"
    else:
        dp_backend = "none"
    x, ys, yd = synthetic_batch(B, T, seed=1234 + rank)
    x, ys, yd = (torch.as_tensor(a).to(dev) for a in (x, ys, yd))  # inputs resident in HBM before timing
    opt = train.Adam(1e-3)
    sed_loss, doa_loss, lw = losses.BinaryCrossentropy(), losses.MSE, (1.0, 1000.0)
    featurize = None
    if with_features:
        from seld_amd import _lib, feature_extractor as FE
        if T != 3000:
            raise SystemExit("--with-features extracts 60-s clips: --frames must be 3000")
        rng = np.random.default_rng(77 + rank)
        wavs = torch.as_tensor((rng.standard_normal((B, 4, 1440000)) * 0.1).astype(np.float32)).to(dev)
        fx = FE.FeatureExtractor(24000, "foa", 64, win_length=960, hop_length=480, n_fft=1024, device=local)
        # the normaliser's statistics are FITTED on the device (feature_extractor.calculate_statistics, :218-224) from this rank's clips,
        # once, as the reference fits them once per dataset
        f_mean, f_std = FE.FeatureStatistics(64, 7, device=local).update(fx.batch(wavs)).result()
        f_mean, f_std = f_mean.reshape(-1).contiguous(), f_std.reshape(-1).contiguous()

        def featurize():
            st_ = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            f = fx.batch(wavs)                                   # [B, 3001, 64, 7]: one pair of launches
            for b in range(B):
                _lib.check(fx.lib.seld_feat_normalize(f[b].data_ptr(), f_mean.data_ptr(), f_std.data_ptr(), x[b].data_ptr(), 3001, 3000,
                                                      64 * 7, 1e-8, st_))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_pass(level, n, **kw):
        model.lib.seld_profile_reset(model.ctx)
        model.lib.seld_profile_enable(model.ctx, level)
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            if featurize:
                featurize()
            r = train.trainstep(model, x, (ys, yd), sed_loss, doa_loss, lw, opt, **kw)
        barrier()
        el = time.perf_counter() - t0
        model.lib.seld_profile_enable(model.ctx, 0)
        return el, r

    for _ in range(warmup):
        if featurize:
            featurize()
        train.trainstep(model, x, (ys, yd), sed_loss, doa_loss, lw, opt)
    elapsed, (y_p, sl, dl) = timed_pass(timing_level, steps)
    elapsed_local = elapsed
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert np.isfinite(float(sl.item())), "non-finite loss"
    kernels = read_timers(model)
    res = {"model": model, "elapsed": elapsed, "kernels": kernels, "comm": None, "profile": None, "dp_backend": dp_backend}
    if world > 1 and allreduce_ablation:
        # exposed communication per step and rank: the same steps again without the gradient all-reduce (timing aid only:
        # replicas then drift apart, nothing is reported from them but the time)
        barrier()
        t1 = time.perf_counter()
        for _ in range(steps):
            if featurize:
                featurize()
            train.trainstep(model, x, (ys, yd), sed_loss, doa_loss, lw, opt, allreduce=False)
        torch.cuda.synchronize()
        local_no_comm = time.perf_counter() - t1
        barrier()
        both = torch.tensor([elapsed_local, local_no_comm], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(both) for _ in range(world)]
        dist.all_gather(allr, both)
        res["comm"] = {"exposed_ms_per_step_by_rank": [round(float((a[0] - a[1]).item()) / steps * 1e3, 4) for a in allr],
                       "ms_per_step_without_allreduce_by_rank": [round(float(a[1].item()) / steps * 1e3, 4) for a in allr],
                       "note": "timed region repeated with the gradient all-reduce skipped; exposed = with - without, per rank"}
    if profile_level > timing_level:
        el2, _ = timed_pass(profile_level, steps)
        res["profile"] = {"kernels": read_timers(model), "ms_per_step_with_events": round(el2 / steps * 1e3, 3)}
    return res


def record(name, B, T, steps, warmup, world, res, opts, traffic_tab, valu_clock_mhz, with_features, dtype="float32"):
    bf16 = dtype == "bfloat16"
    per_kernel, breakdown = rooflines(res["kernels"], B, T, steps, opts, traffic_tab, valu_clock_mhz, bf16=bf16)
    fine = None
    if res["profile"]:
        fine, fine_breakdown = rooflines(res["profile"]["kernels"], B, T, steps, opts, traffic_tab, valu_clock_mhz, per_step_groups=True, bf16=bf16)
        # the dominant kernel is chosen among the FINE groups (a level-1 group such as rn_stages_bwd lumps products, BatchNorm passes and
        # waits for the side stream); groups that enclose other groups are not candidates
        cand = {k: v for k, v in fine.items() if k not in ("rn_stages_fwd", "rn_stages_bwd")}
        roofline = max(cand.values(), key=lambda r: r["ms_per_step"]) if cand else None
        breakdown = fine_breakdown
        per_kernel = fine
    else:
        roofline = max(per_kernel.values(), key=lambda r: r["ms_per_step"]) if per_kernel else None
    out = {
        "metric": METRIC, "value": round(world * B * steps / res["elapsed"], 2), "unit": "clips/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": round(res["elapsed"] / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
        "config": {"workload": workload_text(name, B, T, with_features) + (" — bf16 single-product mode (SELD_DTYPE_BF16): conv / GEMM operands rounded to "
                                                                             "bf16, fp32 accumulation and tensors; NOT within 1e-4 (DESIGN.md section 3c)" if bf16 else ""),
                   "global_batch": world * B, "parallelism": f"dp{world}", "dp_backend": res.get("dp_backend", "none"),
                   "doa_loss": "MSE", "loss_weight": "1,1000"},
        "roofline": roofline, "roofline_by_kernel": per_kernel, "kernel_ms_per_step": breakdown,
    }
    if res["profile"]:
        out["roofline_pass"] = {"what": "per-kernel HIP-event scopes in a second pass of the same steps (not the pass timed for `value`)",
                                "ms_per_step_with_events": res["profile"]["ms_per_step_with_events"]}
    return out


COMPACT_LIMIT = 3072      # bytes: the driver keeps an 8 KB tail of stdout; the record must fit it with room to spare


def compact_record(out, detail_path=None):
    """The LAST stdout line: the contract's fields + the dominant kernel's roofline + cpu_baseline + three scalars for the
    sub-configs.  Everything else (per-kernel rooflines, sub-records, peaks read on the box, comm ablation) is the DETAIL record,
    printed on an earlier line (prefix `BENCH_DETAIL `) and written to `detail_path`.  tests/test_host_logic_cpu.py holds the
    size bound on a canned full record."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    c = {k: out.get(k) for k in keep}
    cfg = out.get("config", {})
    c["config"] = {k: cfg.get(k) for k in ("workload", "global_batch", "parallelism", "dp_backend") if k in cfg}
    r = out.get("roofline")
    if r:
        rk = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_ms", "launches_per_step",
              "ms_per_step", "cycles_per_step", "fp32_floor_cycles_per_step", "frac_of_fp32_peak_of_used_cus", "cus_used")
        c["roofline"] = {k: r[k] for k in rk if k in r}
    else:
        c["roofline"] = None
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cb}
        c["cpu_baseline"]["sample"] = c["cpu_baseline"]["sample"][:260]
        if isinstance(cb.get("b2"), dict):
            c["cpu_baseline"]["b2_clips_s"] = cb["b2"].get("value")
        if cb.get("value"):
            c["gpu_over_cpu"] = round(out["value"] / cb["value"], 1)
    for key in ("seldnet_bf16", "xception_gru", "resnet50_gru", "mother_stage"):
        sub = out.get("configs", {}).get(key)
        if sub:
            c[f"{key}_clips_s"] = sub.get("value")
    if out.get("features"):
        c["features_clips_s"] = out["features"].get("clips_per_s")
        fr = out["features"].get("roofline") or {}
        c["features_hbm_frac"] = fr.get("frac")
    if out.get("inference"):
        c["inference_files_s"] = out["inference"].get("files_per_s")
    if out.get("conv_stack_mfma"):
        c["conv_stack_mfma"] = out["conv_stack_mfma"]
    if detail_path:
        c["detail"] = detail_path
    line = json.dumps(c)
    if len(line) > COMPACT_LIMIT:        # never let optional fields push the record out of the driver's tail
        for k in ("conv_stack_mfma", "features_hbm_frac", "detail", "gpu_over_cpu"):
            c.pop(k, None)
        c["config"]["workload"] = c["config"]["workload"][:200]
        if c.get("cpu_baseline"):
            c["cpu_baseline"]["sample"] = c["cpu_baseline"]["sample"][:120]
        line = json.dumps(c)
    assert len(line) <= COMPACT_LIMIT, len(line)
    return line


def conv_stack_mfma(per_kernel):
    """FLOP-weighted useful fraction of the conv stack's MFMA ceiling in THIS run (HIP-event times; the counter-based busy
    fraction is a separate rocprofv3 --pmc pass, profiles/*_mfma_util.json)."""
    rows = [r for k, r in per_kernel.items() if k.startswith("conv") and r["bound"] == "mfma"]
    if len(rows) < 7:       # conv1 fwd + conv2/3 fwd, dgrad, wgrad: only a timing-level-2 pass has them all
        return None
    t = sum(r["ms_per_step"] for r in rows)
    fl = sum(r["achieved"] * r["ms_per_step"] for r in rows)        # TFLOP/s x ms
    pk = sum(r["peak"] * r["ms_per_step"] for r in rows)
    return {"useful_frac_of_ceiling": round(fl / pk, 4), "TFLOPs_fp32_equivalent": round(fl / t, 1), "ms_per_step": round(t, 4)}


def spawn_ranks(n):
    """Launcher for `python bench.py --gpus N` run WITHOUT torch.distributed.run: N child processes of this same command line, one per
    GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (rendezvous on 127.0.0.1, a free port).  The parent never
    touches the GPU (children are fresh interpreters started before any HIP call; nothing is exec'd over an initialised process).  Rank 0's
    stdout is passed through unchanged, so its compact record stays the LAST stdout line; the other ranks' stdout goes to stderr with a
    rank prefix.  Returns the first non-zero exit code (the remaining ranks are then terminated), else 0."""
    import socket
    import subprocess
    import threading
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.PIPE, stderr=None, text=True))

    def relay(r, p):
        for line in p.stdout:
            sys.stderr.write(f"[rank {r}] {line}")

    threads = [threading.Thread(target=relay, args=(r, p), daemon=True) for r, p in enumerate(procs) if r]
    for t in threads:
        t.start()
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code
                sys.stderr.write(f"bench.py: rank {r} exited with {code}; stopping the other ranks\n")
                for q in live:
                    procs[q].terminate()        # the exact children started above, by handle
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=2)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed steps (default 40: a >= 100 ms timed region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None,
                    help="clips per GPU; default 32 (BASELINE.json configs[1]), 16 for --model resnet50_gru (configs[4]: 128 over 8 GPUs)")
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--timing-level", type=int, default=1, help="1: major kernel groups (default), 2: every group, 3: + per-kernel scopes of resnet50_block")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=0|1",
                    help="kernel-selection option of the C library (seld_set_option), for A/B runs; the default build is the product")
    ap.add_argument("--cpu-steps", type=int, default=10, help="cpu_baseline: timed steps per batch size (median), after --cpu-warmup")
    ap.add_argument("--cpu-warmup", type=int, default=3)
    ap.add_argument("--cpu-budget-s", type=float, default=150.0,
                    help="cpu_baseline at the GPU batch: fewer than --cpu-steps steps are timed (never fewer than 3) if they would exceed this")
    ap.add_argument("--no-features", action="store_true", help="skip the feature-stage leg")
    ap.add_argument("--no-inference", action="store_true", help="skip the sliding-window inference leg")
    ap.add_argument("--no-configs", action="store_true",
                    help="N=1, --model seldnet only: skip the `configs` sub-records (BASELINE configs[3] xception_gru and configs[4] resnet50_gru "
                         "with in-step features, each timed like the headline after it)")
    ap.add_argument("--model", default="seldnet", choices=["seldnet", "xception_gru", "resnet50_gru"],
                    help="model_config of the reference: seldnet.json (the headline, BASELINE configs[1]), xception_gru.json "
                         "(configs[3]) or resnet50_gru.json (configs[4]); the FIRST blocks of the latter two are defined by "
                         "spec/XCEPTION_BLOCK.md / spec/RESNET50_BLOCK.md: absent from the reference snapshot")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the DETAIL record (per-kernel rooflines, sub-records, peaks read on the box) is written; it is also printed "
                         "on the line before the compact record, prefixed BENCH_DETAIL")
    ap.add_argument("--with-features", action="store_true",
                    help="configs[4]'s 'on-device STFT feature_extractor': every timed step first extracts and normalises the features of "
                         "its clips from 60-s FOA waveforms resident in HBM (feature_extractor.py:53-88, data_loader.py:117-149,226-234)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 16 if args.model == "resnet50_gru" else 32

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without an outer launcher: this process becomes the launcher (it has made no GPU call yet and makes
        # none) and starts one fresh rank process per GPU; under torch.distributed.run WORLD_SIZE is set and this branch is not taken
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    probe = os.environ.get("SELD_BENCH_LAUNCH_PROBE")
    if probe:       # tests/test_host_logic_cpu.py: what a rank was started with, before anything touches a GPU ("fail1": rank 1 fails, the rest linger)
        print(json.dumps({"rank": rank, "local": local, "world": world, "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                          "argv": sys.argv[1:]}), flush=True)
        if probe == "fail1":
            if rank == 1:
                raise SystemExit(3)
            time.sleep(60)
        raise SystemExit(0)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or unset WORLD_SIZE and let "
                         f"bench.py start its own ranks")
    # rehearsal of the N > 1 code path on a one-GPU box (tests/test_dp_gpu.py::test_bench_two_ranks_rehearsal): every rank on device
    # SELD_BENCH_DEVICE over the gloo backend (RCCL refuses two ranks on one device).  The driver's runs set neither variable.
    if os.environ.get("SELD_BENCH_DEVICE") is not None:
        local = int(os.environ["SELD_BENCH_DEVICE"])
    torch.cuda.set_device(local)
    dist = torch.distributed
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SELD_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    B, T = args.batch, args.frames
    dev = torch.device("cuda", local)
    # run on a non-default stream: the legacy null stream serialises against every other blocking stream
    # and makes event records / launches heavier
    run_stream = torch.cuda.Stream(device=dev)
    run_stream.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(run_stream)
    timing = 0 if args.no_kernel_timing else args.timing_level
    res = run_workload(args.model, B, T, args.steps, args.warmup, world, rank, local, dev, opts=args.opt, timing_level=timing,
                       with_features=args.with_features, allreduce_ablation=True,
                       # N = 1: a SECOND pass of the same steps with every kernel group timed (never the pass timed for `value`): the
                       # per-kernel rooflines of the detail record and the conv stack's MFMA figure of the compact one
                       profile_level=2 if (world == 1 and timing == 1) else 0)
    if rank == 0:
        import ctypes as C
        model = res["model"]
        mhz = C.c_double()
        valu_clock_mhz = float(mhz.value) if model.lib.seld_k_valu_clock_mhz(2 * B, C.byref(mhz)) == 0 else None
        traffic_tab = load_traffic()
        out = record(args.model, B, T, args.steps, args.warmup, world, res, args.opt, traffic_tab, valu_clock_mhz, args.with_features)
        out["conv_stack_mfma"] = conv_stack_mfma(out["roofline_by_kernel"])
        out["peaks_on_box"] = box_peaks(model.lib, local)
        if res["comm"]:
            out["comm"] = res["comm"]
        if world == 1 and not args.no_features:
            out["features"] = features_leg(dev)
            # PMC bytes of the extraction kernel per LAUNCH (tools/bench_features.py: 8 clips per launch) -> per clip, like `achieved`
            ft = (traffic_tab.get("feat_dft") or {}).get("hbm_bytes_per_launch")
            out["features"]["roofline"]["traffic"] = None if ft is None else int(ft / 8)
            out["features"]["roofline"]["kernel"] = "feat_dft"
        del model
        res["model"] = None
        if world == 1 and args.model == "seldnet" and not args.no_inference and not args.opt:
            out["inference"] = inference_leg(dev, local)
        if world == 1 and args.model == "seldnet" and not args.no_configs and not args.opt and T == 3000:
            # BASELINE.json configs[3] and configs[4] as sub-records of the same driver-timed run: each its own model, warm-ups, barrier-
            # bracketed timed steps and rooflines (configs[4] = the per-GPU share, 16 clips, of the batch-128 DP-8 job, features in the step)
            out["configs"] = {}
            sub_steps = max(10, args.steps // 2)
            # configs[1]'s literal wording ("seldnet.json bf16 batch=32"): the same workload in bf16 single-product mode, as a SECOND record —
            # the headline above stays the fp32-equivalent mode that meets the 1e-4 parity bar
            for key, sub, sb, feat, lvl, dt in (("seldnet_bf16", "seldnet", 32, False, 2, "bfloat16"), ("xception_gru", "xception_gru", 32, False, 2, "float32"),
                                                ("resnet50_gru", "resnet50_gru", 16, True, 3, "float32")):
                torch.cuda.empty_cache()
                r = run_workload(sub, sb, T, sub_steps, args.warmup, world, rank, local, dev, timing_level=1, with_features=feat,
                                 profile_level=0 if args.no_kernel_timing else lvl, dtype=dt)
                out["configs"][key] = record(sub, sb, T, sub_steps, args.warmup, world, r, (), traffic_tab, valu_clock_mhz, feat, dtype=dt)
                out["configs"][key].pop("metric")
                r["model"] = None
                del r
            torch.cuda.empty_cache()
            out["configs"]["mother_stage"] = mother_stage_leg(dev, local, steps=sub_steps)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, T, args.cpu_steps, args.cpu_warmup, args.cpu_budget_s, model_config_of(args.model))
        # DETAIL first (one long line + a file), the compact record LAST: the driver parses the last stdout line from an 8 KB tail
        detail_path = args.detail_out
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f)
        except OSError:
            detail_path = None
        print("BENCH_DETAIL " + json.dumps(out), flush=True)
        print(compact_record(out, os.path.relpath(detail_path, ROOT) if detail_path else None), flush=True)
    if res.get("model") is not None:
        res["model"].close()       # every rank: the library's communicator goes before the host's process group does
        res["model"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
