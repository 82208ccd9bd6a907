"""MFMA utilisation per kernel from ONE rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE with --kernel-trace only):
    util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * n_simd * scale),  n_simd = 256 CUs x 4
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md); what
GRBM_GUI_ACTIVE counts per dispatch on an 8-XCD part is CALIBRATED, not assumed: `scale` is chosen so that the back-to-back MFMA probe
(tools/mfma_probe.hip: every SIMD issues MFMAs without a gap) reads 100 % — pass the probe's CSV as the calibration file.
usage: pmc_mfma.py <bench counter_collection.csv> <probe counter_collection.csv> <provenance note> > profiles/mfma_util.json"""
import csv
import json
import sys
from collections import defaultdict

N_SIMD = 256 * 4
GROUPS = {  # group -> (kernel-name substring, fp32-equivalent GFLOP per launch at B=32, T=3000 or None)
    "conv1_fwd": "conv_first_fwd_pool_sb_kernel", "conv1_gram (side stream)": "conv_first_gram_kernel",
    "conv2_fwd_dgrad (W=16)": "conv64_fwd_sbd_kernel<4", "conv3_fwd_dgrad (W=4)": "conv64_fwd_sbd_kernel<2",
    "conv2_wgrad": "conv64_wgrad_sb_kernel<4", "conv3_wgrad": "conv64_wgrad_sb_kernel<2",
    "gemm_sb (4-wave: GRU in-projections, dX)": "gemm_sb_kernel", "gemm_sb16 (16-wave)": "gemm_sb16_kernel",
    "gemm_tn_sb (GRU kernel gradients)": "gemm_tn_sb", "gemm_f32 (heads)": "gemm_f32_kernel",
    "gru_fwd": "gru_fwd_kernel", "gru_bwd": "gru_bwd_kernel",
    # feature stage (tools/profile_mfma_features.sh): the transforms, the mel projection and (mic) GCC-PHAT's inverse transform on the matrix cores
    "feat_dft (foa, n_fft 1024)": "feat_dft_kernel<3, 0>", "feat_dft (mic, n_fft 1024)": "feat_dft_kernel<3, 1>",
}


def load(path):
    rows = defaultdict(lambda: defaultdict(float))     # dispatch id -> counter -> value
    names = {}
    for r in csv.DictReader(open(path)):
        d = r.get("Dispatch_Id") or r.get("Correlation_Id")
        rows[d][r["Counter_Name"]] += float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    return rows, names


def per_kernel(path):
    rows, names = load(path)
    acc = defaultdict(lambda: [0.0, 0.0, 0])
    for d, c in rows.items():
        a = acc[names[d]]
        a[0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[1] += c.get("GRBM_GUI_ACTIVE", 0.0)
        a[2] += 1
    return acc


bench, probe = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
# calibration: the bf16 probe kernel (probe<1>) keeps every matrix pipe busy
cal = [v for k, v in probe.items() if "probe<1>" in k]
assert cal and cal[0][1] > 0, "calibration kernel probe<1> not found in " + sys.argv[2]
scale = cal[0][0] / (cal[0][1] * N_SIMD)
out = {"_provenance": sys.argv[3],
       "_method": "util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs * scale); scale calibrated so that back-to-back "
                  "v_mfma_f32_32x32x16_bf16 on every SIMD (tools/mfma_probe.hip) reads 1.0",
       "_calibration": {"probe_busy_cycles_per_launch": cal[0][0] / cal[0][2], "probe_gui_active_per_launch": cal[0][1] / cal[0][2], "scale": scale}}
for grp, pat in GROUPS.items():
    ks = [k for k in bench if pat in k]
    if not ks:
        continue
    busy = sum(bench[k][0] for k in ks)
    act = sum(bench[k][1] for k in ks)
    n = sum(bench[k][2] for k in ks)
    out[grp] = {"mfma_util": round(busy / (act * N_SIMD * scale), 4) if act else None, "mfma_busy_cycles_per_launch": int(busy / n),
                "launches_sampled": n, "kernel": ks[0][:80]}
# the conv stack, weighted by fp32-equivalent FLOP per step (SURVEY.md section 8(d) MAC counts at B=32, T=3000: conv1 fwd 49.5 GFLOP; conv2 22.6 each
# for fwd / dgrad / wgrad; conv3 5.7 each; conv1's kernel gradient runs as the Gram matrix on the side stream and is listed, not weighted)
W = {"conv1_fwd": 49.5, "conv2_fwd_dgrad (W=16)": 2 * 22.6, "conv2_wgrad": 22.6, "conv3_fwd_dgrad (W=4)": 2 * 5.7, "conv3_wgrad": 5.7}
if all(k in out and out[k]["mfma_util"] is not None for k in W):
    out["conv_stack_flop_weighted_mfma_util"] = round(sum(W[k] * out[k]["mfma_util"] for k in W) / sum(W.values()), 4)
json.dump(out, sys.stdout, indent=1)
