"""Per-kernel HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each with
--kernel-trace only), as MI355X_MICROARCH.md prescribes: bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024
(FETCH_SIZE/WRITE_SIZE count KiB; gfx950 reports half of a wide coalesced read stream).
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <provenance note> > traffic.json"""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "seld_amd", "csrc")


def source_hashes():
    """sha256 (16 hex) of every kernel source: bench.py drops a stored figure whose kernel's source has changed since the counter pass."""
    return {f: hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:16]
            for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))}


GROUPS = {  # bench.py timer group -> kernel-name substring
    "conv1_fwd": "conv_first_fwd_pool_sb_kernel", "conv1_fwd_f32": "conv_first_fwd_pool_kernel",
    "conv1_wgrad": "conv_first_msparse_kernel", "conv1_gram": "conv_first_gram_kernel",
    "conv1_wgrad_fused": "conv_first_wgrad_fused_kernel",
    "gru_fwd": "gru_fwd_kernel", "gru_bwd": "gru_bwd_kernel",
    "conv64_fwd_dgrad_W16": "conv64_fwd_sbd_kernel<4", "conv64_fwd_dgrad_W4": "conv64_fwd_sbd_kernel<2",
    "conv64_fwd_dgrad_W16_sbr": "conv64_fwd_sbr_kernel<4", "conv64_fwd_dgrad_W4_sbr": "conv64_fwd_sbr_kernel<2",
    "feat_dft": "feat_dft_kernel", "feat_frame": "feat_wave_kernel", "feat_frame_workgroup": "feat_frame_kernel", "feat_topdb": "feat_topdb_kernel",
    "conv2_wgrad": "conv64_wgrad_sb_kernel<4", "conv3_wgrad": "conv64_wgrad_sb_kernel<2",
    "conv2_wgrad_f32": "conv64_wgrad_kernel<4>", "conv3_wgrad_f32": "conv64_wgrad_kernel<2>",
    "pool1_fwd": "bn_relu_ext_kernel", "gemm": "gemm_f32_kernel", "gemm_tn": "gemm_tn_kernel",
    "gemm_sb_4wave": "gemm_sb_kernel", "gemm_sb_16wave": "gemm_sb16_kernel",
}


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_provenance": sys.argv[3], "_source_hashes": source_hashes()}
for grp, pat in GROUPS.items():
    names = [k for k in fetch if pat in k]
    if not names:
        continue
    n = sum(nf[k] for k in names)
    fb = sum(fetch[k] * nf[k] for k in names) / n * 1024 * 2
    wb = sum(write.get(k, 0.0) * nf[k] for k in names) / n * 1024
    out[grp] = {"hbm_bytes_per_launch": int(fb + wb), "fetch_bytes": int(fb), "write_bytes": int(wb), "launches_sampled": n,
                "kernel": names[0][:80]}
json.dump(out, sys.stdout, indent=1)
