#!/bin/bash
# One round's evidence set, run on the GPU box:  gpurun -- 'bash tools/profile_round.sh r01_h'
# Writes gpurun_out/prof_<tag>/: bench.json (the compact record: bench.py's last line), bench_detail.json (every kernel group timed),
# stats_kernel_stats.csv + bench_under_rocprof.json (rocprofv3 --kernel-trace --stats of the same command),
# traffic.json (HBM bytes per launch from two separate PMC passes, tools/pmc_traffic.py).
set -e -o pipefail
tag=${1:-round}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd "$root"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -o pmc -- \
        python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-features --no-inference --no-kernel-timing --no-configs --detail-out /dev/null > /dev/null 2> "$out/rocprof_$c.log"
done
f=$(find "$out/pmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
w=$(find "$out/pmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
python3 "$root/tools/pmc_traffic.py" "$f" "$w" "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, build $tag" > "$out/traffic.json"
# feature stage (feature_extractor.extract_features on 60-s FOA clips): kernel stats + its own PMC passes, merged into traffic.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/feat_stats" -o feat -- \
    python3 "$root/tools/bench_features.py" > "$out/bench_features_under_rocprof.json" 2> "$out/rocprof_feat.log"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/featpmc_$c" -o pmc -- \
        python3 "$root/tools/bench_features.py" > /dev/null 2> "$out/rocprof_feat_$c.log"
done
ff=$(find "$out/featpmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
fw=$(find "$out/featpmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
python3 "$root/tools/pmc_traffic.py" "$ff" "$fw" "feature stage: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, build $tag" > "$out/traffic_features.json"
rm -rf "$out/featpmc_FETCH_SIZE" "$out/featpmc_WRITE_SIZE"
# the counter passes come FIRST and their figures replace profiles/traffic.json of this copy of the repo, so that the bench line below carries
# `roofline.traffic` measured on the build it times (bench.py drops a stored figure whose kernel source changed since the pass)
python3 - "$out/traffic.json" "$out/traffic_features.json" "$root/profiles/traffic.json" <<'PY'
import json, sys
a, b = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
assert a["_source_hashes"] == b["_source_hashes"]
for k, v in b.items():
    if k == "_provenance": a["_provenance_features"] = v
    elif k != "_source_hashes": a[k] = v
json.dump(a, open(sys.argv[3], "w"), indent=1)
PY
cp "$root/profiles/traffic.json" "$out/traffic_merged.json"
cd "$root"
# bench.py prints the DETAIL record on an earlier line and the compact record LAST: the compact line goes to bench.json, the detail
# (per-kernel rooflines from the level-2 profile pass, sub-records, peaks read on the box) to bench_detail.json
timeout -k 10 500 python3 bench.py --detail-out "$out/bench_detail.json" 2> "$out/bench.err" | tail -n 1 > "$out/bench.json"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-features --no-inference --no-configs --no-kernel-timing --detail-out "$out/bench_under_rocprof_detail.json" 2> "$out/rocprof_stats.log" | tail -n 1 > "$out/bench_under_rocprof.json"
find "$out/feat_stats" -name '*kernel_trace.csv' -delete
# seldnet.json in bf16 single-product mode (BASELINE configs[1]'s literal wording): kernel stats
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bf16_stats" -o bf16 -- \
    python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-features --no-inference --no-configs --no-kernel-timing --opt bf16_single=1 --detail-out /dev/null 2> "$out/rocprof_bf16.log" | tail -n 1 > "$out/bench_bf16_under_rocprof.json"
find "$out/bf16_stats" -name '*kernel_trace.csv' -delete
# BASELINE config 4 (xception_gru.json; FIRST block per spec/XCEPTION_BLOCK.md): bench line + kernel stats
cd "$root"
timeout -k 10 200 python3 bench.py --model xception_gru --steps 10 --warmup 3 --no-cpu-baseline --no-features --detail-out "$out/bench_xception_gru_detail.json" 2>> "$out/bench.err" | tail -n 1 > "$out/bench_xception_gru.json"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/xc_stats" -o xc -- \
    python3 "$root/bench.py" --model xception_gru --steps 5 --warmup 2 --no-cpu-baseline --no-features --no-kernel-timing --detail-out /dev/null > /dev/null 2> "$out/rocprof_xc.log"
find "$out/xc_stats" -name '*kernel_trace.csv' -delete
# BASELINE config 5 (resnet50_gru.json; FIRST block per spec/RESNET50_BLOCK.md, 16 clips): bench line + kernel stats
cd "$root"
timeout -k 10 200 python3 bench.py --model resnet50_gru --steps 10 --warmup 3 --no-cpu-baseline --no-features --detail-out "$out/bench_resnet50_gru_detail.json" 2>> "$out/bench.err" | tail -n 1 > "$out/bench_resnet50_gru.json"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rn_stats" -o rn -- \
    python3 "$root/bench.py" --model resnet50_gru --steps 5 --warmup 2 --no-cpu-baseline --no-features --no-kernel-timing --detail-out /dev/null > /dev/null 2> "$out/rocprof_rn.log"
find "$out/rn_stats" -name '*kernel_trace.csv' -delete
rm -rf "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE"   # raw per-dispatch counters are large; the per-kernel summary stays
find "$out/stats" -name '*kernel_trace.csv' -delete
cat "$out/bench.json"
