#!/bin/bash
# MFMA-utilisation counters of the headline step (north_star: "MFMA utilisation on the conv stack against gfx950 peak"), run on the GPU box:
#   gpurun -- 'bash tools/profile_mfma.sh r03_a'
# One PMC pass over a short bench run + one over the back-to-back MFMA probe (the calibration), --kernel-trace only, programs directly
# after `--`.  Writes gpurun_out/mfma_<tag>/mfma_util.json (tools/pmc_mfma.py); copy it to profiles/mfma_util.json.
set -e -o pipefail
tag=${1:-round}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/mfma_$tag
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$root/tools/mfma_probe.hip" -o /tmp/mfma_probe 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_bench" -o pmc -- \
    python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-features --no-inference --no-kernel-timing --no-configs --detail-out /dev/null > /dev/null 2> "$out/rocprof_bench.log"
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_probe" -o pmc -- \
    /tmp/mfma_probe > "$out/probe.txt" 2> "$out/rocprof_probe.log"
b=$(find "$out/pmc_bench" -name '*counter_collection.csv' | head -1)
p=$(find "$out/pmc_probe" -name '*counter_collection.csv' | head -1)
python3 "$root/tools/pmc_mfma.py" "$b" "$p" "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, --kernel-trace only), build $tag" > "$out/mfma_util.json"
head -c 600 "$b" > "$out/counter_collection_head.csv"
rm -rf "$out/pmc_bench" "$out/pmc_probe"
cat "$out/mfma_util.json"
