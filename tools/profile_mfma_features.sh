#!/bin/bash
# MFMA-utilisation counters of the feature stage (foa and mic at n_fft 1024), run on the GPU box:  gpurun -- 'bash tools/profile_mfma_features.sh r03_f'
# As tools/profile_mfma.sh: one PMC pass per program (--kernel-trace only, the program directly after `--`), calibrated on the MFMA probe.
set -e -o pipefail
tag=${1:-round}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/mfma_feat_$tag
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$root/tools/mfma_probe.hip" -o /tmp/mfma_probe 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_probe" -o pmc -- /tmp/mfma_probe > "$out/probe.txt" 2> "$out/rocprof_probe.log"
p=$(find "$out/pmc_probe" -name '*counter_collection.csv' | head -1)
for prog in bench_features bench_features_other_modes; do
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_$prog" -o pmc -- \
        python3 "$root/tools/$prog.py" > /dev/null 2> "$out/rocprof_$prog.log"
    b=$(find "$out/pmc_$prog" -name '*counter_collection.csv' | head -1)
    python3 "$root/tools/pmc_mfma.py" "$b" "$p" "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on tools/$prog.py, build $tag" > "$out/mfma_util_$prog.json"
    rm -rf "$out/pmc_$prog"
done
rm -rf "$out/pmc_probe"
cat "$out"/mfma_util_*.json
