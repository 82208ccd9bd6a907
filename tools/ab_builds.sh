#!/bin/bash
# Same-box A/B of two builds of the library (boxes differ by ~5 %, so only same-box numbers compare):
#   make -C seld_amd/csrc && cp seld_amd/libseld_hip.so seld_amd/libseld_hip_prev.so    # build A (e.g. of the previous commit)
#   ... change code, make ...                                                            # build B = seld_amd/libseld_hip.so
#   gpurun -- 'bash tools/ab_builds.sh [bench args]'
# Alternates B, A, B, A and prints clips/s and ms/step of each run.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
for v in new prev new prev; do
    if [ $v = prev ]; then export SELD_HIP_LIB=$root/seld_amd/libseld_hip_prev.so; else unset SELD_HIP_LIB; fi
    timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null |
        python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
