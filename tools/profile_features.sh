set -e -o pipefail
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/prof_r03_c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/feat_stats" -o feat -- python3 "$root/tools/bench_features.py" > "$out/bench_features_under_rocprof.json" 2> "$out/rocprof_feat.log"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/featpmc_$c" -o pmc -- python3 "$root/tools/bench_features.py" > /dev/null 2> "$out/rocprof_feat_$c.log"
done
ff=$(find "$out/featpmc_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
fw=$(find "$out/featpmc_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
python3 "$root/tools/pmc_traffic.py" "$ff" "$fw" "feature stage: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, build r03_c" > "$out/traffic_features.json"
rm -rf "$out/featpmc_FETCH_SIZE" "$out/featpmc_WRITE_SIZE"
find "$out/feat_stats" -name '*kernel_trace.csv' -delete
cd $root && python3 tools/bench_features.py > $out/bench_features.json 2>/dev/null
cat $out/bench_features.json; cat $out/traffic_features.json | head -30
