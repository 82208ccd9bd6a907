"""Main-stream idle time inside one training step, from a rocprofv3 --kernel-trace CSV: per stream the busy time, and the idle gaps between consecutive
main-stream kernels grouped by the pair of kernels either side (an event wait on the side stream shows up here).
    python tools/step_gaps.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
lo, hi = idx[-3] + 1, idx[-2] + 1
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in step)
by = collections.defaultdict(float)
for r in step:
    by[r["Stream_Id"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"step {(t1 - t0) / 1e3:.1f} us, {len(step)} kernels; busy per stream: " + ", ".join(f"{k}: {v:.1f}" for k, v in by.items()))
main = max(by, key=by.get)
ms = [r for r in step if r["Stream_Id"] == main]
short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:34]
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for a, b in zip(ms, ms[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > 0:
        tot += g
    if g > 2:
        e = agg[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))]
        e[0] += 1
        e[1] += g
print(f"main-stream idle {tot:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {v[1]:8.1f} us in {v[0]:3d} gaps  {k[0]} -> {k[1]}")
