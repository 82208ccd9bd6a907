#!/usr/bin/env python3
"""One steady-state training step from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): every kernel in start order with its queue,
start offset from the step's first kernel, duration and the idle gap to the previous kernel of the same queue.
    python tools/step_timeline.py tr_kernel_trace.csv [marker kernel substring] [step index from the end, default 2]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "weight_prep_kernel"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = starts[-back - 1], starts[-back]
t0 = int(rows[lo]["Start_Timestamp"])
last_end = {}
print(f"step = kernels {lo}..{hi - 1}, {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us first kernel to first kernel")
queues = {}
for r in rows[lo:hi]:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"q{q} {(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {r['Kernel_Name'][:70]}")
