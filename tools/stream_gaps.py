#!/usr/bin/env python3
"""Main-stream occupancy of a training step from a rocprofv3 --kernel-trace results .db: per queue, the busy time, the idle gaps
between consecutive kernels and the ten largest gaps (with the kernels either side) inside the steady-state steps.
    python tools/stream_gaps.py results.db [first_step_marker_kernel]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
marker = sys.argv[2] if len(sys.argv) > 2 else "conv_first_fwd_pool"
starts = [i for i, r in enumerate(rows) if marker in r[0]]
if len(starts) < 6:
    sys.exit("marker kernel not found often enough")
lo, hi = starts[len(starts) // 2], starts[-2]          # steady-state steps only
steps = sum(1 for i in starts if lo <= i < hi)
sel = rows[lo:hi]
span = (sel[-1][1] - sel[0][1]) / 1e3
print(f"{steps} steps, {span / steps:.1f} us per step (first kernel to first kernel)")
byq = defaultdict(list)
for r in sel:
    byq[r[3]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for _, s, e, _ in rs) / 1e3
    gaps = [(rs[i + 1][1] - rs[i][2], rs[i][0][:40], rs[i + 1][0][:40]) for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"queue {q}: {len(rs) / steps:.1f} kernels per step, busy {busy / steps:.1f} us per step, "
          f"gaps {sum(g[0] for g in pos) / 1e3 / steps:.1f} us per step (median gap {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3:.2f} us)")
    agg = defaultdict(lambda: [0, 0.0])
    for g, a, b in pos:
        agg[(a, b)][0] += 1
        agg[(a, b)][1] += g / 1e3
    for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"    {t / steps:6.2f} us per step  ({n / steps:.1f} x {t / n:5.2f} us)  {a} -> {b}")
