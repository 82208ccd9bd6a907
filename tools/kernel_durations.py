"""Print per-kernel durations (us) from a rocprofv3 --kernel-trace results .db, in dispatch order, grouped."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
prev, acc = None, []
for name, s, e in rows + [(None, 0, 0)]:
    if name != prev and prev is not None:
        if pat in prev:
            print("%-60s n=%3d  %s" % (prev[:60], len(acc), " ".join("%.1f" % a for a in acc[:12])))
        acc = []
    prev = name
    acc.append((e - s) / 1e3)
