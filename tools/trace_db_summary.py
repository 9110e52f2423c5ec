#!/usr/bin/env python3
"""Per (kernel, blocks) durations from a rocprofv3 results .db (--kernel-trace): usage: tools/trace_db_summary.py <results.db> [n] [filter]"""
import collections
import sqlite3
import subprocess
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
flt = sys.argv[3] if len(sys.argv) > 3 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute(f"select s.kernel_name, d.grid_size_x/d.workgroup_size_x, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id").fetchall()
acc = collections.defaultdict(list)
for name, b, t in rows:
    acc[(name, b)].append(t / 1e3)
names = {k[0] for k in acc}
dem = dict(zip(names, subprocess.run(["c++filt"] + list(names), capture_output=True, text=True).stdout.splitlines())) if names else {}
tot = sum(sum(v) for v in acc.values())
print("total %.3f ms over %d launches" % (tot / 1e3, len(rows)))
for (name, b), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    d = dem.get(name, name).split("(")[0].replace("void ", "").replace("motifs::", "").replace(".kd", "")
    if flt and flt not in d:
        continue
    print("%-40s blocks %6d calls %4d avg %8.1f us  min %8.1f  total %8.2f ms" % (d[:40], b, len(v), sum(v) / len(v), min(v), sum(v) / 1e3))
    n -= 1
    if n == 0:
        break
