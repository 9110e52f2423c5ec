#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace results .db: start offset, duration and the gap to the previous kernel's end, in launch order.
usage: tools/trace_timeline.py <results.db> [first] [count]"""
import sqlite3
import subprocess
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x/d.workgroup_size_x, d.grid_size_y from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
names = list({r[0] for r in rows})
dem = dict(zip(names, subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()))
t0 = rows[first][1]
prev = None
for name, s, e, bx, by in rows[first:first + count]:
    d = dem[name].split("(")[0].replace("void ", "").replace("motifs::", "").replace(".kd", "")
    print("%9.1f us  +%7.1f  dur %8.1f  %-44s [%d,%d]" % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0.0, (e - s) / 1e3, d[:44], bx, by))
    prev = e
