"""Per-launch-shape kernel durations from a rocprofv3 kernel trace: the same kernel runs at several sizes inside
bench.py (100k-read scan launches, 20k-read dense launches), which the --stats average mixes.
usage: python tools/summarize_trace.py gpurun_out/<tag>_stats/out_kernel_trace.csv profiles/<name>.json"""
import collections
import csv
import json
import sys

src, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
with open(src) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "motifs::" not in name:
            continue
        key = (name.replace("motifs::", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]),
               int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = [{"kernel": k[0], "blocks": [k[1], k[2], k[3]], "threads": k[4], "calls": len(v), "avg_us": sum(v) / len(v),
         "min_us": min(v), "max_us": max(v), "total_ms": sum(v) / 1e3} for k, v in acc.items()]
rows.sort(key=lambda r: -r["total_ms"])
json.dump({"source": src, "note": "durations by (kernel, launch shape); rocprofv3 --kernel-trace of bench.py --steps 5 --warmup 1 "
           "--no-cpu --train-steps 2", "rows": rows}, open(out, "w"), indent=1)
for r in rows[:12]:
    print("%-34s blocks %-18s calls %4d avg %8.1f us" % (r["kernel"][:34], r["blocks"], r["calls"], r["avg_us"]))
