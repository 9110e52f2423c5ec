"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh into profiles/<tag>_scan_hbm_traffic.json.

Entries are keyed by (kernel, grid size): tools/prof_scan.py launches the same kernel at several sizes (100k-read scan
launches, 20k-read dense launches), and an average over all launches of a kernel is the traffic of none of them (round 2
reported 462 MB for a candidate launch that moves 629 MB).  bench.py takes the entry whose grid matches the launch it timed.

Units and corrections as MI355X_MICROARCH.md prescribes: the counters are in KB; on gfx950 FETCH_SIZE tallies the
128-B requests of a wide coalesced read at 64 B, so the read side is doubled; WRITE_SIZE is exact.
usage: python tools/summarize_traffic.py <tag> <out.json> [commit]"""
import collections
import csv
import json
import sys

tag, out = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    with open(f"gpurun_out/{tag}_pmc_{ctr}/out_counter_collection.csv") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == ctr and "motifs::" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("motifs::", "")
                acc[(name, int(r["Grid_Size"]), int(r["Workgroup_Size"]))][ctr].append(float(r["Counter_Value"]))
res = {"commit": commit,
       "command": f"bash tools/profile_round.sh {tag}  (on the GPU box: rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 "
                  f"tools/prof_scan.py; the same with WRITE_SIZE), then python tools/summarize_traffic.py {tag} <out> <commit>",
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/prof_scan.py at BASELINE configs[1] (gpu_scan: both strands per call, as bench.py drives it; 100k reads x "
               "200 bp, 200 PWMs len 12; dense launches: 20k reads).  One entry per (kernel, grid size in threads): the average over the "
               "launches of THAT shape.  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 read-side correction).",
       "launches": []}
for (name, grid, wg), v in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
    w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    res["launches"].append({"kernel": name, "grid_threads": grid, "workgroup": wg, "blocks": grid // wg, "launches": len(v["FETCH_SIZE"]),
                            "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024})
json.dump(res, open(out, "w"), indent=1)
for e in res["launches"]:
    print("%-34s blocks %8d x %4d  launches %d  %8.1f MB" % (e["kernel"][:34], e["blocks"], e["workgroup"], e["launches"], e["hbm_bytes_per_launch"] / 1e6))
