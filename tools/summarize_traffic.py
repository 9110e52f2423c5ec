"""Fold the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh into profiles/<tag>_scan_hbm_traffic.json.

Units and corrections as MI355X_MICROARCH.md prescribes: the counters are in KB; on gfx950 FETCH_SIZE tallies the
128-B requests of a wide coalesced read at 64 B, so the read side is doubled; WRITE_SIZE is exact."""
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
            if r["Counter_Name"] == ctr:
                acc[r["Kernel_Name"].split("(")[0]][ctr].append(float(r["Counter_Value"]))
res = {"commit": commit, "command": "bash tools/profile_round.sh " + tag + "  (on the GPU box: rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/prof_scan.py; the same with WRITE_SIZE), then python tools/summarize_traffic.py " + tag + " <out> <commit>",
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh) over tools/prof_scan.py "
               "at BASELINE configs[1] (100k reads x 200 bp, 200 PWMs len 12; dense launches: 20k reads). Per-launch averages. "
               "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 read-side correction).", "kernels": {}}
for k, v in acc.items():
    if "motifs::" not in k:
        continue
    f = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
    w = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    name = k.replace("void ", "").replace("motifs::", "")
    res["kernels"][name] = {"FETCH_SIZE_KB_avg": f, "WRITE_SIZE_KB_avg": w, "launches": len(v["FETCH_SIZE"]),
                            "hbm_bytes_per_launch": (2 * f + w) * 1024}
for short in ("scan_cand_kernel", "stage_hits", "emit_records"):
    for name, v in res["kernels"].items():
        if name.startswith(short):
            res[short] = {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in res["kernels"].items()}, indent=1))
