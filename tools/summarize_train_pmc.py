"""Fold the PMC passes of tools/profile_r04.sh over tools/prof_train.py (one 64-mini-batch train step: forward + backward + AdaBelief)
into profiles/<tag>_train_kernels_pmc.{txt,json}: per (kernel, grid) the launches of the step, the time (kernel trace of the FETCH
pass), HBM bytes (2 x FETCH_SIZE + WRITE_SIZE KB: the gfx950 read-side correction of MI355X_MICROARCH.md), the share of wave cycles
spent waiting and the matrix-pipe busy share, plus the step's totals (`step_hbm`).
usage: python tools/summarize_train_pmc.py <tag> [commit]      (reads gpurun_out/<tag>_train_{FETCH_SIZE,WRITE_SIZE,sq1,sq2})"""
import collections
import csv
import json
import sys

tag = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else None


def load(d):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/{tag}_train_{d}/out_counter_collection.csv")):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("motifs::", "")
        acc[(name, int(r["Grid_Size"]) // int(r["Workgroup_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


dur = collections.defaultdict(list)
for r in csv.DictReader(open(f"gpurun_out/{tag}_train_FETCH_SIZE/out_kernel_trace.csv")):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("motifs::", "")
    blocks = 1
    for ax in "XYZ":
        blocks *= max(int(r[f"Grid_Size_{ax}"]) // max(int(r[f"Workgroup_Size_{ax}"]), 1), 1)
    dur[(name, blocks)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
f, w, s1, s2 = load("FETCH_SIZE"), load("WRITE_SIZE"), load("sq1"), load("sq2")
rows = []
for (k, b), v in dur.items():
    g = lambda t, c: sum(t.get((k, b, c), [0.0]))
    rd, wr = g(f, "FETCH_SIZE") * 2 * 1024, g(w, "WRITE_SIZE") * 1024
    rows.append({"kernel": k, "blocks": b, "launches": len(v), "avg_us": sum(v) / len(v), "total_us": sum(v), "hbm_read_bytes": rd, "hbm_write_bytes": wr,
                 "hbm_gbs": (rd + wr) / max(sum(v), 1e-9) / 1e3, "wait_share_of_wave_cycles": g(s1, "SQ_WAIT_ANY") / max(g(s1, "SQ_WAVE_CYCLES"), 1.0),
                 "mfma_busy_share": g(s1, "SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / max(g(s1, "SQ_BUSY_CYCLES") / 32.0, 1.0),
                 "valu_insts": g(s2, "SQ_INSTS_VALU"), "lds_bank_conflict_cycles": g(s2, "SQ_LDS_BANK_CONFLICT")})
rows.sort(key=lambda r: -r["total_us"])
tot_us = sum(r["total_us"] for r in rows)
tot_b = sum(r["hbm_read_bytes"] + r["hbm_write_bytes"] for r in rows)
res = {"commit": commit, "command": "bash tools/profile_r04.sh train  (rocprofv3 --pmc <set> --kernel-trace -- python3 tools/prof_train.py, G=64, one step; separate passes "
                                    "for FETCH_SIZE, WRITE_SIZE and two SQ sets), then python tools/summarize_train_pmc.py " + tag,
       "step": {"kernel_us": tot_us, "launches": sum(r["launches"] for r in rows), "hbm_bytes": tot_b, "hbm_gbs_over_kernel_time": tot_b / tot_us / 1e3,
                "frac_of_8TBs": tot_b / tot_us / 1e3 / 8000.0},
       "kernels": rows}
json.dump(res, open(f"profiles/{tag}_train_kernels_pmc.json", "w"), indent=1)
with open(f"profiles/{tag}_train_kernels_pmc.txt", "w") as fh:
    fh.write(f"one 64-mini-batch train step (384 reads x 200 bp, 200 filters of 12): {tot_us / 1e3:.2f} ms of kernels in {res['step']['launches']} launches, "
             f"{tot_b / 1e9:.1f} GB of HBM traffic = {res['step']['hbm_gbs_over_kernel_time']:.0f} GB/s = {res['step']['frac_of_8TBs']:.2f} of 8 TB/s\n")
    fh.write(f"{'kernel':32s} {'blocks':>7s} {'n':>4s} {'avg us':>8s} {'total us':>9s} {'read MB':>9s} {'write MB':>9s} {'GB/s':>6s} {'wait':>5s} {'mfma':>5s}\n")
    for r in rows[:40]:
        fh.write(f"{r['kernel'][:32]:32s} {r['blocks']:7d} {r['launches']:4d} {r['avg_us']:8.1f} {r['total_us']:9.1f} {r['hbm_read_bytes'] / 1e6:9.1f} "
                 f"{r['hbm_write_bytes'] / 1e6:9.1f} {r['hbm_gbs']:6.0f} {r['wait_share_of_wave_cycles']:5.2f} {r['mfma_busy_share']:5.2f}\n")
print(open(f"profiles/{tag}_train_kernels_pmc.txt").read()[:1500])
