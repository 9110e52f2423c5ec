"""Per-launch counter values from rocprofv3 --pmc passes, one line per (kernel, grid size, counter): the same kernel at
another launch size is another line, never folded into one average.
usage: python tools/pmc_summary.py <kernel-name substring> <pass dir> [<pass dir> ...]"""
import collections
import csv
import sys

pat = sys.argv[1]
for d in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d + "/out_counter_collection.csv")):
        if pat in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("motifs::", "")[:40]
            acc[(name, int(r["Grid_Size"]) // int(r["Workgroup_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, blocks, c), v in sorted(acc.items()):
        print("%-40s blocks %8d %-28s %.4g per launch (launches %d, min %.4g max %.4g)" % (k, blocks, c, sum(v) / len(v), len(v), min(v), max(v)))
