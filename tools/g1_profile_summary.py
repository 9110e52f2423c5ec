"""Per-step kernel table of tools/g1_step.py run under `rocprofv3 --kernel-trace --output-format csv`: the launches between the last AdaBelief
kernels, grouped by (kernel, grid).  usage: python tools/g1_profile_summary.py <kernel_trace.csv> [steps=40] [top=50]"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    rows = list(csv.DictReader(open(path)))
    idx = [i for i, r in enumerate(rows) if "adabelief" in r["Kernel_Name"].lower()]
    sel = rows[idx[-steps - 1] + 1:idx[-1] + 1]
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in sel:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("motifs::", "")[:44]
        key = (name, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_X"]))
        tot[key][0] += 1
        tot[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"launches per step {len(sel) / steps:.0f}, kernel time per step {sum(v[1] for v in tot.values()) / steps:.0f} us (under the profiler)")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k[0]:44s} blocks {k[1]:5d}x{k[2]:<3d} thr {k[3]:4d}  n/step {v[0] / steps:5.1f}  avg {v[1] / v[0]:6.1f} us  per step {v[1] / steps:7.1f} us")


if __name__ == "__main__":
    main()
