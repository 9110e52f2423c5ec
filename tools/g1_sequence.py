"""The launches of ONE step of tools/g1_step.py in stream order, from `rocprofv3 --kernel-trace --output-format csv`: name, grid, duration and the
gap to the previous kernel's end - where a small step's time goes (kernels against the spaces between them).
usage: python tools/g1_sequence.py <kernel_trace.csv> [which_step_from_the_end=2]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adabelief" in r["Kernel_Name"].lower()]
    sel = rows[idx[-back - 1] + 1:idx[-back] + 1]
    prev_end, t0 = None, int(sel[0]["Start_Timestamp"])
    ksum = gsum = 0.0
    for n, r in enumerate(sel):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("motifs::", "")[:40]
        print(f"{n:3d} {(s - t0) / 1e3:8.1f} us  gap {gap:5.1f}  dur {(e - s) / 1e3:6.1f}  {name:40s} {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{int(r['Grid_Size_Y']) // int(r['Workgroup_Size_Y'])}x{int(r['Grid_Size_Z']) // int(r['Workgroup_Size_Z'])} thr {r['Workgroup_Size_X']}")
        ksum += (e - s) / 1e3
        gsum += max(gap, 0.0)
        prev_end = e
    print(f"launches {len(sel)}  kernel time {ksum:.0f} us  gaps {gsum:.0f} us  span {(prev_end - t0) / 1e3:.0f} us")


if __name__ == "__main__":
    main()
