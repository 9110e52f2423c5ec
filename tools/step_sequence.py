"""The last launches of a `rocprofv3 --kernel-trace --output-format csv` in stream order: start (relative), gap to the previous kernel's end, duration, name, grid -
where a short step's time goes (e.g. N=12500 ASYNC=1 COUNTS=1 WSTREAM=1 python tools/scan_step_time.py under the profiler: a rank's share of configs[2]).
usage: python tools/step_sequence.py <kernel_trace.csv> [how_many=24]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-n:]
t0, prev = int(sel[0]["Start_Timestamp"]), None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("motifs::", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} us  gap {((s - prev) / 1e3 if prev else 0):6.1f}  dur {(e - s) / 1e3:7.1f}  {name:44s} "
          f"{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{int(r['Grid_Size_Y']) // int(r['Workgroup_Size_Y'])}")
    prev = e
