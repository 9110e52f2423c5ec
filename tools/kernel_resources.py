#!/usr/bin/env python3
"""Per-kernel VGPR / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py motifs.jl_amd/csrc/scan_mfma.hip [filter-regex]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import load_build  # noqa: E402

b = load_build()
src = sys.argv[1]
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + b.COMMON + b.EXTRA.get(os.path.basename(src), []) + [
    "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key.split(" ")[0]] = int(m.group(1))
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'occ':>4s}")
for r in rows:
    if flt and not flt.search(r["name"]):
        continue
    print(f"{r['name'][:70]:70s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('SGPRs', 0):5d} {r.get('ScratchSize', 0):7d} {r.get('Occupancy', 0):4d}")
