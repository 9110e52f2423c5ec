import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from _pkg import load_pkg
from test_scan_gpu import dev_scan_hits
pkg = load_pkg(); lib = pkg._lib
ctx = lib.Context(0)
g = np.load("tests/golden/scan_small.npz")
print("K", g["bank"].shape, "lens", g["lens"], "codes", g["codes"].shape, "batch", g["batch"])
for rc in (0, 1):
    h, s = dev_scan_hits(torch, ctx, pkg, g["bank"], g["lens"], g["codes"], rc, int(g["batch"]))
    want = g[f"found_rc{rc}"]; ws = g[f"score_rc{rc}"]
    A = set(map(tuple, h.tolist())); B = set(map(tuple, want.tolist()))
    print("rc", rc, len(h), len(want), "extra", sorted(A - B)[:10], "missing", sorted(B - A)[:10], "dups", len(h) - len(A))
    if len(h) == len(want):
        bad = np.nonzero((h != want).any(axis=1))[0]
        print(" order mismatches", len(bad), bad[:5], h[bad[:3]], want[bad[:3]])
        print(" score mismatches", int((s != ws).sum()))
