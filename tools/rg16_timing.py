"""per-turn timestamps of k_rowgemm16 (build with -DRG16_TIMING, MOTIFS_HIP_LIB=build/exp/libmotifs_timing.so)"""
import ctypes, os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("G", "64"); os.environ.setdefault("ARENA_GB", "24"); os.environ.setdefault("REPS", "1")
exec(open(os.path.join(ROOT, "tools", "prof_train.py")).read())
h = ctypes.CDLL(os.environ["MOTIFS_HIP_LIB"])
buf = np.zeros((256, 2, 16, 5), dtype=np.uint64)
print("rc", h.motifs_debug_rg16_ts(buf.ctypes.data_as(ctypes.c_void_p)))
for b in (0, 100, 255):
    t0 = int(buf[b, 0, 0, 0])
    print("block", b)
    for role in (0, 1):
        for j in range(11):
            r = buf[b, role, j].astype(np.int64) - t0
            print(" role %d turn %2d: " % (role, j) + " ".join("%7d" % x for x in r[:4]) + "   | seg: " + " ".join("%6d" % (r[k + 1] - r[k]) for k in range(3)))
