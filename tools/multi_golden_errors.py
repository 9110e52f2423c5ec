#!/usr/bin/env python3
"""Errors of a launch of G mini-batches (configs[1] shape, the state and reads of the fixture) against the float64 oracle's sums in
tests/golden/model_cfg2_multi.npz, per parameter array: largest difference over the largest entry, and element-wise on the entries
above 1e-3 of the largest.  The switches of csrc/ are read once per process, so an A/B is one run per setting:

    python tools/multi_golden_errors.py 24 [64]          (G in {24, 64}: the sums the golden holds)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T  # noqa: E402
from _pkg import load_pkg  # noqa: E402

pkg = load_pkg()
mo = T.mo
ctx = pkg._lib.Context(0)
gm, hp, cdl_o = T.multi_golden_state()
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("MOTIFS_")) or "default"
for G in [int(a) for a in sys.argv[1:]] or [24, 64]:
    cdl = T.to_model(pkg, ctx, hp, 200, cdl_o, arena=int((0.3 * G + 2) * (1 << 30)))
    loss, flat = T.gpu_loss_grad(pkg, ctx, cdl, gm["codes"][: G * hp.batch_size], G)
    want = gm["losses"][:G]
    got = T.split_grad(cdl, flat)
    line = ["G=%d [%s] loss %.2e" % (G, tag, np.abs(loss - want).max() / want.max())]
    for n in T.NAMES:
        w = gm["grad%d_%s" % (G, n)].astype(np.float64)
        line.append("%s %.1e/%.1e" % (n, T.rel_inf(got[n], w), T.rel_elem(got[n], w)))
    print("  ".join(line), flush=True)
    for n in ("D", "F"):               # where the differences sit: quantiles of |got - want| / max|want|, entries past 2e-6 / 1e-5, and the filters they belong to
        w = gm["grad%d_%s" % (G, n)].astype(np.float64)
        e = np.abs(got[n].astype(np.float64).reshape(w.shape) - w) / np.abs(w).max()
        q = np.quantile(e, [0.5, 0.9, 0.99, 0.999, 0.9999])
        ch = np.argwhere(e > 2e-6)
        axis = 0 if n == "D" else 2
        print("    %s: quantiles 50/90/99/99.9/99.99 %% %s  max %.1e  >2e-6: %d  >1e-5: %d  distinct filters among them: %d of %d" % (
            n, " ".join("%.1e" % x for x in q), e.max(), int((e > 2e-6).sum()), int((e > 1e-5).sum()),
            len(set(ch[:, axis].tolist())) if len(ch) else 0, w.shape[axis]), flush=True)
    cdl.model.close()
