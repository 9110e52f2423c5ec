"""SURVEY 8f-4 at BASELINE configs[1] size: code retrieval over N reads with a randomly initialised model, then the
quantile filter and the triplet enumeration + grouping on the device.  Prints wall times (not part of bench.py: the
records depend on a trained model; this measures the machinery at the record counts the reference would see)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from _pkg import load_pkg  # noqa: E402

pkg = load_pkg()
lib, sy, md, post = pkg._lib, pkg.synth, pkg.model, pkg.post
N, L = int(os.environ.get("N", 100_000)), 200
hp = md.Hyperparam(filter_len=12, M=200)
ctx = lib.Context(0)
cdl = md.ucdl(hp, L, ctx=ctx, seed=3, arena_bytes=40 << 30)
codes = sy.gen_codes(N, L, sy.SEED_BASE + 2, n_plant=5, k=12)
t0 = time.perf_counter()
rec = md.code_retrieval(codes, cdl)
t_ret = time.perf_counter() - t0
n = len(rec)
print(f"code retrieval: {N} reads -> {n} records in {t_ret:.2f} s (host buffers in and out)", flush=True)
dev = torch.from_numpy(rec.view(np.uint8).reshape(n, 12)).cuda()
for p in (0.05, 0.35, 0.75):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, m, thr = post.filter_code_components(ctx, dev, n, p)
    torch.cuda.synchronize()
    t_f = time.perf_counter() - t0
    t0 = time.perf_counter()
    H = post.enumerate_triplets(ctx, out, m, hp.h)
    torch.cuda.synchronize()
    t_e = time.perf_counter() - t0
    print(f"quantile {p}: threshold {thr:.4f}, {m} records kept in {t_f*1e3:.1f} ms; {H['n_triplets']} triplets, "
          f"{len(H['counts'])} keys, max count {int(H['counts'].max()) if len(H['counts']) else 0}, enumerate+group {t_e*1e3:.1f} ms "
          f"(incl. D2H of the grouped result)", flush=True)
    del out, H
