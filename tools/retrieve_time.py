"""Wall time of code retrieval (a16, _1_code_retrieval.jl:33-56) at BASELINE configs[1] shape, by arena size (= mini-batches per
launch) and input kind."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from _pkg import load_pkg
pkg = load_pkg(); lib, sy, md = pkg._lib, pkg.synth, pkg.model
N, L = int(os.environ.get("N", 30000)), 200
hp = md.Hyperparam(filter_len=12, M=200)
ctx = lib.Context(0)
codes = sy.gen_codes(N, L, sy.SEED_BASE + 2, n_plant=5, k=12)
onehot = sy.codes_to_onehot(codes)
for arena_gb in (8, 40, 120):
    cdl = md.ucdl(hp, L, ctx=ctx, seed=3, arena_bytes=arena_gb << 30)
    for kind, data in ((lib.DATA_CODES_U8, codes), (lib.DATA_ONEHOT_F32, onehot)):
        cdl.model.retrieve_codes(data, kind, N)
        t0 = time.perf_counter()
        rec = cdl.model.retrieve_codes(data, kind, N)
        dt = time.perf_counter() - t0
        print(f"arena {arena_gb} GiB kind {kind}: {N} reads -> {len(rec)} records in {dt:.3f} s = {N / dt:.0f} reads/s", flush=True)
    cdl.model.close()
