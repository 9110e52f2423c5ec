"""200 optimiser steps of 64 mini-batches (the library's own init and AdaBelief updates, new reads every step): every loss finite, the losses falling;
a soak of the large-step kernels over changing magnitudes.  usage: python tools/soak_train.py [steps] [G]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from _pkg import load_pkg  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pkg = load_pkg()
lib, sy, md = pkg._lib, pkg.synth, pkg.model
hp = md.Hyperparam(filter_len=12, M=200)
L = 200
ctx = lib.Context(0)
cdl = md.ucdl(hp, L, ctx=ctx, seed=1, arena_bytes=int((0.3 * G + 2) * (1 << 30)))
S = G * hp.batch_size
loss = torch.zeros(G, dtype=torch.float32, device="cuda")
grad = torch.zeros(cdl.model.nP, dtype=torch.float32, device="cuda")
dcodes = torch.zeros(lib.Context.codes_bytes(S, L), dtype=torch.uint8, device="cuda")
first = last = None
times = []
for it in range(steps):
    codes = sy.gen_codes(S, L, 1000 + it, n_plant=5, k=12)
    raw = torch.from_numpy(codes).cuda()
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, S, L, dcodes.data_ptr())
    t0 = time.perf_counter()
    cdl.model.loss_grad_dev(dcodes.data_ptr(), G, loss.data_ptr(), grad.data_ptr())
    cdl.model.adabelief_dev(grad.data_ptr(), 1.0 / G)
    ctx.synchronize()
    times.append(time.perf_counter() - t0)
    l = loss.cpu().numpy()
    g = grad.cpu().numpy()
    assert np.isfinite(l).all() and np.isfinite(g).all(), (it, l[:4])
    if first is None:
        first = float(l.mean())
    last = float(l.mean())
    if it % 25 == 0:
        print(f"step {it}: mean loss {last:.4f}  |grad|max {np.abs(g).max():.3e}", flush=True)
print(f"soak ok: {steps} steps of {G} mini-batches, mean loss {first:.3f} -> {last:.3f}; ms per step: steps 5-25 {1e3 * np.median(times[5:25]):.2f} (codes mostly dead), "
      f"last 20 {1e3 * np.median(times[-20:]):.2f} (codes alive)")
assert last < first
