"""Density of the D-layer codes (Z, Y after the last ISTA step's ReLU) and of X at the bench's configs[1] shape and the reference's
init (round-3 verdict, item 3b: below ~25 % the D-layer codes could travel as per-read lists as the syntax layer's already do)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _pkg import load_pkg  # noqa: E402

pkg = load_pkg()
lib, sy, md = pkg._lib, pkg.synth, pkg.model
ctx = lib.Context(0)
for steps in (0, 30):
    hp = md.Hyperparam(filter_len=12, M=200)
    cdl = md.ucdl(hp, 200, ctx=ctx, seed=sy.SEED_BASE + 2, arena_bytes=8 << 30)
    G = 4
    codes = sy.gen_codes(G * 6, 200, 77, n_plant=5, k=12)
    for _ in range(steps):
        cdl.model.train_step(codes, G, want_l1=False)
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(G * 6, 200), dtype=torch.uint8, device="cuda")
    loss = torch.zeros(G, dtype=torch.float32, device="cuda")
    grad = torch.zeros(cdl.model.nP, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, G * 6, 200, dcodes.data_ptr())
    cdl.model.loss_grad_dev(dcodes.data_ptr(), G, loss.data_ptr(), grad.data_ptr(), True)
    ctx.synchronize()
    zy = cdl.model.dump("ZY")
    x = cdl.model.dump("X")
    print(f"after {steps} optimiser steps: ZY {zy.size} entries, {np.count_nonzero(zy) / zy.size:.4f} non-zero; "
          f"X {x.size} entries, {np.count_nonzero(x) / x.size:.4f} non-zero; loss {loss.cpu().numpy()}")
    cdl.model.close()
