"""The same launch of G mini-batches N times on live codes (the configs[1] fixture's state): largest deviation of the losses and of the gradient from the
first run.  The engine's float atomics leave ulp-level noise; a race in a kernel's LDS protocol would show as an outlier.
usage: python tools/repeat_step.py [G] [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T  # noqa: E402
from _pkg import load_pkg  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pkg = load_pkg()
mo = T.mo
ctx = pkg._lib.Context(0)
gold = np.load(os.path.join(ROOT, "tests", "golden", "model_cfg2.npz"))
hp = mo.Hyperparam(filter_len=12, M=200)
cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
for n in mo.PARAM_VECS + ["D", "F"]:
    setattr(cdl_o, n, torch.tensor(gold["init_" + n].astype(np.float64)))
cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in gold["warm"]]
cdl = T.to_model(pkg, ctx, hp, 200, cdl_o, arena=int((0.3 * G + 2) * (1 << 30)))
codes = pkg.synth.gen_codes(G * hp.batch_size, 200, 91, n_plant=5, k=12)
l0, g0 = T.gpu_loss_grad(pkg, ctx, cdl, codes, G)
worst_l = worst_g = 0.0
for it in range(N):
    l, g = T.gpu_loss_grad(pkg, ctx, cdl, codes, G)
    worst_l = max(worst_l, float(np.abs(l - l0).max() / np.abs(l0).max()))
    worst_g = max(worst_g, float(np.abs(g.astype(np.float64) - g0).max() / np.abs(g0).max()))
print(f"G={G}: {N} repeats, loss {l0[:2]}, largest deviation from the first run: loss {worst_l:.2e}, gradient {worst_g:.2e} of its largest entry")
assert worst_l < 1e-5 and worst_g < 1e-5
