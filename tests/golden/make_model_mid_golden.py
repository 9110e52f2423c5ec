"""Generates tests/golden/model_mid.npz with the CPU oracle (oracle/model_oracle.py) in float64: two mini-batches of
3 reads at three mid-size shapes that take other template instances of the LDS-resident matrix-core kernels than
configs[0]/[1] (window heights 8 and 12, channel chunks of 32 / 80, odd K, 40- and 48-wide row GEMMs); parameters
rounded to float32 first (what the library holds).  A few minutes on 8 cores.  The reference cannot run here (no
Julia) and ships no fixtures; see the oracle header."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model_oracle as mo  # noqa: E402

torch.set_num_threads(8)
SHAPES = [(10, 48, 8, 17, 10, 60), (12, 40, 12, 24, 8, 64), (9, 64, 8, 12, 12, 72)]   # filter_len, M, h, K, q, bp
NAMES = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "D", "F", "penalty_xyz", "mu"]
G, B = 2, 3
out = {"shapes": np.array(SHAPES, dtype=np.int64)}
for i, (fl, M, h, K, q, bp) in enumerate(SHAPES):
    hp = mo.Hyperparam(filter_len=fl, M=M, h=h, K=K, q=q, batch_size=B, num_pass_xyz=2, num_pass_df=2)
    rng = np.random.default_rng(fl * 1000 + M)
    codes = rng.integers(0, 4, size=(G * B, bp)).astype(np.uint8)
    cdl = mo.UCDL(hp, rng).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl, n, getattr(cdl, n).detach().float().double())
    warm = np.array([cdl.lambda_sparsity_warmup, cdl.lambda_stepsize_warmup, cdl.omega_stepsize_warmup], dtype=np.float32)
    cdl.lambda_sparsity_warmup, cdl.lambda_stepsize_warmup, cdl.omega_stepsize_warmup = [float(x) for x in warm]
    out[f"s{i}_codes"], out[f"s{i}_warm"] = codes, warm
    for n in mo.PARAM_VECS + ["D", "F"]:
        out[f"s{i}_init_{n}"] = getattr(cdl, n).detach().numpy().astype(np.float32)
    t0 = time.time()
    tot = {n: 0.0 for n in NAMES}
    for g in range(G):
        val, grads = mo.loss_and_grads(codes[g * B:(g + 1) * B], cdl, hp, torch.float64)
        out[f"s{i}_loss{g}"] = np.float64(val.item())
        for n, gr in zip(NAMES, grads):
            tot[n] = tot[n] + gr.numpy()
    for n in NAMES:                                      # the library returns the SUM over the mini-batches of a call
        out[f"s{i}_grad_{n}"] = tot[n].astype(np.float32) if tot[n].size > 64 else tot[n]
    print("shape %d: %.1f s" % (i, time.time() - t0), flush=True)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_mid.npz"), **out)
print("saved", flush=True)
