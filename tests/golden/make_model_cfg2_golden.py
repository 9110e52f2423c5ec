"""Generates tests/golden/model_cfg2.npz with the CPU oracle (oracle/model_oracle.py) in float64: ONE mini-batch
(6 reads) at BASELINE configs[1] shape (200 bp, 200 filters of length 12, h=12, K=24, q=32), parameters rounded
to float32 first (what the library holds).  Takes several minutes on 8 cores.  The reference cannot run here (no
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
hp = mo.Hyperparam(filter_len=12, M=200)
rng = np.random.default_rng(20260102)
codes = rng.integers(0, 4, size=(6, 200)).astype(np.uint8)
codes[2, 40:52] = codes[0, 10:22]                        # a shared 12-mer, so that some filter sees structure
cdl = mo.UCDL(hp, rng).to(torch.float64)
for n in mo.PARAM_VECS + ["D", "F"]:                      # round the state to float32
    setattr(cdl, n, getattr(cdl, n).detach().float().double())
warm = np.array([cdl.lambda_sparsity_warmup, cdl.lambda_stepsize_warmup, cdl.omega_stepsize_warmup], dtype=np.float32)
cdl.lambda_sparsity_warmup, cdl.lambda_stepsize_warmup, cdl.omega_stepsize_warmup = [float(x) for x in warm]
out = dict(codes=codes, warm=warm)
for n in mo.PARAM_VECS + ["D", "F"]:
    out["init_" + n] = getattr(cdl, n).detach().numpy().astype(np.float32)
names = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "D", "F",
         "penalty_xyz", "mu"]
t0 = time.time()
val, grads = mo.loss_and_grads(codes, cdl, hp, torch.float64)
print("oracle fwd+bwd: %.1f s, loss %.9g" % (time.time() - t0, val.item()), flush=True)
out["loss0"] = np.float64(val.item())
for n, gr in zip(names, grads):
    a = gr.numpy()
    out["grad0_" + n] = a.astype(np.float32) if a.size > 64 else a
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_cfg2.npz"), **out)
print("saved", flush=True)
