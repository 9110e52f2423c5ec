"""Generates tests/golden/model_cfg3.npz with the CPU oracle (oracle/model_oracle.py) in float64: ONE mini-batch
(6 reads) at BASELINE configs[3] shape (500 bp, 512 filters of length 20, h=12, K=24, q=32), parameters rounded to
float32 first (what the library holds).  update_D uses the oracle's needed-lag form and the syntax-layer synthesis /
filter gradient their direct forms (mo.NEEDED_LAGS, mo.FAST_SYNTAX; equal to the literal forms: tests/test_oracle_model.py::test_needed_lag_update_D_equals_the_literal_one) - the literal one
would take hours here.  Only the loss, the gradients of the seven vectors and of D, and a strided sample + checksums of
the gradient of F are stored, to keep the fixture small.  The reference cannot run here (no Julia) and ships no
fixtures; see the oracle header."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model_oracle as mo  # noqa: E402

torch.set_num_threads(8)
mo.NEEDED_LAGS = mo.FAST_SYNTAX = True
hp = mo.Hyperparam(filter_len=20, M=512)
L = 500
rng = np.random.default_rng(20260104)
codes = rng.integers(0, 4, size=(6, L)).astype(np.uint8)
codes[2, 140:160] = codes[0, 30:50]                      # a shared 20-mer, so that some filter sees structure
codes[4, 300:320] = codes[0, 30:50]
cdl = mo.UCDL(hp, rng).to(torch.float64)
# With ucdl(hp)'s own scale (0.05 * rand) 512 filters overshoot at this shape: every code is zero after the first pass and
# all gradients vanish - a fixture that tests nothing.  A tenth of it keeps the codes alive through the six passes
# (loss 342 < 500 = the loss of an all-zero code).
SCALE = 0.1
for n in mo.PARAM_VECS:
    setattr(cdl, n, getattr(cdl, n) * SCALE)
cdl.lambda_stepsize_warmup *= SCALE
cdl.omega_stepsize_warmup *= SCALE
for n in mo.PARAM_VECS + ["D", "F"]:                      # round the state to float32
    setattr(cdl, n, getattr(cdl, n).detach().float().double())
warm = np.array([cdl.lambda_sparsity_warmup, cdl.lambda_stepsize_warmup, cdl.omega_stepsize_warmup], dtype=np.float32)
cdl.lambda_sparsity_warmup, cdl.lambda_stepsize_warmup, cdl.omega_stepsize_warmup = [float(x) for x in warm]
seed_state = dict(codes=codes, warm=warm)
out = dict(seed_state)
for n in mo.PARAM_VECS:
    out["init_" + n] = getattr(cdl, n).detach().numpy().astype(np.float32)
out["init_D"] = cdl.D.detach().numpy().astype(np.float32)
out["init_F"] = cdl.F.detach().numpy().astype(np.float16)   # |F| < 1, 0.1*randn: float16 keeps the fixture small; the test loads exactly this
cdl.F = torch.tensor(out["init_F"].astype(np.float64))
names = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "D", "F",
         "penalty_xyz", "mu"]
t0 = time.time()
val, grads = mo.loss_and_grads(codes, cdl, hp, torch.float64)
print("oracle fwd+bwd: %.1f s, loss %.9g" % (time.time() - t0, val.item()), flush=True)
out["loss0"] = np.float64(val.item())
for n, gr in zip(names, grads):
    a = gr.numpy()
    if n == "F":
        flat = a.reshape(-1)
        out["grad0_F_sample_stride"] = np.int64(7)
        out["grad0_F_sample"] = flat[::7].astype(np.float32)
        out["grad0_F_absmax"] = np.float64(np.abs(flat).max())
        out["grad0_F_sum"] = np.float64(flat.sum())
        out["grad0_F_sumsq"] = np.float64((flat * flat).sum())
    else:
        out["grad0_" + n] = a.astype(np.float32) if a.size > 64 else a
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_cfg3.npz"), **out)
print("saved", flush=True)
