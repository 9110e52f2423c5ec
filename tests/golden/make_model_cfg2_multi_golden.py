"""Generates tests/golden/model_cfg2_multi.npz with the CPU oracle (oracle/model_oracle.py) in float64: the 64 mini-batches
(384 reads) of the step bench.py times, at BASELINE configs[1] shape (200 bp, 200 filters of length 12, h=12, K=24, q=32), on the
state of model_cfg2.npz (its codes survive the shrinkage).  Stored: every mini-batch's loss, and the gradient summed over the
first 24 and over all 64 mini-batches (src/train.jl:42-44 takes one gradient per mini-batch; a launch of G mini-batches returns
their sum).  The oracle runs its needed-lag filter gradient and its direct syntax sums (NEEDED_LAGS / FAST_SYNTAX:
the same sums as the literal forms, held together by tests/test_oracle_model.py) - about 3.5 s per mini-batch on 8 cores.
The reference cannot run here (no Julia) and ships no fixtures; see the oracle header."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_pkg  # noqa: E402  (synthetic reads only: motifs.jl_amd/synth.py is numpy, no device)
from oracle import model_oracle as mo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
G_ALL, G_MID, SEED = 64, 24, 91
NAMES = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "D", "F", "penalty_xyz", "mu"]

torch.set_num_threads(8)
mo.NEEDED_LAGS = True
mo.FAST_SYNTAX = True
hp = mo.Hyperparam(filter_len=12, M=200)
gold = np.load(os.path.join(HERE, "model_cfg2.npz"))
cdl = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
for n in mo.PARAM_VECS + ["D", "F"]:
    setattr(cdl, n, torch.tensor(gold["init_" + n].astype(np.float64)))
cdl.lambda_sparsity_warmup, cdl.lambda_stepsize_warmup, cdl.omega_stepsize_warmup = [float(x) for x in gold["warm"]]
codes = load_pkg().synth.gen_codes(G_ALL * hp.batch_size, 200, SEED, n_plant=5, k=12)
B = hp.batch_size
losses = np.zeros(G_ALL, dtype=np.float64)
acc = {n: 0.0 for n in NAMES}
out = dict(codes=codes, g_mid=np.int64(G_MID), seed=np.int64(SEED))
t0 = time.time()
for g in range(G_ALL):
    val, grads = mo.loss_and_grads(codes[g * B:(g + 1) * B], cdl, hp, torch.float64)
    losses[g] = val.item()
    for n, gr in zip(NAMES, grads):
        acc[n] = acc[n] + gr.numpy().astype(np.float64)
    if g + 1 == G_MID:
        for n in NAMES:
            a = np.array(acc[n])
            out["grad%d_%s" % (G_MID, n)] = a.astype(np.float32) if a.size > 64 else a
    print("mini-batch %d: loss %.9g (%.0f s)" % (g, losses[g], time.time() - t0), flush=True)
for n in NAMES:
    a = np.array(acc[n])
    out["grad%d_%s" % (G_ALL, n)] = a.astype(np.float32) if a.size > 64 else a
out["losses"] = losses
np.savez_compressed(os.path.join(HERE, "model_cfg2_multi.npz"), **out)
print("saved", flush=True)
