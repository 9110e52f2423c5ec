"""Generates tests/golden/model_cfg2_multi.npz with the CPU oracle (oracle/model_oracle.py) in float64: the 64 mini-batches (384 reads)
of the launches bench.py's train leg times (24 and 64 mini-batches at BASELINE configs[1] shape: 200 bp, 200 filters of length 12,
h=12, K=24, q=32), on the state of model_cfg2.npz (every code alive, so the dense-image kernels of the step all carry values).
Stored: every mini-batch's loss, and the gradient summed over the first 24 and over all 64 mini-batches (src/train.jl:42-44 takes one
gradient per mini-batch; a launch of G mini-batches returns their sum).

The two data-dependent selections of the forward pass are taken on float32-rounded values (oracle DECISIONS_F32): the reference
computes in Float32 (_0_const.jl:1) and `create_ZY_mask` (model.jl:194-204) keeps the entries `>= median(non-zero ZY)`; with all
453 600 codes of a mini-batch alive inside [0, 4e-4] the two middle values are ~15 float32 ulps apart on average and LESS THAN ONE ulp
apart in about one mini-batch of six, where Float32's `a/2 + b/2` rounds to `a` and the mask keeps one entry more than a float64
median does - a different function (gradient of F off by up to 1.7e-3 of its largest entry on a handful of entries of one filter; the
float32 run of this oracle and the HIP engine show it on the same mini-batches).  `decisions_differ[g]` marks the mini-batches where
the float64-decision run gives another gradient.  What stays undecidable: middle values within rounding noise of a one-ulp gap, which
one float32 implementation sees as adjacent floats and another as two ulps apart - the test allows a few entries of F for them.

The oracle runs its needed-lag filter gradient and its direct syntax sums (NEEDED_LAGS / FAST_SYNTAX: the same sums as the literal
forms, held together by tests/test_oracle_model.py) - about 3.5 s per mini-batch and mode on 8 cores, ~8 minutes in all.
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
differ = np.zeros(G_ALL, dtype=bool)
acc = {n: 0.0 for n in NAMES}
out = dict(codes=codes, g_mid=np.int64(G_MID), seed=np.int64(SEED))
t0 = time.time()


def store(G):
    for n in NAMES:
        a = np.array(acc[n])
        out["grad%d_%s" % (G, n)] = a.astype(np.float32) if a.size > 64 else a


for g in range(G_ALL):
    mo.DECISIONS_F32 = True
    val, grads = mo.loss_and_grads(codes[g * B:(g + 1) * B], cdl, hp, torch.float64)
    mo.DECISIONS_F32 = False
    _, grads64 = mo.loss_and_grads(codes[g * B:(g + 1) * B], cdl, hp, torch.float64)
    losses[g] = val.item()
    for n, gr, gr64 in zip(NAMES, grads, grads64):
        acc[n] = acc[n] + gr.numpy().astype(np.float64)
        differ[g] |= not np.array_equal(gr.numpy(), gr64.numpy())
    if g + 1 == G_MID:
        store(G_MID)
    print("mini-batch %d: loss %.9g%s (%.0f s)" % (g, losses[g], "  float64 decisions differ" if differ[g] else "", time.time() - t0), flush=True)
store(G_ALL)
out["losses"] = losses
out["decisions_differ"] = differ
np.savez_compressed(os.path.join(HERE, "model_cfg2_multi.npz"), **out)
print("saved", flush=True)
