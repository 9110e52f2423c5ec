"""Generates tests/golden/model_cfg1.npz with the CPU oracle (oracle/model_oracle.py) in float64:
one mini-batch at BASELINE configs[0] shape (100 bp, 32 filters of length 8), reference defaults otherwise.
The reference cannot run here (no Julia) and ships no fixtures; see the oracle header."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model_oracle as mo  # noqa: E402

torch.set_num_threads(8)
hp = mo.Hyperparam(filter_len=8, M=32)
rng = np.random.default_rng(20260101)
codes = rng.integers(0, 4, size=(12, 100)).astype(np.uint8)
cdl = mo.UCDL(hp, rng).to(torch.float64)
init = {n: getattr(cdl, n).detach().numpy().copy() for n in mo.PARAM_VECS + ["D", "F"]}
warm = np.array([cdl.lambda_sparsity_warmup, cdl.lambda_stepsize_warmup, cdl.omega_stepsize_warmup])
out = dict(codes=codes, warm=warm, **{"init_" + k: v for k, v in init.items()})
names = ["lambda_sparsity", "kappa_sparsity", "lambda_stepsize", "omega_stepsize", "kappa_stepsize", "D", "F",
         "penalty_xyz", "mu"]
for g in range(2):
    val, grads = mo.loss_and_grads(codes[6 * g:6 * g + 6], cdl, hp, torch.float64)
    v32, g32 = mo.loss_and_grads(codes[6 * g:6 * g + 6], mo.UCDL.__new__(mo.UCDL).__class__ and cdl, hp, torch.float32)
    cdl.to(torch.float64)
    out[f"loss{g}"] = np.float64(val.item())
    out[f"loss{g}_f32"] = np.float64(v32.item())
    for n, gr, gr32 in zip(names, grads, g32):
        out[f"grad{g}_{n}"] = gr.numpy()
        rel = (gr - gr32.double()).abs().max().item() / max(gr.abs().max().item(), 1e-30)
        print(g, n, "f32-vs-f64 rel err", rel)
    print("loss", val.item(), v32.item())
rec = mo.code_retrieval(codes, cdl, hp, torch.float64)
out["codes_rec"] = np.stack([rec["position"].astype(np.int64), rec["fil"].astype(np.int64), rec["seq"].astype(np.int64)], 1)
out["codes_mag"] = rec["mag"].view(np.uint16)
print("code records", len(rec))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_cfg1.npz"), **out)
