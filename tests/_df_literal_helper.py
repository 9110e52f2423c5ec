"""Helper of test_model_gpu.py::test_df_telescoping_equals_literal_sequence: loss and gradient of the configs[1]-shape
fixture in a fresh process (the MOTIFS_DF_LITERAL switch is read once per process).  usage: python _df_literal_helper.py out.npz"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from _pkg import load_pkg  # noqa: E402
import test_model_gpu as T  # noqa: E402

pkg = load_pkg()
mo = T.mo
g = np.load(os.path.join(HERE, "golden", "model_cfg2.npz"))
hp = mo.Hyperparam(filter_len=12, M=200)
cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
for n in mo.PARAM_VECS + ["D", "F"]:
    setattr(cdl_o, n, torch.tensor(g["init_" + n].astype(np.float64)))
cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g["warm"]]
ctx = pkg._lib.Context(0)
cdl = T.to_model(pkg, ctx, hp, 200, cdl_o)
loss, flat = T.gpu_loss_grad(pkg, ctx, cdl, g["codes"], 1)
np.savez(sys.argv[1], loss=loss, flat=flat)
