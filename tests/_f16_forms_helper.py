"""Helper of test_model_gpu.py::test_binary16_gemms_on_other_shapes: loss and gradient of the three mid-shape fixtures and of the configs[0]-shape
fixture in a fresh process (the MOTIFS_*_F16_MIN_* switches are read once per process).  usage: python _f16_forms_helper.py out.npz"""
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
ctx = pkg._lib.Context(0)
out = {}


def run(tag, g, prefix, hp, bp, codes, n_groups):
    cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
    for n in mo.PARAM_VECS + ["D", "F"]:
        setattr(cdl_o, n, torch.tensor(g[prefix + "init_" + n].astype(np.float64)))
    cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g[prefix + "warm"]]
    cdl = T.to_model(pkg, ctx, hp, bp, cdl_o)
    loss, flat = T.gpu_loss_grad(pkg, ctx, cdl, codes, n_groups)
    out[tag + "_loss"], out[tag + "_flat"] = loss, flat
    cdl.model.close()


gm = np.load(os.path.join(HERE, "golden", "model_mid.npz"))
for i in range(3):
    fl, M, h, K, q, bp = [int(x) for x in gm["shapes"][i]]
    hp = mo.Hyperparam(filter_len=fl, M=M, h=h, K=K, q=q, batch_size=3, num_pass_xyz=2, num_pass_df=2)
    run(f"mid{i}", gm, f"s{i}_", hp, bp, gm[f"s{i}_codes"], 2)
g1 = np.load(os.path.join(HERE, "golden", "model_cfg1.npz"))
run("cfg0", g1, "", mo.Hyperparam(filter_len=8, M=32), 100, g1["codes"], 2)
np.savez(sys.argv[1], **out)
