import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T
from _pkg import load_pkg
pkg = load_pkg(); mo = T.mo; ctx = pkg._lib.Context(0)
gm = np.load(os.path.join(ROOT, "tests", "golden", "model_cfg2_multi.npz"))
gold = np.load(os.path.join(ROOT, "tests", "golden", "model_cfg2.npz"))
dg = np.load(os.path.join(ROOT, "build", "diag_f.npz"))
hp = mo.Hyperparam(filter_len=12, M=200)
cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
for n in mo.PARAM_VECS + ["D", "F"]:
    setattr(cdl_o, n, torch.tensor(gold["init_" + n].astype(np.float64)))
cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in gold["warm"]]
cdl = T.to_model(pkg, ctx, hp, 200, cdl_o, arena=4 << 30)
for g in range(24):
    loss, flat = T.gpu_loss_grad(pkg, ctx, cdl, gm["codes"][g*6:(g+1)*6], 1)
    got = T.split_grad(cdl, flat)
    w = dg["F%d" % g].astype(np.float64).ravel(); a = got["F"].astype(np.float64)
    d = np.abs(a - w); i = int(d.argmax())
    print("mb %d loss %.3e  F %.2e/%.2e  worst idx %d got %.6e want %.6e  n>1e-5max: %d   D %.2e" % (g, abs(loss[0]-gm["losses"][g])/gm["losses"][g], T.rel_inf(a, w), T.rel_elem(a, w), i, a[i], w[i], int((d > 1e-5*np.abs(w).max()).sum()), T.rel_inf(got["D"], dg["D%d" % g])), flush=True)
