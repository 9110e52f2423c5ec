"""Debug: cfg-2 golden inputs through the engine; dump intermediates to gpurun_out/dbg_<tag>.npz."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _pkg import load_pkg
pkg = load_pkg()
import test_model_gpu as T
mo = T.mo
g = np.load(os.path.join(os.path.dirname(T.__file__), "golden", "model_cfg2.npz"))
hp = mo.Hyperparam(filter_len=12, M=200)
cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
for n in mo.PARAM_VECS + ["D", "F"]:
    setattr(cdl_o, n, torch.tensor(g["init_" + n].astype(np.float64)))
cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g["warm"]]
ctx = pkg._lib.Context(0)
cdl = T.to_model(pkg, ctx, hp, 200, cdl_o)
loss, flat = T.gpu_loss_grad(pkg, ctx, cdl, g["codes"], 1, keep=True)
out = {"loss": loss, "flat": flat}
for n in ("X0", "ZY0", "X", "ZY"):
    try:
        out[n] = cdl.model.dump(n)
    except Exception as e:
        print("no", n, e)
np.savez("gpurun_out/dbg_%s.npz" % sys.argv[1], **out)
got = T.split_grad(cdl, flat)
for n in T.NAMES:
    print(n, T.rel_inf(got[n], g[f"grad0_{n}"].astype(np.float64)))
print("loss", loss, g["loss0"])
