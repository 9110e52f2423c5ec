"""Train-step probe at BASELINE configs[1] shape: time and arena use of motifs_model_loss_grad_dev."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from _pkg import load_pkg  # noqa: E402

pkg = load_pkg()
lib, sy, md = pkg._lib, pkg.synth, pkg.model
L, M, fl = int(os.environ.get("L", 200)), int(os.environ.get("M", 200)), int(os.environ.get("FL", 12))
G = int(os.environ.get("G", 8))
reps = int(os.environ.get("REPS", 3))
hp = md.Hyperparam(filter_len=fl, M=M)
ctx = lib.Context(0)
cdl = md.ucdl(hp, L, ctx=ctx, seed=1, arena_bytes=int(float(os.environ.get("ARENA_GB", 16)) * (1 << 30)))
S = G * hp.batch_size
codes = sy.gen_codes(S, L, 5, n_plant=5, k=fl)
raw = torch.from_numpy(codes).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(S, L), dtype=torch.uint8, device="cuda")
loss = torch.zeros(G, dtype=torch.float32, device="cuda")
grad = torch.zeros(cdl.model.nP, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, S, L, dcodes.data_ptr())
for it in range(reps + 1):
    ctx.synchronize()
    t0 = time.perf_counter()
    cdl.model.loss_grad_dev(dcodes.data_ptr(), G, loss.data_ptr(), grad.data_ptr())
    cdl.model.adabelief_dev(grad.data_ptr(), 1.0 / G)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"G={G} S={S} step {it}: {dt*1e3:.1f} ms  ({S/dt:.0f} seq/s)  loss {loss.cpu().numpy()[:3]}", flush=True)
