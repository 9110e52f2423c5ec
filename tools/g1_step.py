"""The reference's schedule on its own: one optimiser step per 6-read mini-batch (train.jl:40-46) at the bench shape
(200 filters of 12, 200 bp, h=12 K=24 q=32).  Prints ms per step; under `rocprofv3 --kernel-trace --stats` the
per-kernel table of exactly these steps.

  python tools/g1_step.py [--steps 200] [--groups 1] [--null-stream]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

import torch  # noqa: E402
from _pkg import load_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--null-stream", action="store_true")
    a = ap.parse_args()
    pkg = load_pkg()
    lib, md, sy = pkg._lib, pkg.model, pkg.synth
    ctx = lib.Context(0)
    if a.null_stream:
        ctx.set_stream(0)
    hp = md.Hyperparam(filter_len=12, M=200)
    L, G = 200, a.groups
    S = G * hp.batch_size
    cdl = md.ucdl(hp, L, ctx=ctx, seed=1, arena_bytes=int((0.3 * G + 2) * (1 << 30)))
    codes = sy.gen_codes(S, L, 78, n_plant=5, k=12)
    raw = torch.from_numpy(codes).cuda()
    dev = torch.zeros(lib.Context.codes_bytes(S, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, S, L, dev.data_ptr())
    loss = torch.zeros(G, dtype=torch.float32, device="cuda")
    grad = torch.zeros(cdl.model.nP, dtype=torch.float32, device="cuda")
    for _ in range(5):
        cdl.model.dp_train_step_dev(None, dev.data_ptr(), G, G, loss.data_ptr(), grad.data_ptr())
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cdl.model.dp_train_step_dev(None, dev.data_ptr(), G, G, loss.data_ptr(), grad.data_ptr())
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"groups": G, "steps": a.steps, "ms_per_step": dt / a.steps * 1e3, "loss0": float(loss[0].item()),
                      "null_stream": a.null_stream, "graphs": os.environ.get("MOTIFS_NO_GRAPH") is None}))
    cdl.model.close()
    ctx.close()


if __name__ == "__main__":
    main()
