#!/usr/bin/env python3
"""Times the hit-record scan at the BASELINE configs[3] / configs[4] shard shapes (both strands, ordered records, HBM-resident)
for each chunk-group setting (MOTIFS_CG_CHUNKS: 0 = the table gathered from L2 as in round 3, 1 / 2 / 4 = chunks of 128 PWMs
per group, auto), one context per setting in ONE process so that the variants share a box and its clock state.

    python tools/shard_time.py --cfg 4 [--n 25000] [--ws-gib 4] [--modes auto,0,1,2,4] [--reps 3]

Per variant: ms per step, the HIP-event time of its three stages (candidate kernel / stage_hits + row scan / emit_records),
the plan the library chose, and a checksum of the records (every variant must give the same one)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=4)
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--ws-gib", type=float, default=-1)
    ap.add_argument("--modes", default="auto,0,2,1,4")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch

    from _pkg import load_pkg

    pkg = load_pkg()
    lib, sy = pkg._lib, pkg.synth
    if args.cfg == 3:
        n, L, K, lo, hi, ws = args.n or 62_500, 500, 512, 20, 20, 0
    else:
        n, L, K, lo, hi, ws = args.n or 25_000, 1000, 2048, 8, 20, 4 << 30
    if args.ws_gib >= 0:
        ws = int(args.ws_gib * (1 << 30))
    seed = sy.SEED_BASE + 2
    pw, ln = sy.gen_pwm_bank(K, seed + 7, len_lo=lo, len_hi=hi, alpha=0.3)
    bk = sy.pad_bank(pw, ln)
    cd = sy.gen_codes(n, L, seed + 31, n_plant=5, k=hi)
    dev = torch.device("cuda", 0)
    raw = torch.from_numpy(cd).to(dev)
    dc = torch.zeros(lib.Context.codes_bytes(n, L), dtype=torch.uint8, device=dev)
    out = {"cfg": args.cfg, "seqs": n, "seq_len": L, "pwms": K, "workspace_limit_bytes": ws, "variants": {}}
    bufs = None
    for mode in args.modes.split(","):
        if mode == "auto":
            os.environ.pop("MOTIFS_CG_CHUNKS", None)
        else:
            os.environ["MOTIFS_CG_CHUNKS"] = mode
        ctx = lib.Context(0)
        os.environ.pop("MOTIFS_CG_CHUNKS", None)
        ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, n, L, dc.data_ptr())
        ctx.set_workspace_limit(ws)
        need = ctx.pwm_scan_hits_both_dev(bk, ln, dc.data_ptr(), n, L, None, None, 0)
        if bufs is None:
            cap = int(max(need)) + 1024
            bufs = ([torch.empty((cap, 3), dtype=torch.int32, device=dev) for _ in range(2)],
                    [torch.empty(cap, dtype=torch.int16, device=dev) for _ in range(2)], torch.zeros((2, K), dtype=torch.int64, device=dev), cap)
        h, s, k, cap = bufs

        def one():
            return ctx.pwm_scan_hits_both_dev(bk, ln, dc.data_ptr(), n, L, [x.data_ptr() for x in h], [x.data_ptr() for x in s], cap,
                                              counts_ptr=k.data_ptr())
        for _ in range(2):
            got = one()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            got = one()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        ctx.enable_timing(True)
        ctx.reset_timing()
        for _ in range(args.reps):
            one()
        ctx.synchronize()
        km = {nm: ctx.kernel_ms(sl) for nm, sl in (("cand", lib.KS_SCAN_COUNT), ("stage", lib.KS_SCAN_OFFSETS), ("emit", lib.KS_SCAN_FILL))}
        ctx.enable_timing(False)
        assert tuple(got) == tuple(need)
        chk = [int(h[r][: got[r]].to(torch.int64).sum().item()) * 31 + int(s[r][: got[r]].to(torch.int64).sum().item()) for r in (0, 1)]
        # order-sensitive: a weighted sum over a stride of the records
        idx = torch.arange(0, got[0], 997, device=dev)
        chk.append(int((h[0][idx].to(torch.int64).sum(dim=1) * (idx % 1013 + 1)).sum().item()))
        out["variants"][mode] = {"plan": ctx.scan_plan(), "ms_per_step": dt * 1e3, "bases_per_s": n * L / dt, "hits": int(sum(got)),
                                 "stage_ms_per_step": {nm: v[0] / args.reps for nm, v in km.items()},
                                 "launches_per_step": {nm: v[1] / args.reps for nm, v in km.items()}, "checksum": chk}
        print(mode, json.dumps(out["variants"][mode]), flush=True)
        ctx.close()
    sums = {json.dumps(v["checksum"]) for v in out["variants"].values()}
    out["all_variants_agree"] = len(sums) == 1
    print(json.dumps(out))
    assert out["all_variants_agree"], "variants disagree"


if __name__ == "__main__":
    main()
