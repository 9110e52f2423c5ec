#!/usr/bin/env python3
"""Achieved parity errors of the HIP sparse-coding engine against the float64 oracle, every golden in one GPU run
(round-3 verdict, item 7): loss, per-array gradient (largest difference over the largest entry; largest element-wise relative
difference over the entries above 1e-3 of the largest), the ZY / X intermediates of the tiny shapes, and how many code-record
magnitudes differ by one binary16 ulp.  Writes profiles/r04_parity_errors.json (+ a markdown table on stdout); the
tolerances of tests/test_model_gpu.py are set from it.

    python tools/parity_errors.py [--runs 3]      (several runs: the engine's float atomics leave ulp-level noise)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stats(got, want):
    got, want = np.asarray(got, np.float64).ravel(), np.asarray(want, np.float64).ravel()
    mx = max(np.abs(want).max(), 1e-300)
    big = np.abs(want) > 1e-3 * mx
    elem = float((np.abs(got - want)[big] / np.abs(want)[big]).max()) if big.any() else 0.0
    return {"inf_over_max": float(np.abs(got - want).max() / mx), "elem_rel_big": elem, "n": int(got.size), "n_big": int(big.sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--name", default="r04_parity_errors", help="profiles/<name>.json (e.g. a second table with the MOTIFS_*_F16_MIN_* switches at 1: "
                    "every golden through the binary16 GEMMs)")
    args = ap.parse_args()
    import torch

    import test_model_gpu as T
    from _pkg import load_pkg
    from oracle import model_oracle as mo

    pkg = load_pkg()
    ctx = pkg._lib.Context(0)
    G = os.path.join(ROOT, "tests", "golden")
    out = {}

    def merge(name, rec):
        cur = out.setdefault(name, {})
        for k, v in rec.items():
            if isinstance(v, dict):
                c2 = cur.setdefault(k, {})
                for kk, vv in v.items():
                    c2[kk] = max(c2.get(kk, 0), vv) if isinstance(vv, float) else vv
            else:
                cur[k] = max(cur.get(k, 0), v) if isinstance(v, float) else v

    def from_golden(name, g, hp, L, n_groups, prefix="", arena=1 << 30, want_grad=None, f_sample=False, codes_key="codes"):
        cdl_o = mo.UCDL(hp, np.random.default_rng(0)).to(torch.float64)
        for n in mo.PARAM_VECS + ["D", "F"]:
            setattr(cdl_o, n, torch.tensor(g[prefix + "init_" + n].astype(np.float64)))
        cdl_o.lambda_sparsity_warmup, cdl_o.lambda_stepsize_warmup, cdl_o.omega_stepsize_warmup = [float(x) for x in g[prefix + "warm"]]
        cdl = T.to_model(pkg, ctx, hp, L, cdl_o, arena=arena)
        try:
            loss, flat = T.gpu_loss_grad(pkg, ctx, cdl, g[prefix + codes_key], n_groups)
            got = T.split_grad(cdl, flat)
            rec = {"loss_rel": float(max(abs(loss[k] - g[f"{prefix}loss{k}"]) / abs(g[f"{prefix}loss{k}"]) for k in range(n_groups)))}
            for n in T.NAMES:
                if f_sample and n == "F":
                    gf = got["F"].astype(np.float64)
                    st = int(g["grad0_F_sample_stride"])
                    rec["grad_F(sampled)"] = {"inf_over_max": float(np.abs(gf[::st] - g["grad0_F_sample"].astype(np.float64)).max() / g["grad0_F_absmax"])}
                    continue
                rec["grad_" + n] = stats(got[n], want_grad(n))
            return rec, cdl
        except Exception:
            cdl.model.close()
            raise

    for run in range(args.runs):
        # tiny shapes: loss, gradients (sum over two mini-batches), ZY and X intermediates
        for seed in (0, 1, 2):
            hp, codes, cdl_o = T.tiny(seed)
            Gn, B = 2, hp.batch_size
            cdl = T.to_model(pkg, ctx, hp, codes.shape[1], cdl_o)
            loss, flat = T.gpu_loss_grad(pkg, ctx, cdl, codes, Gn, keep=True)
            got = T.split_grad(cdl, flat)
            ln = mo.LengthInfo.make(hp, codes.shape[1])
            projs = mo.Projectors(hp, ln, torch.float64)
            want = {n: 0.0 for n in T.NAMES}
            rec = {"loss_rel": 0.0, "ZY": {"inf_over_max": 0.0}, "X": {"inf_over_max": 0.0}}
            for g_ in range(Gn):
                val, grads = mo.loss_and_grads(codes[g_ * B:(g_ + 1) * B], cdl_o, hp, torch.float64)
                rec["loss_rel"] = float(max(rec["loss_rel"], abs(loss[g_] - val.item()) / abs(val.item())))
                for n, gr in zip(T.NAMES, grads):
                    want[n] = want[n] + gr.numpy()
                S = mo.onehot_batch(codes[g_ * B:(g_ + 1) * B], torch.float64)
                with torch.no_grad():
                    _, Z, Y, X = mo.retrieve_code(S, cdl_o.to(torch.float64), hp, ln, projs)
                zy = torch.cat((Z[..., 0::4], Y[..., 0::4]), dim=1).permute(0, 2, 1).numpy()
                rec["ZY"]["inf_over_max"] = max(rec["ZY"]["inf_over_max"], T.rel_inf(cdl.model.dump("ZY").reshape(Gn, B, ln.c, hp.twoM)[g_], zy))
                rec["X"]["inf_over_max"] = max(rec["X"]["inf_over_max"],
                                               T.rel_inf(cdl.model.dump("X").reshape(Gn, B, ln.l, hp.K)[g_], X[:, :, 0, :].permute(0, 2, 1).numpy()))
            for n in T.NAMES:
                rec["grad_" + n] = stats(got[n], want[n])
            cdl.model.close()
            merge(f"tiny seed {seed}", rec)
        gm = np.load(os.path.join(G, "model_mid.npz"))
        for i in range(3):
            fl, M, h, K, q, bp = [int(x) for x in gm["shapes"][i]]
            hp = mo.Hyperparam(filter_len=fl, M=M, h=h, K=K, q=q, batch_size=3, num_pass_xyz=2, num_pass_df=2)
            rec, cdl = from_golden(f"mid {i}", gm, hp, bp, 2, prefix=f"s{i}_", want_grad=lambda n, i=i: gm[f"s{i}_grad_{n}"])
            cdl.model.close()
            merge(f"mid {i}: fl {fl} M {M} h {h} K {K} {bp} bp", rec)
        g1 = np.load(os.path.join(G, "model_cfg1.npz"))
        hp = mo.Hyperparam(filter_len=8, M=32)
        rec, cdl = from_golden("cfg0", g1, hp, 100, 2, want_grad=lambda n: g1[f"grad0_{n}"] + g1[f"grad1_{n}"])
        recs = pkg.model.code_retrieval(g1["codes"], cdl)
        d = np.abs(recs["mag"].view(np.uint16).astype(np.int64) - g1["codes_mag"].astype(np.int64))
        rec["code_records"] = int(len(d))
        rec["code_mag_off_by_one_ulp"] = int((d == 1).sum())
        rec["code_mag_off_by_more"] = int((d > 1).sum())
        rec["code_indices_exact"] = bool(np.array_equal(np.stack([recs["position"], recs["fil"], recs["seq"]], 1).astype(np.int64), g1["codes_rec"]))
        cdl.model.close()
        merge("configs[0] shape (100 bp, 32 filters of 8), 2 mini-batches", rec)
        g2 = np.load(os.path.join(G, "model_cfg2.npz"))
        rec, cdl = from_golden("cfg1", g2, mo.Hyperparam(filter_len=12, M=200), 200, 1, want_grad=lambda n: g2[f"grad0_{n}"])
        cdl.model.close()
        merge("configs[1] shape (200 bp, 200 filters of 12), 1 mini-batch", rec)
        g3 = np.load(os.path.join(G, "model_cfg3.npz"))
        rec, cdl = from_golden("cfg3", g3, mo.Hyperparam(filter_len=20, M=512), 500, 1, arena=16 << 30, want_grad=lambda n: g3[f"grad0_{n}"], f_sample=True)
        cdl.model.close()
        merge("configs[3] shape (500 bp, 512 filters of 20), 1 mini-batch", rec)
    ctx.close()
    res = {"runs": args.runs, "switches": {k: v for k, v in os.environ.items() if k.startswith("MOTIFS_")}, "what": "largest error over the runs; gradients: |got - want|_inf / |want|_inf and the largest element-wise relative error over the "
                                      "entries above 1e-3 of the largest; float64 oracle (oracle/model_oracle.py)", "cases": out}
    for d in ("profiles", "gpurun_out"):       # (gpurun_out/ is what travels back from the GPU box)
        if os.path.isdir(os.path.join(ROOT, d)):
            with open(os.path.join(ROOT, d, args.name + ".json"), "w") as fh:
                json.dump(res, fh, indent=1, default=float)
    print("| case | loss rel | worst gradient, inf / max | worst gradient, element-wise (entries > 1e-3 max) | other |")
    print("|---|---|---|---|---|")
    for name, rec in out.items():
        gi = max((v["inf_over_max"], k) for k, v in rec.items() if k.startswith("grad_"))
        ge = max((v.get("elem_rel_big", 0.0), k) for k, v in rec.items() if k.startswith("grad_"))
        other = []
        for k in ("ZY", "X"):
            if k in rec:
                other.append(f"{k} {rec[k]['inf_over_max']:.1e}")
        if "code_records" in rec:
            other.append(f"{rec['code_mag_off_by_one_ulp']} of {rec['code_records']} code magnitudes off by one binary16 ulp, {rec['code_mag_off_by_more']} by more")
        print(f"| {name} | {rec['loss_rel']:.1e} | {gi[0]:.1e} ({gi[1][5:]}) | {ge[0]:.1e} ({ge[1][5:]}) | {'; '.join(other)} |")


if __name__ == "__main__":
    main()
