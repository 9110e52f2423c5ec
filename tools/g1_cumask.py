"""Does the reference-schedule train step (one optimiser step per 6-read mini-batch: ~264 launches of 2-10 us, each reading what the
one before wrote) run faster when the whole step stays on ONE XCD - one L2 - instead of the eight that a plain stream spreads its
blocks over (every dependent read is then a trip to the Infinity Cache / another XCD's L2)?  The context's stream is replaced by a
stream with a CU mask (hipExtStreamCreateWithCUMask through ctypes).  Two guesses of how mask bits map to XCDs are tried:
  contiguous: bits [32 x, 32 x + 32) = XCD x;   interleaved: bits {i : i % 8 == x} = XCD x.

    MOTIFS_NO_GRAPH=1 python tools/g1_cumask.py [--groups 1] [--steps 200]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from _pkg import load_pkg  # noqa: E402


def masked_stream(hip, bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--groups", type=int, default=1)
    a = ap.parse_args()
    pkg = load_pkg()
    lib, md, sy = pkg._lib, pkg.model, pkg.synth
    hip = C.CDLL("libamdhip64.so")
    variants = {"plain": None,
                "contiguous_1xcd": list(range(32)), "interleaved_1xcd": [i for i in range(256) if i % 8 == 0],
                "contiguous_2xcd": list(range(64)), "interleaved_2xcd": [i for i in range(256) if i % 8 < 2],
                "contiguous_4xcd": list(range(128)), "interleaved_4xcd": [i for i in range(256) if i % 8 < 4]}
    out = {}
    for name, bits in variants.items():
        ctx = lib.Context(0)
        if bits is not None:
            st = masked_stream(hip, bits)
            ctx.set_stream(st.value)
        hp = md.Hyperparam(filter_len=12, M=200)
        L, G = 200, a.groups
        S = G * hp.batch_size
        cdl = md.ucdl(hp, L, ctx=ctx, seed=1, arena_bytes=int((0.3 * G + 2) * (1 << 30)))
        codes = sy.gen_codes(S, L, 78, n_plant=5, k=12)
        raw = torch.from_numpy(codes).cuda()
        dev = torch.zeros(lib.Context.codes_bytes(S, L), dtype=torch.uint8, device="cuda")
        loss = torch.zeros(G, dtype=torch.float32, device="cuda")
        grad = torch.zeros(cdl.model.nP, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, S, L, dev.data_ptr())
        for _ in range(20):
            cdl.model.dp_train_step_dev(None, dev.data_ptr(), G, G, loss.data_ptr(), grad.data_ptr())
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            cdl.model.dp_train_step_dev(None, dev.data_ptr(), G, G, loss.data_ptr(), grad.data_ptr())
        ctx.synchronize()
        out[name] = {"ms_per_step": (time.perf_counter() - t0) / a.steps * 1e3, "loss0": float(loss[0].item())}
        print(name, out[name], flush=True)
        cdl.model.close()
        ctx.close()
    print(json.dumps({"groups": a.groups, "graphs": os.environ.get("MOTIFS_NO_GRAPH") is None, "variants": out}))


if __name__ == "__main__":
    main()
