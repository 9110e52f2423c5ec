"""Do the two strands of a scan overlap usefully when they run on two streams?  Two contexts (own stream, own workspaces),
one host thread each, against the one-call both-strands entry on one stream."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _pkg import load_pkg
pkg = load_pkg(); lib, sy = pkg._lib, pkg.synth
N, L, K, PL = 100000, 200, 200, 12
seed = sy.SEED_BASE + 2
codes = sy.gen_codes(N, L, seed, n_plant=5, k=PL)
pwms, lens = sy.gen_pwm_bank(K, seed, len_lo=PL, len_hi=PL, alpha=0.3)
bank = sy.pad_bank(pwms, lens)
A, B = lib.Context(0), lib.Context(0)
raw = torch.from_numpy(codes).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
A.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr()); A.synchronize()
need = A.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
cap = max(need) + 1024
hits = [torch.empty((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
hsc = [torch.empty(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
torch.cuda.synchronize()
def both():
    A.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [h.data_ptr() for h in hits], [s.data_ptr() for s in hsc], cap)
def one(ctx, rc):
    ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, rc, hits[rc].data_ptr(), hsc[rc].data_ptr(), cap)
def par():
    t = threading.Thread(target=one, args=(B, 1)); t.start(); one(A, 0); t.join()
for f in (both, par): f(); f()
for name, f in (("one stream, both strands in one call", both), ("two streams, one strand each", par)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): f()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per step")
