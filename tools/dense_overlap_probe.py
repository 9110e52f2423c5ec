"""Dense (a17) contract: does the candidate filter of one chunk hide under the write stream of another when the chunks
run on two streams?  Two contexts, one host thread each, chunks of the 20k-read launch."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _pkg import load_pkg
pkg = load_pkg(); lib, sy = pkg._lib, pkg.synth
N, L, K, PL = 20000, 200, 200, 12
seed = sy.SEED_BASE + 2
codes = sy.gen_codes(N, L, seed, n_plant=5, k=PL)
pwms, lens = sy.gen_pwm_bank(K, seed, len_lo=PL, len_hi=PL, alpha=0.3)
bank = sy.pad_bank(pwms, lens)
A, B = lib.Context(0), lib.Context(0)
raw = torch.from_numpy(codes).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
A.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr()); A.synchronize()
Lout = L - PL + 1
pitch = lib.Context.codes_pitch(L)
dense = torch.empty((Lout, N, K), dtype=torch.int16, device="cuda")
nbytes = N * L + K * 4 * PL * 2 + N * K * Lout * 2
def whole():
    A.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, dense.data_ptr(), Lout); A.synchronize()
def chunks(ctx, lo, hi, step):
    for s in range(lo, hi, step):
        ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr() + s * pitch, step, L, dense.data_ptr() + s * Lout * K * 2, Lout)
    ctx.synchronize()
def par(step):
    t = threading.Thread(target=chunks, args=(B, N // 2, N, step)); t.start(); chunks(A, 0, N // 2, step); t.join()
def bench(name, f):
    f(); f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{name}: {dt * 1e3:.3f} ms  {nbytes / dt / 1e12:.2f} TB/s")
bench("one call", whole)
bench("one stream, 4 chunks", lambda: chunks(A, 0, N, 5000))
for step in (10000, 5000, 2500):
    bench(f"two streams, chunks of {step}", lambda: par(step))
