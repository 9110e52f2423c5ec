"""Small driver for rocprofv3 runs: a few launches of each scan kernel at BASELINE configs[1] size."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from _pkg import load_pkg  # noqa: E402

pkg = load_pkg()
lib, sy = pkg._lib, pkg.synth
N, L, K, PL = int(os.environ.get("N", 100000)), 200, int(os.environ.get("K", 200)), int(os.environ.get("PL", 12))
reps = int(os.environ.get("REPS", 3))
codes = sy.gen_codes(N, L, sy.SEED_BASE + 2, n_plant=5, k=PL)
pwms, lens = sy.gen_pwm_bank(K, sy.SEED_BASE + 2, len_lo=PL, len_hi=PL, alpha=float(os.environ.get("ALPHA", 0.3)))
bank = sy.pad_bank(pwms, lens)
ctx = lib.Context(0)
raw = torch.from_numpy(codes).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
# the hit-record path exactly as bench.py drives it: gpu_scan, both strands in one call (one candidate launch for the two banks)
need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
n = max(need)
hits = [torch.empty((n + 16, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
sc = [torch.empty(n + 16, dtype=torch.int16, device="cuda") for _ in range(2)]
for _ in range(reps):
    ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [h.data_ptr() for h in hits], [s.data_ptr() for s in sc], n + 16)
nb = min(N, 20000)
Lout = L - PL + 1
dense = torch.empty((Lout, nb, K), dtype=torch.int16, device="cuda")
for _ in range(reps):
    ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), nb, L, dense.data_ptr(), Lout)
ctx.synchronize()
print("hits", n)
