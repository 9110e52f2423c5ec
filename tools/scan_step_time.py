"""Wall time of the both-strands hit-record scan at BASELINE configs[1] (the step bench.py times), median of REPS calls after a warm-up, with the
HIP-event times of its three stages.  For A/B runs of a launch parameter in one gpurun call (switches are read once per process)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from _pkg import load_pkg  # noqa: E402

pkg = load_pkg()
lib, sy = pkg._lib, pkg.synth
N, L, K, PL = int(os.environ.get("N", 100000)), 200, 200, 12
reps = int(os.environ.get("REPS", 40))
codes = sy.gen_codes(N, L, sy.SEED_BASE + 2, n_plant=5, k=PL)
pwms, lens = sy.gen_pwm_bank(K, sy.SEED_BASE + 2, len_lo=PL, len_hi=PL, alpha=0.3)
bank = sy.pad_bank(pwms, lens)
ctx = lib.Context(0)
raw = torch.from_numpy(codes).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
n = max(need)
hits = [torch.empty((n + 16, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
sc = [torch.empty(n + 16, dtype=torch.int16, device="cuda") for _ in range(2)]
args = (bank, lens, dcodes.data_ptr(), N, L, [h.data_ptr() for h in hits], [s.data_ptr() for s in sc], n + 16)
for _ in range(30):
    ctx.pwm_scan_hits_both_dev(*args)
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    ctx.pwm_scan_hits_both_dev(*args)
    ts.append(time.perf_counter() - t0)
ctx.enable_timing(slots=[lib.KS_SCAN_COUNT, lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL])
ctx.reset_timing()
for _ in range(10):
    ctx.pwm_scan_hits_both_dev(*args)
st = [ctx.kernel_ms(s)[0] / 10 for s in (lib.KS_SCAN_COUNT, lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL)]
print("hits %d  step median %.4f ms (min %.4f)  stages cand %.3f stage %.3f emit %.3f  %s" % (
    sum(need), 1e3 * float(np.median(ts)), 1e3 * min(ts), st[0], st[1], st[2],
    " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MOTIFS_"))))
