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
if os.environ.get("ASYNC") == "1":        # the call returns once the totals are known (motifs_ctx_set_records_in_stream_order)
    ctx.set_records_in_stream_order(True)
if os.environ.get("WSTREAM") == "1":      # as bench.py: one ordinary stream for torch and the library
    _ws = torch.cuda.Stream()
    torch.cuda.set_stream(_ws)
    ctx.set_stream(_ws.cuda_stream)
raw = torch.from_numpy(codes).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
n = max(need)
hits = [torch.empty((n + 16, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
sc = [torch.empty(n + 16, dtype=torch.int16, device="cuda") for _ in range(2)]
args = (bank, lens, dcodes.data_ptr(), N, L, [h.data_ptr() for h in hits], [s.data_ptr() for s in sc], n + 16)
if os.environ.get("COUNTS") == "1":       # with the 2 x K hit histogram, as bench.py's step asks for it
    counts = torch.zeros((2, K), dtype=torch.int64, device="cuda")
    args = args + (0, lib.SCAN_BATCH, counts.data_ptr())
if os.environ.get("TIMED") == "1":        # as bench.py's timed region: the candidate kernel's launches stamped with HIP events
    ctx.enable_timing(slots=[lib.KS_SCAN_COUNT])
for _ in range(30):
    ctx.pwm_scan_hits_both_dev(*args)
ts = []
ctx.synchronize()
t_all = time.perf_counter()
for _ in range(reps):
    t0 = time.perf_counter()
    ctx.pwm_scan_hits_both_dev(*args)
    ts.append(time.perf_counter() - t0)
ctx.synchronize()
t_all = (time.perf_counter() - t_all) / reps
print("per step over the whole loop (synchronised at its end): %.4f ms" % (1e3 * t_all))
ctx.enable_timing(slots=[lib.KS_SCAN_COUNT, lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL])
ctx.reset_timing()
for _ in range(10):
    ctx.pwm_scan_hits_both_dev(*args)
st = [ctx.kernel_ms(s)[0] / 10 for s in (lib.KS_SCAN_COUNT, lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL)]
print("hits %d  step median %.4f ms (min %.4f)  stages cand %.3f stage %.3f emit %.3f  %s" % (
    sum(need), 1e3 * float(np.median(ts)), 1e3 * min(ts), st[0], st[1], st[2],
    " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("MOTIFS_"))))
