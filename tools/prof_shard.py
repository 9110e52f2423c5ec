"""Small driver for rocprofv3 runs: a few hit-record scans (both strands) at the BASELINE configs[4] bank shape (2048 PWMs of 8-20
positions on 1000 bp reads; N reads, default 10 000 = two ordering batches in one launch) or, CFG=3, the configs[3] shape
(512 PWMs of 20 positions on 500 bp reads, N default 20 000): the chunk-group kernels stage_hits_cg / emit_records_cg."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from _pkg import load_pkg  # noqa: E402

pkg = load_pkg()
lib, sy = pkg._lib, pkg.synth
cfg = int(os.environ.get("CFG", 4))
if cfg == 3:
    N, L, K, lo, hi = int(os.environ.get("N", 20000)), 500, 512, 20, 20
else:
    N, L, K, lo, hi = int(os.environ.get("N", 10000)), 1000, 2048, 8, 20
reps = int(os.environ.get("REPS", 2))
seed = sy.SEED_BASE + 2
pw, ln = sy.gen_pwm_bank(K, seed + 7, len_lo=lo, len_hi=hi, alpha=0.3)
bk = sy.pad_bank(pw, ln)
cd = sy.gen_codes(N, L, seed + 31, n_plant=5, k=hi)
ctx = lib.Context(0)
raw = torch.from_numpy(cd).cuda()
dc = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dc.data_ptr())
need = ctx.pwm_scan_hits_both_dev(bk, ln, dc.data_ptr(), N, L, None, None, 0)
cap = max(need) + 16
h = [torch.empty((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
s = [torch.empty(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
for _ in range(reps):
    got = ctx.pwm_scan_hits_both_dev(bk, ln, dc.data_ptr(), N, L, [x.data_ptr() for x in h], [x.data_ptr() for x in s], cap)
ctx.synchronize()
print("cfg", cfg, "reads", N, "hits", got, "plan", ctx.scan_plan())
