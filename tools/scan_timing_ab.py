import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from _pkg import load_pkg
pkg = load_pkg(); lib, sy = pkg._lib, pkg.synth
ctx = lib.Context(0)
N, L, K = 100000, 200, 200
codes = sy.gen_codes(N, L, 1)
pwms, lens = sy.gen_pwm_bank(K, 2, 12, 12)
bank = sy.pad_bank(pwms, lens)
raw = torch.from_numpy(np.ascontiguousarray(codes)).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
cap = int(max(need) * 1.05) + 1024
hits = [torch.empty((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
hsc = [torch.empty(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
counts = torch.zeros((2, K), dtype=torch.int64, device="cuda")
def step():
    return ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [h.data_ptr() for h in hits], [x.data_ptr() for x in hsc], cap, counts_ptr=counts.data_ptr())
for timing in (False, True, False, True):
    ctx.enable_timing(timing); ctx.reset_timing()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    print("timing", timing, "ms/step %.4f" % (dt * 1e3))
