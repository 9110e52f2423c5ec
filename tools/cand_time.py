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
ctx.enable_timing(slots=[lib.KS_SCAN_COUNT]); ctx.reset_timing()
for _ in range(3): ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, 0, None, None, 0, allow_small=True)
ctx.reset_timing()
for _ in range(10): ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, 0, None, None, 0, allow_small=True)
ms, n = ctx.kernel_ms(lib.KS_SCAN_COUNT)
print("cand ms per launch", ms / n, "nostore" if os.environ.get("MOTIFS_NOSTORE") else "store")
