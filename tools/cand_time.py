"""Time of the candidate kernel alone (HIP events of the KS_SCAN_COUNT slot) at BASELINE configs[1] size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
torch.cuda.synchronize()
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
for _ in range(3): ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, 0, None, None, 0, allow_small=True)
ctx.enable_timing(slots=[lib.KS_SCAN_COUNT, lib.KS_SCAN_OFFSETS]); ctx.reset_timing()
for _ in range(10): n = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, 0, None, None, 0, allow_small=True)
ms, k = ctx.kernel_ms(lib.KS_SCAN_COUNT)
ms2, k2 = ctx.kernel_ms(lib.KS_SCAN_OFFSETS)
print(f"dbg={os.environ.get('MOTIFS_CAND_DBG', '0')}: cand {ms / k:.4f} ms, stage+scan {ms2 / k2:.4f} ms, hits {n}")
