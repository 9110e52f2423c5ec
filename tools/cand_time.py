"""Times of the scan's stages (HIP events of the KS_SCAN_* slots) at BASELINE configs[1] size, one strand: candidate
kernel, stage_hits + row scan, emit_records (needs real output buffers)."""
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
cap = 32_000_000
hits = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
scores = torch.empty(cap, dtype=torch.int16, device="cuda")
for _ in range(3): ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, 0, hits.data_ptr(), scores.data_ptr(), cap)
ctx.enable_timing(slots=[lib.KS_SCAN_COUNT, lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL]); ctx.reset_timing()
for _ in range(10): n = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, 0, hits.data_ptr(), scores.data_ptr(), cap)
ms, k = ctx.kernel_ms(lib.KS_SCAN_COUNT)
ms2, k2 = ctx.kernel_ms(lib.KS_SCAN_OFFSETS)
ms3, k3 = ctx.kernel_ms(lib.KS_SCAN_FILL)
print(f"cand {ms / k:.4f} ms, stage+scan {ms2 / k2:.4f} ms, emit {ms3 / max(k3, 1):.4f} ms, hits {n}")
