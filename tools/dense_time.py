"""Time of the dense contract (motifs_pwm_scan_dense_dev: candidate kernel + stage_hits<.,.,2>) at the bench's launch size:
20k reads x 200 bp, 200 PWMs of length 12; HIP events of the KS_SCAN_DENSE slot."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _pkg import load_pkg
pkg = load_pkg(); lib, sy = pkg._lib, pkg.synth
ctx = lib.Context(0)
N, L, K = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 200, 200
codes = sy.gen_codes(N, L, 1)
pwms, lens = sy.gen_pwm_bank(K, 2, 12, 12)
bank = sy.pad_bank(pwms, lens)
Lout = L - 12 + 1
raw = torch.from_numpy(np.ascontiguousarray(codes)).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
dense = torch.empty((Lout, N, K), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
for _ in range(3): ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, dense.data_ptr(), Lout)
ctx.synchronize()
ctx.enable_timing(slots=[lib.KS_SCAN_DENSE]); ctx.reset_timing()
for _ in range(20): ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, dense.data_ptr(), Lout)
ctx.synchronize()
ms, k = ctx.kernel_ms(lib.KS_SCAN_DENSE)
byts = N * L + K * 4 * 12 * 2 + N * K * Lout * 2
print(f"N {N}: dense {ms / k:.4f} ms per launch, {byts / (ms / k) / 1e6:.0f} GB/s = {byts / (ms / k) / 8e9:.3f} of 8 TB/s")
