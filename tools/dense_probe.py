"""Times motifs_pwm_scan_dense_dev at 20k reads x 200 bp x 200 PWMs (the dense leg of bench.py on its own)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from _pkg import load_pkg
pkg = load_pkg(); lib, sy = pkg._lib, pkg.synth
N, L, K, PL = 20000, 200, 200, 12
codes = sy.gen_codes(N, L, sy.SEED_BASE + 2, n_plant=5, k=PL)
pwms, lens = sy.gen_pwm_bank(K, sy.SEED_BASE + 2, len_lo=PL, len_hi=PL, alpha=0.3)
bank = sy.pad_bank(pwms, lens)
ctx = lib.Context(0)
raw = torch.from_numpy(codes).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
Lout = L - PL + 1
dense = torch.empty((Lout, N, K), dtype=torch.int16, device="cuda")
for _ in range(2): ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, dense.data_ptr(), Lout)
ctx.enable_timing(True); ctx.reset_timing()
for _ in range(5): ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, dense.data_ptr(), Lout)
ms, n = ctx.kernel_ms(lib.KS_SCAN_DENSE)
dense_bytes = N * L + K * 4 * PL * 2 + N * K * Lout * 2
print("dense scan of %d reads: %.3f ms per call, %.2f TB/s on the dense-score contract" % (N, ms / n, dense_bytes / (ms / n * 1e-3) / 1e12))
