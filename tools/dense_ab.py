"""A/B of the a17 dense-tensor path under environment switches (set before the context is created): prints GB/s on the dense contract."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _pkg import load_pkg
pkg = load_pkg(); lib, sy = pkg._lib, pkg.synth
N, L, K, PL = 100000, 200, 200, 12
codes = sy.gen_codes(N, L, sy.SEED_BASE + 2, n_plant=5, k=PL)
pwms, lens = sy.gen_pwm_bank(K, sy.SEED_BASE + 2, len_lo=PL, len_hi=PL, alpha=0.3)
bank = sy.pad_bank(pwms, lens)
ctx = lib.Context(0)
raw = torch.from_numpy(codes).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
Lout = L - PL + 1
for nb in (20000, 24576):
    t = torch.empty((Lout, nb, K), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    for _ in range(30):
        ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), nb, L, t.data_ptr(), Lout)
    ctx.enable_timing(True); ctx.reset_timing()
    for _ in range(10):
        ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), nb, L, t.data_ptr(), Lout)
    ms, n = ctx.kernel_ms(lib.KS_SCAN_DENSE); ctx.enable_timing(False)
    by = nb * L + K * 4 * PL * 2 + nb * K * Lout * 2
    pos = int((t > 0).sum().item()); chk = int(t[t > 0].to(torch.int64).sum().item())
    print(os.environ.get("MOTIFS_DENSE_COMPACT", "-"), "reads", nb, "ms %.4f" % (ms / n), "GB/s %.0f frac %.3f" % (by / (ms / n * 1e-3) / 1e9, by / (ms / n * 1e-3) / 1e9 / 8000), "positives", pos, "checksum", chk, flush=True)
