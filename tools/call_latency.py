import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _pkg import load_pkg
pkg = load_pkg(); lib, sy = pkg._lib, pkg.synth
ctx = lib.Context(0)
L, K = 200, 200
pwms, lens = sy.gen_pwm_bank(K, 12, 12, 12, alpha=0.3)
bank = sy.pad_bank(pwms, lens)
for N in (64, 5000, 12500):
    codes = sy.gen_codes(N, L, 11)
    raw = torch.from_numpy(np.ascontiguousarray(codes)).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
    cap = max(need) + 16
    hits = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
    sc = [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
    counts = torch.zeros((2, K), dtype=torch.int64, device="cuda")
    hp, sp, cp, dp = [h.data_ptr() for h in hits], [s.data_ptr() for s in sc], counts.data_ptr(), dcodes.data_ptr()
    for _ in range(20): ctx.pwm_scan_hits_both_dev(bank, lens, dp, N, L, hp, sp, cap, counts_ptr=cp)
    t0 = time.perf_counter()
    for _ in range(200): ctx.pwm_scan_hits_both_dev(bank, lens, dp, N, L, hp, sp, cap, counts_ptr=cp)
    dt = (time.perf_counter() - t0) / 200
    print(f"N={N}: {dt * 1e6:.1f} us per both-strands call")
    ctx.enable_timing(slots=[lib.KS_SCAN_COUNT, lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL]); ctx.reset_timing()
    for _ in range(50): ctx.pwm_scan_hits_both_dev(bank, lens, dp, N, L, hp, sp, cap, counts_ptr=cp)
    parts = [ctx.kernel_ms(s) for s in (lib.KS_SCAN_COUNT, lib.KS_SCAN_OFFSETS, lib.KS_SCAN_FILL)]
    ctx.enable_timing(False)
    print("   per strand: cand %.1f us, stage+scan %.1f us, emit %.1f us" % tuple(ms / max(k, 1) * 1e3 for ms, k in parts))
