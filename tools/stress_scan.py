import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from _pkg import load_pkg
pkg = load_pkg(); lib, sy = pkg._lib, pkg.synth
ctx = lib.Context(0)
ctx.set_stream(0)      # torch's fills / uploads run on the null stream: the library's kernels must queue behind them, not beside
N, L, K = 100000, 200, 200
codes = sy.gen_codes(N, L, 11, n_plant=5, k=12)
codes[::977, 3] = 4
pwms, lens = sy.gen_pwm_bank(K, 12, 12, 12, alpha=0.3)
bank = sy.pad_bank(pwms, lens)
raw = torch.from_numpy(np.ascontiguousarray(codes)).cuda()
dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
cap = max(need) + 16
hits = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
sc = [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
counts = torch.zeros((2, K), dtype=torch.int64, device="cuda")
ref = None
for it in range(300):
    counts.zero_()
    tot = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [h.data_ptr() for h in hits], [s.data_ptr() for s in sc], cap, counts_ptr=counts.data_ptr())
    sig = (tuple(tot), int(hits[0][:tot[0]].to(torch.int64).sum().item()), int(hits[1][:tot[1]].to(torch.int64).sum().item()),
           int(sc[0][:tot[0]].to(torch.int64).sum().item()), int(sc[1][:tot[1]].to(torch.int64).sum().item()), int((counts * torch.arange(1, K + 1, device="cuda")).sum().item()))
    if ref is None: ref = sig
    assert sig == ref, (it, sig, ref)
    if it % 100 == 0: print("iter", it, sig[0], flush=True)
print("stress ok: 300 identical passes", ref[0])
