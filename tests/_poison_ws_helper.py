"""Helper of test_round4_gpu.py::test_scans_on_poisoned_workspaces: hit-record scans of a few awkward shapes in a fresh process (with
MOTIFS_POISON_WS=1 every new workspace starts as 0xFF bytes, so a cell, entry or staging slot that no kernel wrote but one reads gives a
wrong record instead of whatever the allocator left there), both strands, against the CPU port.  Exit status 0 = every record equal."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from _pkg import load_pkg  # noqa: E402
from oracle import scan_oracle as so  # noqa: E402

pkg = load_pkg()
sy, lib = pkg.synth, pkg._lib
# (reads, bp, PWMs, len_lo, len_hi, ordering batch): a short last batch, reads not a multiple of 4, three chunks, one PWM tile, long PWMs (cells, no entries)
SHAPES = [(5003, 120, 200, 12, 12, 5000), (1237, 203, 328, 8, 20, 500), (64, 1001, 24, 6, 9, 33), (900, 150, 40, 24, 40, 5000)]
for i, (N, L, K, lo, hi, batch) in enumerate(SHAPES):
    c = lib.Context(0)                                  # a fresh context: every workspace is allocated (and poisoned) anew
    pwms, lens = sy.gen_pwm_bank(K, 9700 + i, len_lo=lo, len_hi=hi, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    codes = sy.gen_codes(N, L, 9800 + i, n_plant=4, k=min(hi, 12))
    if i == 1:
        codes[7, 50] = 4                                # an all-zero column
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    need = c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0, batch=batch)
    cap = max(need) + 8
    hits = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
    scs = [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
    got = c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [t.data_ptr() for t in hits], [t.data_ptr() for t in scs], cap, batch=batch)
    c.synchronize()
    assert got == need, (got, need)
    for rc in (0, 1):
        f, s = so.get_pos_scores_arr_fast(bank, lens, codes, rc=bool(rc), batch_size=batch)
        oh = np.stack([f["m"], f["n"], f["l"]], axis=1).astype(np.uint32)
        assert len(oh) == got[rc], (i, rc, len(oh), got[rc])
        assert np.array_equal(hits[rc][:got[rc]].cpu().numpy().astype(np.uint32), oh), (i, rc)
        assert np.array_equal(scs[rc][:got[rc]].cpu().numpy().view(np.uint16), s.view(np.uint16)), (i, rc)
    c.close()
print("ok")
