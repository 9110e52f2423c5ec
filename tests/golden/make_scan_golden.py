"""Generates tests/golden/scan_small.npz with the CPU oracle (the reference cannot
run here: no Julia; it also ships no fixtures).  Inputs are seeded; outputs are
what oracle/scan_oracle.c produces, frozen so that later edits to the oracle or
the kernels are caught."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_pkg  # noqa: E402
from oracle import scan_oracle as so  # noqa: E402

sy = load_pkg().synth
N, L, K, batch = 37, 60, 11, 16
codes = sy.gen_codes(N, L, 20260101, n_plant=2, k=8)
pwms, lens = sy.gen_pwm_bank(K, 20260101, len_lo=5, len_hi=13, alpha=0.5)
bank = sy.pad_bank(pwms, lens)
onehot = sy.codes_to_onehot(codes)
out = dict(bank=bank, lens=lens, onehot=onehot, codes=codes, batch=np.int64(batch))
for rc in (0, 1):
    f, s = so.get_pos_scores_arr(bank, lens, onehot, rc=bool(rc), batch_size=batch)
    out[f"found_rc{rc}"] = np.stack([f["m"], f["n"], f["l"]], axis=1).astype(np.uint32)
    out[f"score_rc{rc}"] = s.view(np.uint16)
    print("rc", rc, "hits", len(f))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scan_small.npz"), **out)
