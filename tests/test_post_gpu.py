"""GPU parity of the record consumers (score range, threshold sweep counts, threshold filter, count matrices)
against oracle/post_oracle.py, on records produced by the scan itself.  All integer results are exact."""
import numpy as np
import pytest
import torch

from oracle import post_oracle as po

pytestmark = pytest.mark.gpu


def scan_to_device(ctx, pkg, bank, lens, codes, rc):
    lib = pkg._lib
    N, L = codes.shape
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    n = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, rc, None, None, 0)
    hits = torch.zeros((max(n, 1), 3), dtype=torch.int32, device="cuda")
    sc = torch.zeros(max(n, 1), dtype=torch.int16, device="cuda")
    ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, rc, hits.data_ptr(), sc.data_ptr(), n)
    ctx.synchronize()
    return dcodes, hits, sc, n


@pytest.mark.parametrize("K,lo,hi", [(9, 5, 9), (150, 6, 12)])
def test_consumers_match_oracle(ctx, pkg, K, lo, hi):
    sy, post = pkg.synth, pkg.post
    N, L = 300, 60
    codes = sy.gen_codes(N, L, 31 + K, n_plant=3, k=8)
    pwms, lens = sy.gen_pwm_bank(K, 32 + K, len_lo=lo, len_hi=hi, alpha=0.45)
    bank = sy.pad_bank(pwms, lens)
    maxlen = int(lens.max())
    strands = []
    for rc in (False, True):
        dcodes, hits, sc, n = scan_to_device(ctx, pkg, bank, lens, codes, rc)
        h = hits[:n].cpu().numpy().astype(np.int64)
        s = sc[:n].cpu().numpy().view(np.float16)
        m, nn, ll = h[:, 0], h[:, 1], h[:, 2]
        # score range
        mn, mx = post.score_range(ctx, hits, sc, n, K)
        omn, omx = po.minmax_by_motif(m, s, K)
        assert np.array_equal(mn.view(np.uint16), omn.view(np.uint16)) and np.array_equal(mx.view(np.uint16), omx.view(np.uint16))
        # threshold sweep counts == get_hits at every visited threshold
        thr, nthr = post.sweep_thresholds(np.where(np.isfinite(mn), mn, np.float16(0)), np.where(np.isfinite(mx), mx, np.float16(0)))
        counts = post.sweep_counts(ctx, hits, sc, n, thr)
        for k in range(K):
            sk = s[m == k + 1]
            for j in range(nthr[k]):
                assert counts[k, j] == po.get_hits(sk, thr[k, j])
        # threshold filter (stable)
        thresh = np.array([thr[k, nthr[k] // 2] if nthr[k] else np.float16(0) for k in range(K)], dtype=np.float16)
        oh, os_, kept = post.filter_by_thresh(ctx, hits, sc, n, thresh)
        fm, fn, fl, fs = po.filter_records(m, nn, ll, s, thresh)
        assert kept == len(fm)
        assert np.array_equal(oh[:kept].cpu().numpy().astype(np.int64), np.stack([fm, fn, fl], 1))
        assert np.array_equal(os_[:kept].cpu().numpy().view(np.uint16), fs.view(np.uint16))
        strands.append((hits, n, rc, m, nn, ll))
    # count matrices from both strands
    mats = post.posdicts2countmats(ctx, [(h, n, rc) for h, n, rc, *_ in strands], dcodes.data_ptr(), L, lens, maxlen)
    want = sum(po.countmats(m, nn, ll, rc, codes, lens, K, maxlen).astype(np.float32) for _, _, rc, m, nn, ll in strands)
    for k in range(K):
        w = (want[k, : int(lens[k]), :].T + np.float32(0.01)).astype(np.float16)
        assert np.array_equal(mats[k].view(np.uint16), w.view(np.uint16))
