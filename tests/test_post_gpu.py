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


# (the count matrices fetch a window as 4, 6 or 8 dwords by the longest PWM: 12 / 20 / 28 positions; byte by byte past that)
@pytest.mark.parametrize("K,lo,hi", [(9, 5, 9), (150, 6, 12), (40, 14, 20), (20, 22, 27), (6, 30, 33)])
def test_consumers_match_oracle(ctx, pkg, K, lo, hi):
    sy, post = pkg.synth, pkg.post
    N, L = 300, 60
    codes = sy.gen_codes(N, L, 31 + K, n_plant=3, k=8)
    codes[5, 7] = 4                      # an N inside windows that still score: no count from that position
    codes[N - 1, L - 1] = 4
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


def test_count_matrices_of_a_shard(ctx, pkg):
    """A rank's shard: its code rows start at global read n0 + 1 and its records carry global read numbers; the matrices of the
    shard are the oracle's over exactly those records.  The second call passes the code rows from an odd address (the dword-window
    kernel asks for 4-byte alignment: the byte-by-byte kernel takes over) and must give the same counts."""
    sy, post, lib = pkg.synth, pkg.post, pkg._lib
    N, L, K, s0 = 500, 77, 60, 123
    codes = sy.gen_codes(N, L, 77, n_plant=3, k=9)
    codes[s0 + 5, 40] = 4
    pwms, lens = sy.gen_pwm_bank(K, 78, len_lo=7, len_hi=15, alpha=0.4)
    bank = sy.pad_bank(pwms, lens)
    maxlen = int(lens.max())
    ns = N - s0
    raw = torch.from_numpy(codes[s0:]).cuda()
    pad = 4
    buf = torch.zeros(lib.Context.codes_bytes(ns, L) + 2 * pad, dtype=torch.uint8, device="cuda")
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, ns, L, buf.data_ptr())
    want = np.zeros((K, maxlen, 4), dtype=np.float32)
    strands = []
    for rc in (False, True):
        n = ctx.pwm_scan_hits_dev(bank, lens, buf.data_ptr(), ns, L, rc, None, None, 0, n0=s0)
        hits = torch.zeros((max(n, 1), 3), dtype=torch.int32, device="cuda")
        sc = torch.zeros(max(n, 1), dtype=torch.int16, device="cuda")
        ctx.pwm_scan_hits_dev(bank, lens, buf.data_ptr(), ns, L, rc, hits.data_ptr(), sc.data_ptr(), n, n0=s0)
        ctx.synchronize()
        h = hits[:n].cpu().numpy().astype(np.int64)
        assert n > 500 and h[:, 1].min() > s0 and h[:, 1].max() <= N
        want += po.countmats(h[:, 0], h[:, 1], h[:, 2], rc, codes, lens, K, maxlen).astype(np.float32)
        strands.append((hits, n, rc))
    got = post.posdicts2countmats(ctx, strands, buf.data_ptr(), L, lens, maxlen, n0=s0)
    # the same rows one byte further on: an odd address
    nbytes = lib.Context.codes_bytes(ns, L)
    odd = torch.zeros(nbytes + 2 * pad, dtype=torch.uint8, device="cuda")
    odd[1:1 + nbytes] = buf[:nbytes]
    torch.cuda.synchronize()
    got_odd = post.posdicts2countmats(ctx, strands, odd.data_ptr() + 1, L, lens, maxlen, n0=s0)
    for k in range(K):
        w = (want[k, : int(lens[k]), :].T + np.float32(0.01)).astype(np.float16)
        assert np.array_equal(got[k].view(np.uint16), w.view(np.uint16))
        assert np.array_equal(got_odd[k].view(np.uint16), w.view(np.uint16))


# ---- consumers of the code records (§8f-4) -----------------------------------------------------------------------
def _random_code_records(pkg, nseq, seed, gap_every=0):
    rng = np.random.default_rng(seed)
    rows = []
    for s in range(1, nseq + 1):
        if gap_every and s % gap_every == 0:
            continue                                          # a sequence without components
        ncomp = int(rng.integers(1, 14))
        fils = np.sort(rng.integers(1, 25, size=ncomp))       # retrieval order: syntax filter, then position
        for f in fils:
            rows.append((int(rng.integers(1, 179)), int(f), s, float(np.float16(rng.uniform(0.01, 9.0)))))
    recs = np.zeros(len(rows), dtype=pkg._lib.CODE_DTYPE)
    for i, (p, f, s, m) in enumerate(rows):
        recs[i] = (p, f, s, np.float16(m))
    order = np.lexsort((recs["position"], recs["fil"], recs["seq"]))
    return recs[order]


@pytest.mark.parametrize("nseq,p,gap", [(60, 0.05, 0), (200, 0.35, 7), (40, 0.75, 0)])
def test_quantile_filter_and_triplets_match_oracle(ctx, pkg, nseq, p, gap):
    post = pkg.post
    recs = _random_code_records(pkg, nseq, 100 + nseq, gap)
    n = len(recs)
    dev = torch.from_numpy(recs.view(np.uint8).reshape(n, 12)).cuda()
    want_f, want_thr = po.filter_code_components_using_quantile(recs, p)
    out, m, thr = post.filter_code_components(ctx, dev, n, p)
    assert thr == want_thr and m == len(want_f)
    got_f = out[:m].cpu().numpy().reshape(-1).view(pkg._lib.CODE_DTYPE)
    assert np.array_equal(got_f, want_f)

    H = po.enumerate_triplets(want_f, po.scanning_ranges(want_f), h=12)
    got = post.enumerate_triplets(ctx, out, m, 12)
    assert got["n_triplets"] == sum(len(v) for v in H.values())
    assert [tuple(int(x) for x in k) for k in got["keys"]] == list(H.keys()), "keys or their insertion order differ"
    assert list(got["counts"]) == [len(v) for v in H.values()]
    flat = [(s, pos) for v in H.values() for (s, pos, _) in v]
    assert list(zip(got["values"]["seq_num"].tolist(), got["values"]["pos"].tolist())) == flat
    assert not got["values"]["comp"].any()
