"""Parity tests proper: the HIP scan, called through the C ABI, against the CPU oracle.
Integer hit records and fp16 scores must be bit-exact (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

from oracle import scan_oracle as so

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def dev_scan_hits(torch, ctx, pkg, bank, lens, codes, rc, batch, n0=0, want_counts=False):
    """Device-resident path: encode -> count -> offsets -> fill."""
    L = codes.shape[1]
    N = codes.shape[0]
    lib = pkg._lib
    raw = torch.from_numpy(np.ascontiguousarray(codes)).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    counts = torch.zeros(bank.shape[2], dtype=torch.int64, device="cuda") if want_counts else None
    n = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, rc, None, None, 0, n0=n0, batch=batch)
    hits = torch.zeros((max(n, 1), 3), dtype=torch.int32, device="cuda")
    sc = torch.zeros(max(n, 1), dtype=torch.int16, device="cuda")
    n2 = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, rc, hits.data_ptr(), sc.data_ptr(), n, n0=n0,
                               batch=batch, counts_ptr=counts.data_ptr() if want_counts else None)
    ctx.synchronize()
    assert n2 == n
    h = hits[:n].cpu().numpy().astype(np.uint32)
    s = sc[:n].cpu().numpy().view(np.uint16)
    if want_counts:
        return h, s, counts.cpu().numpy()
    return h, s


def oracle_hits(pkg, bank, lens, codes, rc, batch):
    f, s = so.get_pos_scores_arr(bank, lens, pkg.synth.codes_to_onehot(codes), rc=rc, batch_size=batch)
    return np.stack([f["m"], f["n"], f["l"]], axis=1).astype(np.uint32), s.view(np.uint16)


def test_golden_fixture(torch_cuda, ctx, pkg):
    g = np.load(os.path.join(HERE, "golden", "scan_small.npz"))
    for rc in (0, 1):
        h, s = dev_scan_hits(torch_cuda, ctx, pkg, g["bank"], g["lens"], g["codes"], rc, int(g["batch"]))
        assert np.array_equal(h, g[f"found_rc{rc}"])
        assert np.array_equal(s, g[f"score_rc{rc}"])


CASES = [
    # N, L, K, len_lo, len_hi, batch, alpha
    (50, 100, 32, 8, 8, 5000, 0.4),        # cfg-1 shape
    (33, 47, 7, 3, 7, 10, 0.6),            # odd K, L % 4 != 0, LEN=8 template with short PWMs
    (70, 61, 131, 9, 12, 32, 0.5),         # 66 pairs -> two chunks, LEN=12
    (40, 90, 20, 13, 16, 17, 0.5),         # LEN=16
    (30, 75, 9, 17, 20, 8, 0.5),           # LEN=20
    (25, 70, 6, 21, 24, 25, 0.6),          # LEN=24
    (21, 80, 5, 25, 32, 5, 0.7),           # LEN=32
    (10, 12, 4, 12, 12, 3, 0.9),           # a single window per sequence
    (300, 40, 260, 6, 10, 64, 0.4),        # three chunks
]


@pytest.mark.parametrize("N,L,K,lo,hi,batch,alpha", CASES)
@pytest.mark.parametrize("rc", [False, True])
def test_hits_match_oracle(torch_cuda, ctx, pkg, N, L, K, lo, hi, batch, alpha, rc):
    sy = pkg.synth
    codes = sy.gen_codes(N, L, 100 + N + K, n_plant=2, k=min(8, L))
    codes[N // 2, L // 3] = 4                      # an all-zero column
    pwms, lens = sy.gen_pwm_bank(K, 200 + K, len_lo=lo, len_hi=hi, alpha=alpha)
    bank = sy.pad_bank(pwms, lens)
    h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, batch, want_counts=True)
    oh, os_ = oracle_hits(pkg, bank, lens, codes, rc, batch)
    assert len(oh) > 0
    assert np.array_equal(h, oh), "hit records (m,n,l) or their order differ"
    assert np.array_equal(s, os_), "fp16 scores differ"
    assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))


@pytest.mark.parametrize("N,L,K,batch", [(37, 61, 24, 16), (300, 100, 200, 64), (5, 30, 8, 5000)])
def test_both_strands_entry_equals_two_calls(torch_cuda, ctx, pkg, N, L, K, batch):
    """gpu_scan in one call (motifs_pwm_scan_hits_both_dev): the same records, scores and histograms as the forward and the
    reverse scan one after the other, and as the oracle."""
    torch = torch_cuda
    lib, sy = pkg._lib, pkg.synth
    codes = sy.gen_codes(N, L, 300 + N + K, n_plant=2, k=min(8, L))
    pwms, lens = sy.gen_pwm_bank(K, 400 + K, len_lo=6, len_hi=12, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    raw = torch.from_numpy(np.ascontiguousarray(codes)).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0, n0=3, batch=batch)
    cap = max(max(need), 1)
    hits = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
    sc = [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
    counts = torch.zeros((2, K), dtype=torch.int64, device="cuda")
    got = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [h.data_ptr() for h in hits], [x.data_ptr() for x in sc], cap,
                                     n0=3, batch=batch, counts_ptr=counts.data_ptr())
    ctx.synchronize()
    assert got == need
    for rc in (0, 1):
        h1, s1, c1 = dev_scan_hits(torch, ctx, pkg, bank, lens, codes, rc, batch, n0=3, want_counts=True)
        n = got[rc]
        assert n == len(h1)
        assert np.array_equal(hits[rc][:n].cpu().numpy().astype(np.uint32), h1)
        assert np.array_equal(sc[rc][:n].cpu().numpy().view(np.uint16), s1)
        assert np.array_equal(counts[rc].cpu().numpy(), c1)
        ho, so_ = oracle_hits(pkg, bank, lens, codes, rc, batch)
        ho[:, 1] += 3
        assert np.array_equal(h1, ho) and np.array_equal(s1, so_)


def test_pwm_longer_than_sequence_and_empty(torch_cuda, ctx, pkg):
    sy = pkg.synth
    codes = sy.gen_codes(5, 10, 1)
    pwms, lens = sy.gen_pwm_bank(3, 2, len_lo=11, len_hi=12)
    bank = sy.pad_bank(pwms, lens)
    h, s = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, False, 5000)
    assert len(h) == 0
    # one PWM fits, the others do not
    pwms2, lens2 = sy.gen_pwm_bank(3, 3, len_lo=4, len_hi=4, alpha=0.9)
    pw = [pwms[0], pwms2[1], pwms[2]]
    ln = np.array([lens[0], 4, lens[2]])
    bank = sy.pad_bank(pw, ln)
    h, s = dev_scan_hits(torch_cuda, ctx, pkg, bank, ln, codes, False, 5000)
    oh, os_ = oracle_hits(pkg, bank, ln, codes, False, 5000)
    assert np.array_equal(h, oh) and np.array_equal(s, os_) and set(h[:, 0]) <= {2}


def test_dense_matches_greedy_search_layout(torch_cuda, ctx, pkg):
    """a17 drop-in: the (K, nb, 4L) tensor of _h3_1_alignment.jl:75-80, zeros included."""
    torch = torch_cuda
    sy = pkg.synth
    # K % 8 == 0 takes the matrix-core kernel (zeros + re-scored candidates), the others the packed-add kernel
    for (N, L, K, lo, hi) in [(19, 50, 10, 6, 12), (8, 33, 7, 5, 9), (12, 64, 140, 12, 12), (37, 70, 200, 12, 12),
                              (21, 45, 136, 5, 12), (9, 90, 64, 17, 24), (23, 41, 8, 3, 8), (6, 120, 264, 13, 20)]:
        codes = sy.gen_codes(N, L, 9 + K)
        codes[N // 2, L // 2] = 4                  # an all-zero column
        pwms, lens = sy.gen_pwm_bank(K, 10 + K, len_lo=lo, len_hi=hi, alpha=0.6)
        bank = sy.pad_bank(pwms, lens)
        want = so.greedy_search(bank, lens, sy.codes_to_onehot(codes).astype(np.float16))  # (4L, N, K)
        raw = torch.from_numpy(codes).cuda()
        dcodes = torch.zeros(pkg._lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
        out = torch.full((4 * L, N, K), 0x7BFF, dtype=torch.int16, device="cuda")  # poison: must be overwritten
        torch.cuda.synchronize()
        ctx.encode_dev(raw.data_ptr(), pkg._lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
        ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, out.data_ptr(), 4 * L)
        ctx.synchronize()
        got = out.cpu().numpy().view(np.uint16)
        assert np.array_equal(got, want.view(np.uint16))


def test_host_entry_and_mirror(torch_cuda, ctx, pkg):
    """motifs_pwm_scan (host buffers, f32 one-hot) + gpu_scan dict build (:38-52, :89-99)."""
    sy, sc = pkg.synth, pkg.scan
    N, L, K = 5200, 24, 12                        # crosses the 5000-sequence batch boundary (:71)
    codes = sy.gen_codes(N, L, 77, n_plant=2, k=8)
    pwms, lens = sy.gen_pwm_bank(K, 78, len_lo=6, len_hi=8, alpha=0.35)
    bank = sy.pad_bank(pwms, lens)
    onehot = sy.codes_to_onehot(codes)
    data = sc.FastaData(onehot.reshape(N, 1, 4 * L))
    ms = sc.Motifs(pwms, lens)
    for rc in (False, True):
        found, score = sc.get_pos_scores_arr(ms, data, rc=rc, ctx=ctx)
        of, os_ = so.get_pos_scores_arr(bank, lens, onehot, rc=rc)
        assert np.array_equal(found, of) and np.array_equal(score.view(np.uint16), os_.view(np.uint16))
        assert found["n"].max() > 5000
    # gpu_scan's host entry (one upload, both strands) against the two single-strand calls
    both = ctx.pwm_scan_both(bank, lens, onehot, pkg._lib.DATA_ONEHOT_F32, N, L)
    for rc in (0, 1):
        f1, s1 = ctx.pwm_scan(bank, lens, onehot, pkg._lib.DATA_ONEHOT_F32, N, L, bool(rc))
        assert np.array_equal(both[rc][0], f1) and np.array_equal(both[rc][1].view(np.uint16), s1.view(np.uint16))
    with pytest.raises(pkg._lib.MotifsError) as e:
        ctx.pwm_scan_both(bank, lens, onehot, pkg._lib.DATA_ONEHOT_F32, N, L, cap=10)
    assert e.value.code == pkg._lib.ERR_BUFFER_TOO_SMALL
    sc.scan_w_gpu(ms, data, ctx=ctx)
    # per (m, n): forward hits in ascending l, then reverse-strand hits in ascending l
    m0 = next(i for i, d in enumerate(ms.positions) if d)
    n0 = next(iter(ms.positions[m0]))
    pos, comp = ms.positions[m0][n0], ms.use_comp[m0][n0]
    fwd = [p for p, c in zip(pos, comp) if not c]
    rev = [p for p, c in zip(pos, comp) if c]
    assert pos == fwd + rev and fwd == sorted(fwd) and rev == sorted(rev)
    total = sum(len(v) for d in ms.positions for v in d.values())
    assert total == sum(len(so.get_pos_scores_arr(bank, lens, onehot, rc=r)[0]) for r in (False, True))


def test_f16_input_and_errors(torch_cuda, ctx, pkg):
    sy, lib = pkg.synth, pkg._lib
    codes = sy.gen_codes(20, 30, 5)
    pwms, lens = sy.gen_pwm_bank(6, 6, len_lo=5, len_hi=8, alpha=0.6)
    bank = sy.pad_bank(pwms, lens)
    onehot = sy.codes_to_onehot(codes)
    a = ctx.pwm_scan(bank, lens, onehot.astype(np.float16), lib.DATA_ONEHOT_F16, 20, 30, False)
    b = ctx.pwm_scan(bank, lens, onehot, lib.DATA_ONEHOT_F32, 20, 30, False)
    c = ctx.pwm_scan(bank, lens, codes, lib.DATA_CODES_U8, 20, 30, False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(c[0], b[0]) and len(a[0]) > 0
    bad = onehot.copy()
    bad[3, 5] = 0.5
    with pytest.raises(lib.MotifsError) as e:
        ctx.pwm_scan(bank, lens, bad, lib.DATA_ONEHOT_F32, 20, 30, False)
    assert e.value.code == lib.ERR_NOT_ONEHOT
    nb = bank.copy()
    nb[0, 0, 0] = np.inf
    with pytest.raises(lib.MotifsError) as e:
        ctx.pwm_scan(nb, lens, onehot, lib.DATA_ONEHOT_F32, 20, 30, False)
    assert e.value.code == lib.ERR_NONFINITE
    with pytest.raises(lib.MotifsError) as e:
        ctx.pwm_scan(bank, lens, onehot, lib.DATA_ONEHOT_F32, 20, 30, False, cap=max(len(b[0]) - 1, 1))
    assert e.value.code == lib.ERR_BUFFER_TOO_SMALL
    big = np.zeros((lib.SCAN_MAX_LEN + 1, 4, 2), dtype=np.float16)     # longer than the read, longer than any template: no window, no hit
    f, _ = ctx.pwm_scan(big, np.array([lib.SCAN_MAX_LEN + 1] * 2), onehot, lib.DATA_ONEHOT_F32, 20, 30, False)
    assert len(f) == 0


def test_full_size_properties(torch_cuda, ctx, pkg):
    """BASELINE configs[1] size (100k x 200 bp, 200 PWMs of length 12): the oracle cannot finish this
    in seconds, so check size-independent properties + an oracle comparison on sampled sequences."""
    torch = torch_cuda
    sy, lib = pkg.synth, pkg._lib
    N, L, K = 100_000, 200, 200
    codes = sy.gen_codes(N, L, sy.SEED_BASE + 2, n_plant=5, k=12)
    pwms, lens = sy.gen_pwm_bank(K, sy.SEED_BASE + 2, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    h, s, counts = dev_scan_hits(torch, ctx, pkg, bank, lens, codes, False, lib.SCAN_BATCH, want_counts=True)
    n = len(h)
    assert n > 1_000_000
    # (1) histogram == record counts
    assert np.array_equal(counts, np.bincount(h[:, 0] - 1, minlength=K))
    # (2) reference order: (batch, l, n, m) strictly increasing
    key = (((h[:, 1].astype(np.int64) - 1) // lib.SCAN_BATCH) * 1000 + h[:, 2]) * (N + 1) + h[:, 1]
    key = key * (K + 1) + h[:, 0]
    assert np.all(np.diff(key) > 0)
    # (3) every score is a positive finite half, every l in range
    sf = s.view(np.float16)
    assert np.all(sf > 0) and np.all(np.isfinite(sf)) and h[:, 2].max() <= L - 12 + 1
    # (4) sampled sequences against the oracle
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(N, size=40, replace=False))
    g = so.scan_gather(bank, lens, codes[pick])          # (Lout, 40, K)
    lo, no, ko = np.nonzero(g > 0)
    want = {(int(k) + 1, int(pick[nn]) + 1, int(l) + 1): g[l, nn, k] for l, nn, k in zip(lo, no, ko)}
    sel = np.isin(h[:, 1], pick + 1)
    got = {(int(a), int(b), int(c)): v for (a, b, c), v in zip(h[sel], sf[sel])}
    assert got.keys() == want.keys()
    assert all(got[k_].view(np.uint16) == want[k_].view(np.uint16) for k_ in want)


def test_cfg5_shape_small_n(torch_cuda, ctx, pkg):
    """BASELINE configs[4] shape (1000 bp, 2048 PWMs of length 8..20 -> 16 chunks, LEN=20 template) on a few reads."""
    sy = pkg.synth
    N, L, K = 4, 1000, 2048
    codes = sy.gen_codes(N, L, sy.SEED_BASE + 5, n_plant=3, k=12)
    pwms, lens = sy.gen_pwm_bank(K, sy.SEED_BASE + 5, len_lo=8, len_hi=20, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    for rc in (False, True):
        h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, pkg._lib.SCAN_BATCH, want_counts=True)
        oh, os_ = oracle_hits(pkg, bank, lens, codes, rc, pkg._lib.SCAN_BATCH)
        assert len(oh) > 1000
        assert np.array_equal(h, oh) and np.array_equal(s, os_)
        assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))


def test_dense_rows_take_the_rescoring_path(torch_cuda, ctx, pkg):
    """All-positive PWMs: every window of every PWM is a hit, far more per row of cells than the staging
    slots hold, so records come from the slow path of emit_records; mixed with sparse PWMs in one bank."""
    sy = pkg.synth
    N, L, K = 90, 50, 150
    codes = sy.gen_codes(N, L, 77, n_plant=2, k=8)
    codes[3, 10] = 4
    pwms, lens = sy.gen_pwm_bank(K, 78, len_lo=6, len_hi=12, alpha=0.4)
    rng = np.random.default_rng(5)
    for k in range(0, K, 2):                     # every other PWM: strictly positive log-odds
        pwms[k] = np.abs(pwms[k]) + rng.uniform(0.01, 0.5, size=pwms[k].shape).astype(pwms[k].dtype)
    bank = sy.pad_bank(pwms, lens)
    for rc in (False, True):
        h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, 40, want_counts=True)
        oh, os_ = oracle_hits(pkg, bank, lens, codes, rc, 40)
        assert len(oh) > N * 20 * (K // 2)
        assert np.array_equal(h, oh) and np.array_equal(s, os_)
        assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))


def test_short_buffer_reports_needed_count(torch_cuda, ctx, pkg):
    sy = pkg.synth
    lib = pkg._lib
    N, L, K = 64, 60, 40
    codes = sy.gen_codes(N, L, 9, n_plant=2, k=8)
    pwms, lens = sy.gen_pwm_bank(K, 10, len_lo=8, len_hi=8, alpha=0.5)
    bank = sy.pad_bank(pwms, lens)
    h, s = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, False, 16)
    n = len(h)
    assert n > 10
    raw = torch_cuda.from_numpy(codes).cuda()
    dcodes = torch_cuda.zeros(lib.Context.codes_bytes(N, L), dtype=torch_cuda.uint8, device="cuda")
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    cap = n // 2
    hits = torch_cuda.zeros((n, 3), dtype=torch_cuda.int32, device="cuda")
    sc = torch_cuda.zeros(n, dtype=torch_cuda.int16, device="cuda")
    with pytest.raises(lib.MotifsError) as ei:
        ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, False, hits.data_ptr(), sc.data_ptr(), cap, batch=16)
    assert ei.value.code == lib.ERR_BUFFER_TOO_SMALL
    needed = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, False, hits.data_ptr(), sc.data_ptr(), cap, batch=16,
                                   allow_small=True)
    assert needed == n
    ctx.synchronize()
    assert not hits[cap:].any() and not sc[cap:].any(), "records past cap were written"


def _both_paths(torch, ctx, lib, monkeypatch, bank, lens, codes, K, min_hits):
    N, L = codes.shape
    monkeypatch.setenv("MOTIFS_SCAN_VALU", "1")
    exhaustive = lib.Context(0)
    monkeypatch.delenv("MOTIFS_SCAN_VALU")
    try:
        raw = torch.from_numpy(codes).cuda()
        dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
        ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
        ctx.synchronize()
        for rc in (0, 1):
            out = []
            for cx in (ctx, exhaustive):
                n = cx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, rc, None, None, 0)
                hits = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
                sc = torch.zeros(n, dtype=torch.int16, device="cuda")
                cnt = torch.zeros(K, dtype=torch.int64, device="cuda")
                assert cx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, rc, hits.data_ptr(), sc.data_ptr(), n,
                                            counts_ptr=cnt.data_ptr()) == n
                cx.synchronize()
                out.append((n, hits, sc, cnt))
            (n0, h0, s0, c0), (n1, h1, s1, c1) = out
            assert n0 == n1 and n0 > min_hits
            assert torch.equal(h0, h1) and torch.equal(s0, s1) and torch.equal(c0, c1)
            del out, h0, h1, s0, s1
    finally:
        exhaustive.close()


def test_two_implementations_agree_mixed_lengths(torch_cuda, ctx, pkg, monkeypatch):
    """Mixed PWM lengths (6..20: the LEN = 20 templates, three chunks of PWMs, per-PWM last valid starts), a
    length that is not a multiple of 4, a few all-zero columns: default path against the exhaustive kernel."""
    sy, lib = pkg.synth, pkg._lib
    N, L, K = 20_000, 303, 300
    codes = sy.gen_codes(N, L, 9090, n_plant=5, k=12)
    rng = np.random.default_rng(9)
    codes[rng.integers(0, N, 50), rng.integers(0, L, 50)] = 4
    pwms, lens = sy.gen_pwm_bank(K, 9091, len_lo=6, len_hi=20, alpha=0.35)
    _both_paths(torch_cuda, ctx, lib, monkeypatch, sy.pad_bank(pwms, lens), lens, codes, K, 1_000_000)


def test_two_implementations_agree_at_full_size(torch_cuda, ctx, pkg, monkeypatch):
    """BASELINE configs[1] size, both strands: the default path (matrix-core candidate filter + exact re-scoring
    of ~1 % of the pairs) against the exhaustive kernel that evaluates every window of every PWM in sequential
    binary16 (`MOTIFS_SCAN_VALU=1`, read when a context is created).  Records, scores and histograms must be
    identical: an exhaustive check of the candidate bound eps_k at full size."""
    sy, lib = pkg.synth, pkg._lib
    N, L, K = 100_000, 200, 200
    codes = sy.gen_codes(N, L, sy.SEED_BASE + 2, n_plant=5, k=12)
    pwms, lens = sy.gen_pwm_bank(K, sy.SEED_BASE + 2, alpha=0.3)
    _both_paths(torch_cuda, ctx, lib, monkeypatch, sy.pad_bank(pwms, lens), lens, codes, K, 20_000_000)


@pytest.mark.parametrize("scale,why", [(256.0, "slack too large for the uniform-slack kernel: per-PWM eps in C"),
                                       (4096.0, "partial sums overflow binary16: every window stays a candidate"),
                                       (2.0 ** -16, "subnormal weights: the bank cannot be rescaled exactly")])
def test_banks_outside_the_uniform_slack_range(torch_cuda, ctx, pkg, scale, why):
    """Banks whose magnitudes push the candidate kernel off its fast path (pack_mfma, scan_api.hip): the records
    must still be the reference's, Inf scores and NaN (Inf - Inf) windows included."""
    sy = pkg.synth
    N, L, K = 40, 48, 24
    codes = sy.gen_codes(N, L, 401, n_plant=2, k=8)
    codes[5, 7] = 4
    pwms, lens = sy.gen_pwm_bank(K, 402, len_lo=6, len_hi=12, alpha=0.5)
    pwms = [(np.asarray(p, dtype=np.float32) * scale).astype(np.float16) for p in pwms]
    assert all(np.isfinite(p).all() for p in pwms)
    bank = sy.pad_bank(pwms, lens)
    with np.errstate(over="ignore", invalid="ignore"):
        for rc in (False, True):
            h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, 16, want_counts=True)
            oh, os_ = oracle_hits(pkg, bank, lens, codes, rc, 16)
            assert len(oh) > 0
            assert np.array_equal(h, oh) and np.array_equal(s, os_), why
            assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))
    # the dense tensor of the same bank (K % 8 == 0: matrix-core path)
    torch = torch_cuda
    want = so.greedy_search(bank, lens, sy.codes_to_onehot(codes).astype(np.float16))
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(pkg._lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    out = torch.full((4 * L, N, K), 0x7BFF, dtype=torch.int16, device="cuda")
    ctx.encode_dev(raw.data_ptr(), pkg._lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, out.data_ptr(), 4 * L)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint16), want.view(np.uint16))


def test_random_shapes_against_oracle(torch_cuda, ctx, pkg):
    """A seeded sweep over odd shapes (K across tile and chunk boundaries, mixed PWM lengths, L barely above the
    longest PWM, batches that do not divide N, all-zero columns, both strands): records, scores and histograms
    against the oracle, and the dense tensor where it is small enough."""
    sy = pkg.synth
    rng = np.random.default_rng(20261004)
    for trial in range(36):
        K = int(rng.choice([1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 128, 129, 130, 200, 257]))
        hi = int(rng.choice([4, 8, 9, 12, 13, 16, 20, 21, 24, 29, 32]))
        lo = int(rng.integers(max(1, hi - 6), hi + 1))
        L = int(rng.integers(hi, hi + 70))
        N = int(rng.integers(1, 90))
        batch = int(rng.choice([1, 3, 7, 16, 50, 5000]))
        alpha = float(rng.uniform(0.25, 0.9))
        codes = sy.gen_codes(N, L, 7000 + trial, n_plant=2, k=min(8, L))
        for _ in range(int(rng.integers(0, 4))):
            codes[int(rng.integers(0, N)), int(rng.integers(0, L))] = 4
        pwms, lens = sy.gen_pwm_bank(K, 8000 + trial, len_lo=lo, len_hi=hi, alpha=alpha)
        bank = sy.pad_bank(pwms, lens)
        rc = bool(trial & 1)
        tag = f"trial {trial}: N={N} L={L} K={K} len {lo}..{hi} batch={batch} rc={rc}"
        h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, batch, want_counts=True)
        oh, os_ = oracle_hits(pkg, bank, lens, codes, rc, batch)
        assert np.array_equal(h, oh), tag
        assert np.array_equal(s, os_), tag
        assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K) if len(oh) else np.zeros(K, np.int64)), tag
        if not rc and N * L * K < 400_000:
            torch = torch_cuda
            want = so.greedy_search(bank, lens, sy.codes_to_onehot(codes).astype(np.float16))
            raw = torch.from_numpy(codes).cuda()
            dcodes = torch.zeros(pkg._lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
            out = torch.full((4 * L, N, K), 0x7BFF, dtype=torch.int16, device="cuda")
            ctx.encode_dev(raw.data_ptr(), pkg._lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
            ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, out.data_ptr(), 4 * L)
            ctx.synchronize()
            assert np.array_equal(out.cpu().numpy().view(np.uint16), want.view(np.uint16)), tag
