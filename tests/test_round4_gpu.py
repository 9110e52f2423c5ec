"""GPU tests added in round 4 (through the C ABI).

Chunk groups: banks whose binary16 re-scoring table does not fit one block's LDS (512 PWMs of 20 positions = 104 KB,
2048 PWMs = 418 KB) are re-scored group by group (1, 2 or 4 chunks of 128 PWMs per group, each block holding one group's
slice of the table in LDS) and the record order (findall's: start l, read, PWM; _h3_1_alignment.jl:82) is restored from the
staged words.  `MOTIFS_CG_CHUNKS=c` forces that path on banks of any size, so every group size meets the oracle on the
shapes the other scan tests use; the default (`auto`: chunk groups only when not even one 16-wave block per CU can hold the table) is
checked at the BASELINE configs[3] / configs[4] bank shapes."""
import os

import numpy as np
import pytest

from test_round2_gpu import fast_oracle_hits
from test_scan_gpu import dev_scan_hits, oracle_hits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


@pytest.fixture(scope="module", params=[1, 2, 4])
def cg_ctx(request, pkg):
    """A context whose hit-record scans take the chunk-group path with groups of `param` chunks wherever a bank can."""
    import os

    old = os.environ.get("MOTIFS_CG_CHUNKS")
    os.environ["MOTIFS_CG_CHUNKS"] = str(request.param)
    try:
        c = pkg._lib.Context(0)
    finally:
        if old is None:
            del os.environ["MOTIFS_CG_CHUNKS"]
        else:
            os.environ["MOTIFS_CG_CHUNKS"] = old
    c.cg_param = request.param
    yield c
    c.close()


def _expect_cg(c, nch):
    """The group size the forced context must have used for a bank of `nch` chunks (0: the bank cannot take it)."""
    g = c.cg_param
    return g if nch % g == 0 and (512 // g) * (-(-nch // g)) <= 2048 else 0


# K -> chunks of 128 PWMs: 100 -> 1, 200 -> 2, 300 -> 3, 500 -> 4, 700 -> 6, 1000 -> 8
@pytest.mark.parametrize("N,L,K,lo,hi,batch,alpha", [
    (70, 64, 100, 8, 12, 16, 0.4),
    (300, 100, 200, 12, 12, 64, 0.35),
    (41, 90, 300, 6, 16, 5000, 0.5),
    (230, 120, 500, 14, 20, 100, 0.3),      # rows of 100 reads: partial rows, several parts per batch
    (1100, 60, 700, 8, 20, 600, 0.4),       # 600-read batches: rows of 512 / 256 / 128 reads + a short one
    (97, 77, 1000, 8, 8, 33, 0.45),
])
def test_forced_chunk_groups_match_the_oracle(torch_cuda, cg_ctx, pkg, N, L, K, lo, hi, batch, alpha):
    sy = pkg.synth
    codes = sy.gen_codes(N, L, 4100 + K, n_plant=3, k=min(10, L))
    codes[N // 2, L // 3] = 4
    pwms, lens = sy.gen_pwm_bank(K, 4200 + K, len_lo=lo, len_hi=hi, alpha=alpha)
    bank = sy.pad_bank(pwms, lens)
    nch = -(-K // 128)
    for rc in (False, True):
        h, s, counts = dev_scan_hits(torch_cuda, cg_ctx, pkg, bank, lens, codes, rc, batch, want_counts=True)
        plan = cg_ctx.scan_plan()
        assert plan["cg_chunks"] == _expect_cg(cg_ctx, nch), plan
        if plan["cg_chunks"]:
            assert plan["compact"] and plan["cg_groups"] == -(-nch // plan["cg_chunks"])
        oh, os_ = fast_oracle_hits(bank, lens, codes, rc, batch)
        assert len(oh) > 200
        assert np.array_equal(h, oh) and np.array_equal(s, os_)
        assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))


def test_forced_chunk_groups_overflowing_rows(torch_cuda, cg_ctx, pkg):
    """Every other PWM strictly positive: each of its windows is a hit, far more per (row, group) than the staging slots
    hold, so those rows are re-scored in record order by emit_records_cg's slow path; sparse PWMs share the bank."""
    sy = pkg.synth
    N, L, K = 130, 50, 512
    codes = sy.gen_codes(N, L, 77, n_plant=2, k=8)
    codes[3, 10] = 4
    pwms, lens = sy.gen_pwm_bank(K, 78, len_lo=6, len_hi=12, alpha=0.4)
    rng = np.random.default_rng(5)
    for k in range(0, K, 2):
        pwms[k] = np.abs(pwms[k]) + rng.uniform(0.01, 0.5, size=pwms[k].shape).astype(pwms[k].dtype)
    bank = sy.pad_bank(pwms, lens)
    for rc in (False, True):
        h, s, counts = dev_scan_hits(torch_cuda, cg_ctx, pkg, bank, lens, codes, rc, 64, want_counts=True)
        assert cg_ctx.scan_plan()["cg_chunks"] == cg_ctx.cg_param
        oh, os_ = fast_oracle_hits(bank, lens, codes, rc, 64)
        assert len(oh) > N * 20 * (K // 2)
        assert np.array_equal(h, oh) and np.array_equal(s, os_)
        assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))


def test_forced_chunk_groups_short_buffer_and_super_batches(torch_cuda, cg_ctx, pkg):
    """cap below the record count: the needed count comes back and nothing is written past cap; a workspace bound that
    forces several launches gives the single-launch records."""
    sy, lib, torch = pkg.synth, pkg._lib, torch_cuda
    N, L, K = 900, 60, 512
    codes = sy.gen_codes(N, L, 9, n_plant=2, k=8)
    pwms, lens = sy.gen_pwm_bank(K, 10, len_lo=8, len_hi=16, alpha=0.45)
    bank = sy.pad_bank(pwms, lens)
    h, s = dev_scan_hits(torch, cg_ctx, pkg, bank, lens, codes, False, 100)
    n = len(h)
    assert n > 5000 and cg_ctx.scan_plan()["launches"] == 1
    oh, os_ = fast_oracle_hits(bank, lens, codes, False, 100)
    assert np.array_equal(h, oh) and np.array_equal(s, os_)
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    cg_ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    cap = n // 2
    hits = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    sc = torch.zeros(n, dtype=torch.int16, device="cuda")
    with pytest.raises(lib.MotifsError) as ei:
        cg_ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, False, hits.data_ptr(), sc.data_ptr(), cap, batch=100)
    assert ei.value.code == lib.ERR_BUFFER_TOO_SMALL
    needed = cg_ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), N, L, False, hits.data_ptr(), sc.data_ptr(), cap, batch=100,
                                      allow_small=True)
    assert needed == n
    cg_ctx.synchronize()
    assert not hits[cap:].any() and not sc[cap:].any(), "records past cap were written"
    assert np.array_equal(hits[:cap].cpu().numpy().astype(np.uint32), oh[:cap])
    try:
        per_batch = (L - 8 + 1) * 100 * 4 * 16
        cg_ctx.set_workspace_limit(3 * per_batch)          # cells + entries + staged words of ~2 batches per launch
        h2, s2 = dev_scan_hits(torch, cg_ctx, pkg, bank, lens, codes, False, 100)
        assert cg_ctx.scan_plan()["launches"] >= 4
        assert np.array_equal(h2, oh) and np.array_equal(s2, os_)
    finally:
        cg_ctx.set_workspace_limit(0)


def test_both_strands_entry_with_forced_chunk_groups(torch_cuda, cg_ctx, pkg):
    sy, lib, torch = pkg.synth, pkg._lib, torch_cuda
    N, L, K = 600, 100, 256
    codes = sy.gen_codes(N, L, 31, n_plant=3, k=10)
    pwms, lens = sy.gen_pwm_bank(K, 32, len_lo=10, len_hi=12, alpha=0.35)
    bank = sy.pad_bank(pwms, lens)
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    cg_ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    need = cg_ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0, batch=250)
    cap = max(need)
    hits = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
    scs = [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
    cnt = torch.zeros((2, K), dtype=torch.int64, device="cuda")
    got = cg_ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [t.data_ptr() for t in hits], [t.data_ptr() for t in scs],
                                        cap, batch=250, counts_ptr=cnt.data_ptr())
    cg_ctx.synchronize()
    assert got == need and cg_ctx.scan_plan()["cg_chunks"] == _expect_cg(cg_ctx, 2)
    for rc in (0, 1):
        oh, os_ = fast_oracle_hits(bank, lens, codes, bool(rc), 250)
        assert np.array_equal(hits[rc][:got[rc]].cpu().numpy().astype(np.uint32), oh)
        assert np.array_equal(scs[rc][:got[rc]].cpu().numpy().view(np.uint16), os_)
        assert np.array_equal(cnt[rc].cpu().numpy(), np.bincount(oh[:, 0] - 1, minlength=K))


@pytest.mark.parametrize("K,L,lo,hi,N,batch", [(512, 500, 20, 20, 300, 5000), (2048, 1000, 8, 20, 40, 16)])
def test_large_banks_keep_their_table_in_lds(torch_cuda, ctx, pkg, K, L, lo, hi, N, batch):
    """BASELINE configs[3] / configs[4] bank shapes on the default context.  Neither table fits the 64 KB an 8-wave block may take:
    512 PWMs of 20 positions (104 KB) go to ONE 16-wave block per CU with the whole table (no chunk groups: the records need no
    reordering), 2048 PWMs (418 KB) go group by group.  The records are the CPU port's (and, on the first reads, the literal
    restatement's)."""
    sy = pkg.synth
    codes = sy.gen_codes(N, L, 50400 + K, n_plant=5, k=20)
    pwms, lens = sy.gen_pwm_bank(K, 50400 + K, len_lo=lo, len_hi=hi, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    for rc in (False, True):
        h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, batch, want_counts=True)
        plan = ctx.scan_plan()
        if K == 512:
            assert plan["compact"] and plan["cg_chunks"] == 0, plan
        else:
            assert plan["compact"] and plan["cg_chunks"] in (1, 2, 4) and plan["cg_groups"] == (K // 128) // plan["cg_chunks"], plan
        oh, os_ = fast_oracle_hits(bank, lens, codes, rc, batch)
        assert len(oh) > 20000
        assert np.array_equal(h, oh) and np.array_equal(s, os_)
        assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))
    lh, ls = oracle_hits(pkg, bank, lens, codes[:3], True, batch)
    keep = h[:, 1] <= 3
    order = np.lexsort((h[keep][:, 0], h[keep][:, 1], h[keep][:, 2]))
    assert np.array_equal(h[keep][order], lh) and np.array_equal(s[keep][order], ls)


# ---- BASELINE configs[4] at the size of ONE rank's real shard of the 8-GPU job -------------------------------------------------
def _check_records_in_chunks(torch, hits, hsc, got, cnt_row, n, L, K, B, lens_t, chunk=1 << 27):
    """Size-independent properties of a record list too large for whole-array int64 temporaries: every record valid, every
    window inside its read, positive binary16 score bits, strictly ascending (batch, l, n, m) keys (also across chunk edges),
    histogram == the counts the scan returned."""
    hist = torch.zeros(K, dtype=torch.int64, device="cuda")
    last_key = None
    for c0 in range(0, got, chunk):
        f = hits[c0:min(got, c0 + chunk)].to(torch.int64)
        m, nn, l = f[:, 0], f[:, 1], f[:, 2]
        assert int(m.min()) >= 1 and int(m.max()) <= K and int(nn.min()) >= 1 and int(nn.max()) <= n and int(l.min()) >= 1
        assert bool((l <= L - lens_t[m - 1] + 1).all())
        assert bool((hsc[c0:min(got, c0 + chunk)] > 0).all())
        key = (((nn - 1) // B * (L + 1) + l) * B + (nn - 1) % B) * (K + 1) + m
        assert bool((key[1:] > key[:-1]).all()), "records are not in the reference's order"
        if last_key is not None:
            assert int(key[0]) > last_key, "records are not in the reference's order across a chunk edge"
        last_key = int(key[-1])
        hist += torch.bincount(m - 1, minlength=K)
        del f, m, nn, l, key
    assert torch.equal(hist, cnt_row)


def test_cfg4_rank_shard_full_size_properties(torch_cuda, ctx, pkg):
    """125 000 reads x 1000 bp vs 2048 PWMs of 8-20 positions, default 8 GiB workspace: what one rank of the 8-GPU job scans at
    BASELINE configs[4] (round 3 ran a fifth of it).  ~2e9 records per strand: record offsets, n_out and the row / entry indices
    far above anything the other tests reach.  Totals == the count pass, histogram == records, keys strictly ascending, and the
    reads on both sides of every super-batch edge against the CPU port."""
    torch = torch_cuda
    lib, sy = pkg._lib, pkg.synth
    n, L, K = 125_000, 1000, 2048
    pwms, lens = sy.gen_pwm_bank(K, 4711 + K, len_lo=8, len_hi=20, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    codes = sy.gen_codes(n, L, 4799, n_plant=5, k=20)
    codes[7, 11] = 4
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(n, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, n, L, dcodes.data_ptr())
    del raw
    need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), n, L, None, None, 0)
    assert min(need) > 1_500_000_000, need
    cap = max(need)
    B = lib.SCAN_BATCH
    lens_t = torch.from_numpy(lens).cuda()
    for rc in (0, 1):                                                     # one strand's buffers at a time (28 GB each)
        hits = torch.empty((cap, 3), dtype=torch.int32, device="cuda")
        hsc = torch.empty(cap, dtype=torch.int16, device="cuda")
        cnt = torch.zeros(K, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        got = ctx.pwm_scan_hits_dev(bank, lens, dcodes.data_ptr(), n, L, rc, hits.data_ptr(), hsc.data_ptr(), cap, counts_ptr=cnt.data_ptr())
        ctx.synchronize()
        assert got == need[rc]
        plan = ctx.scan_plan()
        assert plan["cg_chunks"] in (1, 2, 4) and plan["launches"] >= 7, plan
        per_launch = -(-(n // B) // plan["launches"]) * B                 # reads per super-batch launch (whole ordering batches)
        edges = sorted({0, n - 3} | {e for s in range(per_launch, n, per_launch) for e in (s - 3, s)})
        _check_records_in_chunks(torch, hits, hsc, got, cnt, n, L, K, B, lens_t)
        # sampled reads: all records of reads r0+1..r0+3 (1-based), found by bisection on the batch's l = 1 block being sorted by n
        nn_all = hits[:got, 1]
        for r0 in edges:
            sel = torch.nonzero((nn_all > r0) & (nn_all <= r0 + 3)).squeeze(1)
            mine = hits[sel].cpu().numpy().astype(np.uint32)
            mys = hsc[sel].cpu().numpy().view(np.uint16)
            oh, os_ = fast_oracle_hits(bank, lens, codes[r0:r0 + 3], bool(rc), B)
            oh = oh.copy()
            oh[:, 1] += r0
            order = np.lexsort((mine[:, 0], mine[:, 1], mine[:, 2]))
            assert np.array_equal(mine[order], oh) and np.array_equal(mys[order], os_), (rc, r0)
        del hits, hsc, nn_all
        torch.cuda.empty_cache()


# ---- segments of window tiles in the four-reads candidate kernel ---------------------------------------------------------------
@pytest.mark.parametrize("N,L,K,lo,hi,batch", [
    (37, 333, 100, 12, 12, 16),       # 46 KB of one-hot images against the 39 KB a block may take at four blocks per CU: two segments
    (9, 1501, 260, 8, 16, 5000),      # a long read, mixed lengths, 188 window tiles (the last one partial) in several segments
    (64, 200, 200, 12, 12, 5000),     # a launch of 8 blocks: segments to fill the CU slots
    (203, 1000, 130, 17, 20, 50),     # configs[4]'s read length, reads not a multiple of 4, several ordering batches
    (3, 20011, 60, 9, 12, 5000),      # a contig-sized read: 2 500 window tiles in ~70 segments
])
def test_segments_of_window_tiles(torch_cuda, ctx, pkg, N, L, K, lo, hi, batch):
    """scan_cand_kernel_q's blocks take a segment of a read's window tiles when the whole one-hot images do not fit beside the CU's other
    blocks (long reads) or the launch does not fill the CU slots once; the cells they write - and so the records - are those of
    whole reads: the single-strand entry and the both-strands entry (one candidate launch for the two banks) against the CPU port."""
    sy, lib, torch = pkg.synth, pkg._lib, torch_cuda
    codes = sy.gen_codes(N, L, 61000 + L, n_plant=4, k=min(12, lo))
    codes[N // 3, L // 2] = 4
    codes[N - 1, L - 1] = 4
    pwms, lens = sy.gen_pwm_bank(K, 62000 + L, len_lo=lo, len_hi=hi, alpha=0.35)
    bank = sy.pad_bank(pwms, lens)
    want = [fast_oracle_hits(bank, lens, codes, bool(rc), batch) for rc in (0, 1)]
    assert min(len(w[0]) for w in want) > 100
    for rc in (0, 1):
        h, s, counts = dev_scan_hits(torch, ctx, pkg, bank, lens, codes, bool(rc), batch, want_counts=True)
        assert ctx.scan_plan()["compact"]
        assert np.array_equal(h, want[rc][0]) and np.array_equal(s, want[rc][1])
        assert np.array_equal(counts, np.bincount(want[rc][0][:, 0] - 1, minlength=K))
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0, batch=batch)
    assert list(need) == [len(w[0]) for w in want]
    cap = max(need)
    hits = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
    scs = [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
    got = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [t.data_ptr() for t in hits], [t.data_ptr() for t in scs], cap, batch=batch)
    ctx.synchronize()
    assert got == need
    for rc in (0, 1):
        assert np.array_equal(hits[rc][:got[rc]].cpu().numpy().astype(np.uint32), want[rc][0])
        assert np.array_equal(scs[rc][:got[rc]].cpu().numpy().view(np.uint16), want[rc][1])


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_with_long_reads(torch_cuda, ctx, pkg, seed):
    """Shapes drawn at random around the segment boundaries of the candidate kernel (reads of 150 - 3000 positions, PWMs of every template length
    up to 20, banks of one to five chunks, ordering batches that do and do not divide the reads): both strands against the CPU port."""
    rng = np.random.default_rng(8800 + seed)
    L = int(rng.integers(150, 3000))
    lo = int(rng.integers(6, 18))
    hi = int(rng.integers(lo, 21))
    K = int(rng.integers(20, 600))
    N = int(rng.integers(1, max(2, 120000 // L)))
    batch = int(rng.choice([5000, 7, 16, 33]))
    sy = pkg.synth
    codes = sy.gen_codes(N, L, 8900 + seed, n_plant=3, k=min(10, lo))
    codes[N // 2, L // 2] = 4
    pwms, lens = sy.gen_pwm_bank(K, 9000 + seed, len_lo=lo, len_hi=hi, alpha=0.35)
    bank = sy.pad_bank(pwms, lens)
    for rc in (False, True):
        h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, batch, want_counts=True)
        oh, os_ = fast_oracle_hits(bank, lens, codes, rc, batch)
        assert np.array_equal(h, oh) and np.array_equal(s, os_), (N, L, K, lo, hi, batch)
        assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))


def test_records_in_stream_order(torch_cuda, pkg):
    """motifs_ctx_set_records_in_stream_order: the both-strands call returns when the totals are known, the records follow in stream
    order.  Two shards scanned back to back without a host wait in between, into their own buffers, then one synchronize: both record sets
    are the CPU port's (the second call's kernels queue behind the first call's record writes and reuse its workspaces)."""
    sy, lib, torch = pkg.synth, pkg._lib, torch_cuda
    c = lib.Context(0)
    c.set_records_in_stream_order(True)
    try:
        N, L, K = 6000, 120, 200
        pwms, lens = sy.gen_pwm_bank(K, 9300, len_lo=12, len_hi=12, alpha=0.3)
        bank = sy.pad_bank(pwms, lens)
        shards = []
        for i in range(2):
            codes = sy.gen_codes(N, L, 9400 + i, n_plant=4, k=12)
            raw = torch.from_numpy(codes).cuda()
            dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
            need = c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
            cap = max(need) + 8
            hits = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
            scs = [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
            shards.append((codes, dcodes, need, cap, hits, scs))
        c.synchronize()
        torch.cuda.synchronize()
        got = []
        for codes, dcodes, need, cap, hits, scs in shards:          # no wait between the two calls
            got.append(c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [t.data_ptr() for t in hits], [t.data_ptr() for t in scs], cap))
        assert c.scan_plan()["launches"] == 1
        c.synchronize()
        for (codes, dcodes, need, cap, hits, scs), g in zip(shards, got):
            assert g == need
            for rc in (0, 1):
                oh, os_ = fast_oracle_hits(bank, lens, codes, bool(rc), 5000)
                assert len(oh) == g[rc] > 1000
                assert np.array_equal(hits[rc][:g[rc]].cpu().numpy().astype(np.uint32), oh)
                assert np.array_equal(scs[rc][:g[rc]].cpu().numpy().view(np.uint16), os_)
    finally:
        c.close()


def test_stream_order_mode_with_a_short_buffer(torch_cuda, pkg):
    """Records in stream order and a hit buffer that is too small: the call reports MOTIFS_ERR_BUFFER_TOO_SMALL with the needed counts only
    after the stream has drained - the first `cap` records of each strand are the CPU port's, nothing past `cap` was written - so the
    caller may free or re-allocate its buffers straight away (round-4 advisor)."""
    sy, lib, torch = pkg.synth, pkg._lib, torch_cuda
    c = lib.Context(0)
    c.set_records_in_stream_order(True)
    try:
        N, L, K = 6000, 120, 200
        pwms, lens = sy.gen_pwm_bank(K, 9300, len_lo=12, len_hi=12, alpha=0.3)
        bank = sy.pad_bank(pwms, lens)
        codes = sy.gen_codes(N, L, 9411, n_plant=4, k=12)
        raw = torch.from_numpy(codes).cuda()
        dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
        need = c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
        cap = min(need) // 3
        hits = [torch.full((cap + 64, 3), -1, dtype=torch.int32, device="cuda") for _ in range(2)]
        scs = [torch.full((cap + 64,), -1, dtype=torch.int16, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        with pytest.raises(lib.MotifsError) as ei:
            c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [t.data_ptr() for t in hits], [t.data_ptr() for t in scs], cap)
        assert ei.value.code == lib.ERR_BUFFER_TOO_SMALL
        # no synchronize here: the error path has waited
        for rc in (0, 1):
            oh, os_ = fast_oracle_hits(bank, lens, codes, bool(rc), 5000)
            h, s_ = hits[rc].cpu().numpy(), scs[rc].cpu().numpy()
            assert np.array_equal(h[:cap].astype(np.uint32), oh[:cap]) and np.array_equal(s_[:cap].view(np.uint16), os_[:cap])
            assert np.all(h[cap:] == -1) and np.all(s_[cap:] == -1)
    finally:
        c.close()



def test_workspace_limit_bounds_what_a_both_strands_scan_holds(torch_cuda, pkg):
    """motifs_ctx_set_workspace_limit: what a fresh context allocates for a both-strands scan stays under the bound (+ the small fixed
    buffers: bank, totals), also when every stage serves both strands in one launch - the pair plan holds staged words, row counts and
    offsets twice, which the geometry now counts (round-4 advisor: it sized the launch for one set).  The limit is chosen so that the pair
    plan just fits one super-batch; records as in the default plan."""
    sy, lib, torch = pkg.synth, pkg._lib, torch_cuda
    N, L, K = 20000, 200, 200
    pwms, lens = sy.gen_pwm_bank(K, 9500, len_lo=12, len_hi=12, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    codes = sy.gen_codes(N, L, 9501, n_plant=4, k=12)
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    Lout, nch, batch = L - 12 + 1, 2, 5000
    cells = Lout * batch * nch * 16
    staged = Lout * ((batch + 255) // 256) * (2 * 256 * nch) * 4
    rows = Lout * ((batch + 255) // 256) * 8
    pair_per_batch = 2 * (cells + cells // 4) + 2 * (staged + rows)
    ref = None
    for limit, want_launches in ((0, 1), (4 * pair_per_batch, 1), (4 * pair_per_batch - (1 << 20), None)):
        c = lib.Context(0)
        try:
            c.set_workspace_limit(limit)
            torch.cuda.synchronize()
            c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
            need = c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
            cap = max(need) + 8
            hits = [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
            scs = [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
            got = c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [t.data_ptr() for t in hits], [t.data_ptr() for t in scs], cap)
            c.synchronize()
            held = free0 - torch.cuda.mem_get_info()[0]
            if limit:
                assert held <= limit + (8 << 20), (held, limit)
            if want_launches is not None:
                assert c.scan_plan()["launches"] == want_launches
            rec = [(hits[rc][:got[rc]].cpu().numpy().copy(), scs[rc][:got[rc]].cpu().numpy().copy()) for rc in (0, 1)]
            if ref is None:
                ref = rec
            else:
                for rc in (0, 1):
                    assert np.array_equal(rec[rc][0], ref[rc][0]) and np.array_equal(rec[rc][1], ref[rc][1])
        finally:
            c.close()


def test_scans_on_poisoned_workspaces(torch_cuda, pkg):
    """MOTIFS_POISON_WS=1 (read once per process: a fresh one): every workspace a context allocates starts as 0xFF bytes.  Four awkward
    shapes - a short last ordering batch, reads not a multiple of 4 with an all-zero column and three chunks, a launch of one PWM tile on
    1001-position reads, PWMs of 24-40 positions (128-bit cells, no entries) - both strands against the CPU port: a kernel that reads a cell,
    an entry or a staging slot that nothing wrote would return a wrong record."""
    import subprocess
    import sys
    e = dict(os.environ, MOTIFS_POISON_WS="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "_poison_ws_helper.py")], env=e, timeout=600,
                       capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]



def test_histogram_sum_between_scans_in_stream_order_mode(torch_cuda, pkg):
    """motifs_hist_allreduce straight after the both-strands scan that wrote the counts, in stream-order mode (the call has returned while the records
    are still being written) and with the NEXT scan - whose candidate kernel zeroes the SAME counts buffer itself, no fill in front of it - queued
    right behind: three scans of different shards back to back on a private-stream context, each followed by the sum (a one-rank RCCL
    communicator: the values must come back unchanged) and by a device-side copy of the counts on the context's stream: every copy is the
    histogram of its own scan's records."""
    sy, lib, torch = pkg.synth, pkg._lib, torch_cuda
    c = lib.Context(0)
    c.set_records_in_stream_order(True)
    comm = None
    try:
        try:
            comm = lib.Comm(c, lib.Comm.unique_id(), 1, 0)
        except lib.MotifsError as e:
            pytest.skip(f"no RCCL communicator on this box: {e}")
        N, L, K = 7000, 120, 200
        pwms, lens = sy.gen_pwm_bank(K, 9600, len_lo=12, len_hi=12, alpha=0.3)
        bank = sy.pad_bank(pwms, lens)
        st = torch.cuda.ExternalStream(c.get_stream())
        counts = torch.zeros((2, K), dtype=torch.int64, device="cuda")
        shards, copies = [], []
        for i in range(3):
            codes = sy.gen_codes(N, L, 9610 + i, n_plant=4, k=12)
            raw = torch.from_numpy(codes).cuda()
            dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
            need = c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0)
            cap = max(need) + 8
            shards.append((dcodes, need, cap, [torch.zeros((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)],
                           [torch.zeros(cap, dtype=torch.int16, device="cuda") for _ in range(2)]))
            copies.append(torch.zeros_like(counts))
        c.synchronize()
        torch.cuda.synchronize()
        for (dcodes, need, cap, hits, scs), cp in zip(shards, copies):
            got = c.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [t.data_ptr() for t in hits], [t.data_ptr() for t in scs], cap,
                                           counts_ptr=counts.data_ptr())
            assert got == need and c.scan_plan()["launches"] == 1
            comm.hist_allreduce(counts.data_ptr(), K, 2)
            with torch.cuda.stream(st):
                cp.copy_(counts)                      # on the context's stream: behind the sum, in front of the next scan's zero fill
        c.synchronize()
        for (dcodes, need, cap, hits, scs), cp in zip(shards, copies):
            for rc in (0, 1):
                want = torch.bincount(hits[rc][:need[rc], 0].to(torch.int64) - 1, minlength=K)
                assert torch.equal(cp[rc], want)
    finally:
        if comm is not None:
            c.synchronize()
            comm.close()
        c.close()
