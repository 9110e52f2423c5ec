"""GPU parity tests added in round 2 (through the C ABI): PWMs longer than 32 positions, the super-batch chain of a
scan that does not fit its workspace, BASELINE configs[3] scan shape, the stream contract of the context, the
reference's Float32 one-hot training batch, and the RCCL communicator behind the ABI."""
import numpy as np
import pytest

from oracle import scan_oracle as so
from test_scan_gpu import dev_scan_hits, oracle_hits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def fast_oracle_hits(bank, lens, codes, rc, batch):
    """The vectorised CPU port (checked equal to the literal one by tests/test_oracle_scan.py)."""
    got = so.get_pos_scores_arr_fast(bank, lens, codes, rc=rc, batch_size=batch)
    if got is None:
        pytest.skip("host CPU lacks AVX2/F16C")
    f, s = got
    return np.stack([f["m"], f["n"], f["l"]], axis=1).astype(np.uint32), s.view(np.uint16)


def plant_consensus(codes, pwms, lens, seed, frac=0.7):
    """Write the best-scoring word of a random PWM (or its reverse complement) into a share of the reads, with a few
    point changes, so that long PWMs have hits on both strands."""
    rng = np.random.default_rng(seed)
    N, L = codes.shape
    for n in range(N):
        if rng.random() > frac:
            continue
        k = int(rng.integers(len(pwms)))
        if lens[k] > L:
            continue
        word = np.argmax(np.asarray(pwms[k], dtype=np.float32), axis=0).astype(np.uint8)     # (4, len) -> best base per position
        if rng.random() < 0.5:
            word = (3 - word)[::-1]
        for _ in range(int(rng.integers(0, 3))):
            word[int(rng.integers(len(word)))] = rng.integers(4)
        o = int(rng.integers(0, L - len(word) + 1))
        codes[n, o:o + len(word)] = word
    return codes


# ---- PWMs of 33..64 positions (the reference has no cap: _h3_1_alignment.jl:25-31, motif length d13 + h) ----
LONG_CASES = [
    # N, L, K, len_lo, len_hi, batch
    (40, 120, 16, 33, 40, 16),      # LEN = 40 template, K % 8 == 0: streamed dense path too
    (30, 150, 40, 41, 48, 5000),    # LEN = 48
    (25, 200, 9, 49, 64, 7),        # LEN = 64, odd K: dense falls back to zeros + scattered records
    (35, 100, 150, 20, 64, 10),     # mixed lengths across two chunks of 128 PWMs
    (12, 64, 8, 64, 64, 5),         # a single window per read
]


@pytest.mark.parametrize("N,L,K,lo,hi,batch", LONG_CASES)
@pytest.mark.parametrize("rc", [False, True])
def test_long_pwms_match_oracle(torch_cuda, ctx, pkg, N, L, K, lo, hi, batch, rc):
    sy = pkg.synth
    codes = sy.gen_codes(N, L, 900 + N + K)
    pwms, lens = sy.gen_pwm_bank(K, 700 + K, len_lo=lo, len_hi=hi, alpha=0.5)
    plant_consensus(codes, pwms, lens, 1 + K)
    codes[N // 3, L // 2] = 4
    bank = sy.pad_bank(pwms, lens)
    h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, batch, want_counts=True)
    oh, os_ = oracle_hits(pkg, bank, lens, codes, rc, batch)
    assert len(oh) > 0
    assert np.array_equal(h, oh), "hit records (m,n,l) or their order differ"
    assert np.array_equal(s, os_), "fp16 scores differ"
    assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))


@pytest.mark.parametrize("N,L,K,lo,hi", [(20, 120, 16, 33, 40), (15, 150, 9, 49, 64)])
def test_long_pwms_dense_tensor(torch_cuda, ctx, pkg, N, L, K, lo, hi):
    """a17's (K, N, ld_l) tensor for long PWMs, zeros included: the streamed form (K % 8 == 0) and the fallback."""
    torch = torch_cuda
    lib, sy = pkg._lib, pkg.synth
    codes = sy.gen_codes(N, L, 31 + K)
    pwms, lens = sy.gen_pwm_bank(K, 17 + K, len_lo=lo, len_hi=hi, alpha=0.5)
    plant_consensus(codes, pwms, lens, 2 + K)
    bank = sy.pad_bank(pwms, lens)
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    ld = L - int(lens.min()) + 3
    dense = torch.full((ld, N, K), 0x5555, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, dense.data_ptr(), ld)
    ctx.synchronize()
    want = so.scan_gather(bank, lens, codes, Lout=ld)
    assert (want > 0).sum() > 5
    assert np.array_equal(dense.cpu().numpy().view(np.uint16), want.view(np.uint16))


# ---- no cap on the PWM length (_h3_1_alignment.jl:25-31 has none; round 2 refused more than 64 positions) ----
VERY_LONG_CASES = [
    # N, L, K, len_lo, len_hi, batch
    (14, 160, 8, 65, 65, 5),        # one position past the last template size; K % 8 == 0: the streamed dense form too
    (12, 200, 11, 66, 120, 5000),   # mixed lengths up to 120, odd K
    (9, 260, 40, 20, 97, 4),        # short and very long PWMs in one bank (the short ones' windows end long before the padded length)
    (6, 130, 130, 100, 128, 3),     # two chunks of 128 PWMs
    (5, 104, 6, 88, 120, 5),        # PWMs nearly as long as the read (a few windows) and longer than it (none)
]


@pytest.mark.parametrize("N,L,K,lo,hi,batch", VERY_LONG_CASES)
@pytest.mark.parametrize("rc", [False, True])
def test_pwms_past_64_positions_match_oracle(torch_cuda, ctx, pkg, N, L, K, lo, hi, batch, rc):
    """The run-time-length kernels (scan_cand_kernel_g, stage_hits<0>, emit_records<0>) against the literal restatement."""
    sy = pkg.synth
    codes = sy.gen_codes(N, L, 1900 + N + K)
    pwms, lens = sy.gen_pwm_bank(K, 1700 + K, len_lo=lo, len_hi=hi, alpha=0.5)
    plant_consensus(codes, pwms, lens, 5 + K, frac=0.9)
    codes[N // 3, L // 2] = 4
    codes[N - 1, L - 1] = 4
    bank = sy.pad_bank(pwms, lens)
    h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, batch, want_counts=True)
    oh, os_ = oracle_hits(pkg, bank, lens, codes, rc, batch)
    assert len(oh) > 0
    assert np.array_equal(h, oh), "hit records (m,n,l) or their order differ"
    assert np.array_equal(s, os_), "fp16 scores differ"
    assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))


@pytest.mark.parametrize("N,L,K,lo,hi", [(10, 160, 8, 65, 80), (8, 200, 11, 66, 120)])
def test_pwms_past_64_positions_dense_tensor(torch_cuda, ctx, pkg, N, L, K, lo, hi):
    torch = torch_cuda
    lib, sy = pkg._lib, pkg.synth
    codes = sy.gen_codes(N, L, 131 + K)
    pwms, lens = sy.gen_pwm_bank(K, 117 + K, len_lo=lo, len_hi=hi, alpha=0.5)
    plant_consensus(codes, pwms, lens, 12 + K, frac=0.9)
    bank = sy.pad_bank(pwms, lens)
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
    ld = L - int(lens.min()) + 2
    dense = torch.full((ld, N, K), 0x5555, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.pwm_scan_dense_dev(bank, lens, dcodes.data_ptr(), N, L, dense.data_ptr(), ld)
    ctx.synchronize()
    want = so.scan_gather(bank, lens, codes, Lout=ld)
    assert (want > 0).sum() > 0
    assert np.array_equal(dense.cpu().numpy().view(np.uint16), want.view(np.uint16))


def test_host_entry_with_a_very_long_pwm(ctx, pkg):
    """What round 2 refused with MOTIFS_ERR_UNSUPPORTED: the ccall entry on a bank of 65-position PWMs."""
    sy = pkg.synth
    codes = sy.gen_codes(30, 80, 1)
    pwms, lens = sy.gen_pwm_bank(4, 1, len_lo=65, len_hi=65, alpha=0.5)
    plant_consensus(codes, pwms, lens, 3, frac=1.0)
    bank = sy.pad_bank(pwms, lens)
    for rc in (False, True):
        f, s = ctx.pwm_scan(bank, lens, sy.codes_to_onehot(codes), pkg._lib.DATA_ONEHOT_F32, 30, 80, rc)
        of, os_ = so.get_pos_scores_arr(bank, lens, sy.codes_to_onehot(codes), rc=rc)
        assert np.array_equal(f, of) and np.array_equal(s.view(np.uint16), os_.view(np.uint16))
    assert len(f) + 1 > 0


# ---- the super-batch chain (scan_api.hip: launch_no > 0, ping-pong totals, n0 + s0) ----
@pytest.mark.parametrize("rc", [False, True])
def test_super_batches_cross_the_workspace_limit(torch_cuda, pkg, rc):
    """A workspace bound small enough that the scan walks the reads in >= 4 super-batches of whole ordering batches
    (the 1M x 1000 bp x 2048-PWM scan of BASELINE configs[4] needs ~5 per 8 GiB): records, order, scores and histogram
    must not depend on the bound."""
    lib, sy = pkg._lib, pkg.synth
    N, L, K, batch = 1100, 90, 140, 100                      # 11 ordering batches, the last of them short
    codes = sy.gen_codes(N, L, 4242, n_plant=3, k=10)
    pwms, lens = sy.gen_pwm_bank(K, 77, len_lo=8, len_hi=12, alpha=0.35)
    bank = sy.pad_bank(pwms, lens)
    c = lib.Context(0)
    try:
        Lout = L - int(lens.min()) + 1
        per_batch = Lout * batch * 2 * 16 * 3                 # cells (2 chunks) + staged words, roughly
        c.set_workspace_limit(3 * per_batch)                  # ~3 batches per super-batch -> 4 launches
        h, s, counts = dev_scan_hits(torch_cuda, c, pkg, bank, lens, codes, rc, batch, n0=7, want_counts=True)
        c.set_workspace_limit(per_batch // 2)                 # below one batch: one batch per launch, 11 launches
        h1, s1, counts1 = dev_scan_hits(torch_cuda, c, pkg, bank, lens, codes, rc, batch, n0=7, want_counts=True)
        c.set_workspace_limit(0)                              # default: a single launch
        h0, s0, counts0 = dev_scan_hits(torch_cuda, c, pkg, bank, lens, codes, rc, batch, n0=7, want_counts=True)
    finally:
        c.close()
    oh, os_ = fast_oracle_hits(bank, lens, codes, rc, batch)
    oh[:, 1] += 7
    assert len(oh) > 10000
    for hh, ss, cc in ((h, s, counts), (h1, s1, counts1), (h0, s0, counts0)):
        assert np.array_equal(hh, oh) and np.array_equal(ss, os_)
        assert np.array_equal(cc, np.bincount(oh[:, 0] - 1, minlength=K))


# ---- BASELINE configs[3] scan shape: 512 PWMs of length 20 on 500 bp reads ----
@pytest.mark.parametrize("rc", [False, True])
def test_cfg3_scan_shape_matches_oracle(torch_cuda, ctx, pkg, rc):
    sy = pkg.synth
    N, L, K = 160, 500, 512
    codes = sy.gen_codes(N, L, 50403, n_plant=5, k=20)
    pwms, lens = sy.gen_pwm_bank(K, 50403, len_lo=20, len_hi=20, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    h, s, counts = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, 5000, want_counts=True)
    oh, os_ = fast_oracle_hits(bank, lens, codes, rc, 5000)
    assert len(oh) > 20000
    assert np.array_equal(h, oh) and np.array_equal(s, os_)
    assert np.array_equal(counts, np.bincount(oh[:, 0] - 1, minlength=K))
    # and the literal restatement on the first reads (dense (K, nb, 4L) tensor + findall)
    lh, ls = oracle_hits(pkg, bank, lens, codes[:6], rc, 5000)
    keep = h[:, 1] <= 6
    order = np.lexsort((h[keep][:, 0], h[keep][:, 1], h[keep][:, 2]))      # (l, n, k) order of the 6-read scan
    assert np.array_equal(h[keep][order], lh) and np.array_equal(s[keep][order], ls)


# ---- stream contract (ABI 2): NULL is HIP's null stream, get_stream returns what is in use ----
def test_set_stream_null_means_the_null_stream(torch_cuda, pkg):
    torch = torch_cuda
    lib, sy = pkg._lib, pkg.synth
    c = lib.Context(0)
    try:
        own = c.get_stream()
        assert own != 0                                       # a context starts on a private stream
        c.set_stream(0)
        assert c.get_stream() == 0
        side = torch.cuda.Stream()
        c.set_stream(side.cuda_stream)
        assert c.get_stream() == side.cuda_stream
        c.set_stream(0)
        # on the null stream the library's kernels are ordered against torch's default stream without any host wait
        N, L = 64, 50
        codes = sy.gen_codes(N, L, 5)
        raw = torch.from_numpy(codes).cuda()
        dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
        c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
        pitch = lib.Context.codes_pitch(L)
        got = dcodes[: N * pitch].view(N, pitch)[:, :L].cpu().numpy()
        assert np.array_equal(got, codes)
    finally:
        c.close()


def same_step(pa, pb):
    """Two runs of the same optimiser steps.  AdaBelief's early steps move a parameter by ~eta * sign(g), so ulp-level
    noise of a gradient entry near zero can flip a whole step of eta = 1e-3: all but a sliver of the entries agree to
    1e-6, none differs by more than two steps' worth."""
    a = np.concatenate([np.ravel(x) for x in pa])
    b = np.concatenate([np.ravel(x) for x in pb])
    d = np.abs(a - b)
    assert d.max() <= 4.6e-3
    assert (d > 1e-6).mean() < 0.01, (d > 1e-6).mean()


# ---- train.jl:33,41: the DataLoader batch as it is, Float32 one-hot (4L, 1, B) ----
def test_train_step_onehot_equals_train_step_on_codes(ctx, pkg):
    md, sy = pkg.model, pkg.synth
    hp = md.Hyperparam(filter_len=8, M=16, K=8, q=8, h=6)
    L, G = 60, 3
    codes = sy.gen_codes(G * hp.batch_size, L, 99, n_plant=2, k=8)
    onehot = sy.codes_to_onehot(codes)                        # (S, 4L) = the bytes of Julia's (4L, 1, S)
    a = md.ucdl(hp, L, ctx=ctx, seed=5, arena_bytes=1 << 30)
    b = md.ucdl(hp, L, ctx=ctx, seed=5, arena_bytes=1 << 30)
    try:
        for _ in range(2):
            la, l1a = a.model.train_step(codes, G)
            lb, l1b = b.model.train_step_onehot(onehot, G)
            # the same kernels on the same encoded reads; the engine's float atomics leave ulp-level run-to-run noise
            assert np.allclose(la, lb, rtol=2e-6) and abs(l1a - l1b) <= 1e-4 * abs(l1a)
        same_step(a.model.get_params(), b.model.get_params())
        bad = onehot.copy()
        bad[0, :4] = 0.5                                      # not one-hot
        with pytest.raises(pkg._lib.MotifsError) as e:
            b.model.train_step_onehot(bad, G)
        assert e.value.code == pkg._lib.ERR_NOT_ONEHOT
    finally:
        a.model.close()
        b.model.close()


# ---- RCCL behind the ABI: what one device can exercise (library found, id, communicator, both collectives) ----
def test_rccl_communicator_of_one_rank(torch_cuda, pkg):
    torch = torch_cuda
    lib = pkg._lib
    c = lib.Context(0)
    c.set_stream(0)
    try:
        uid = lib.Comm.unique_id()
        assert len(uid) == lib.COMM_ID_BYTES and any(uid)
        comm = lib.Comm(c, uid, 1, 0)
        f = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
        i = torch.arange(400, dtype=torch.int64, device="cuda") * 3 - 7
        comm.allreduce_sum_f32(f.data_ptr(), f.numel())
        comm.hist_allreduce(i.data_ptr(), 200, 2)
        c.synchronize()
        assert torch.equal(f.cpu(), torch.arange(1000, dtype=torch.float32) * 0.5)
        assert torch.equal(i.cpu(), torch.arange(400, dtype=torch.int64) * 3 - 7)
        comm.close()
    finally:
        c.close()


def test_dp_train_step_single_device_equals_train_step(ctx, pkg):
    """motifs_model_dp_train_step_dev with comm = NULL is loss_grad + AdaBelief on the mean: the same update as the host entry."""
    import torch

    md, sy, lib = pkg.model, pkg.synth, pkg._lib
    hp = md.Hyperparam(filter_len=8, M=16, K=8, q=8, h=6)
    L, G = 60, 4
    codes = sy.gen_codes(G * hp.batch_size, L, 123, n_plant=2, k=8)
    a = md.ucdl(hp, L, ctx=ctx, seed=9, arena_bytes=1 << 30)
    b = md.ucdl(hp, L, ctx=ctx, seed=9, arena_bytes=1 << 30)
    try:
        la, _ = a.model.train_step(codes, G, want_l1=False)
        raw = torch.from_numpy(codes).cuda()
        dcodes = torch.zeros(lib.Context.codes_bytes(codes.shape[0], L), dtype=torch.uint8, device="cuda")
        loss = torch.zeros(G, dtype=torch.float32, device="cuda")
        grad = torch.zeros(b.model.nP, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, codes.shape[0], L, dcodes.data_ptr())
        b.model.dp_train_step_dev(None, dcodes.data_ptr(), G, G, loss.data_ptr(), grad.data_ptr())
        ctx.synchronize()
        assert np.allclose(loss.cpu().numpy(), la, rtol=2e-6)
        same_step(a.model.get_params(), b.model.get_params())
    finally:
        a.model.close()
        b.model.close()


# ---- host-buffer entries: threaded one-hot -> codes on the host, records down through the pinned ring ----
def test_host_entry_many_records_equal_device_path(torch_cuda, ctx, pkg):
    """Enough records for more than one 16 MB chunk of the download ring (and several encoder threads): the host entries
    return exactly what the device-resident path and the CPU port give; all-zero columns and a bad column included."""
    sy, lib = pkg.synth, pkg._lib
    N, L, K = 7000, 200, 200
    codes = sy.gen_codes(N, L, 4242, n_plant=5, k=12)
    codes[17, 5] = 4
    codes[6999, 199] = 4
    pwms, lens = sy.gen_pwm_bank(K, 4243, len_lo=12, len_hi=12)
    bank = sy.pad_bank(pwms, lens)
    onehot = sy.codes_to_onehot(codes)
    both = ctx.pwm_scan_both(bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L)
    assert len(both[0][0]) * 12 > (16 << 20), "the test wants more than one chunk of records per strand"
    for rc in (False, True):
        h, s = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, 5000)
        f = both[int(rc)][0]
        got = np.stack([f["m"], f["n"], f["l"]], axis=1).astype(np.uint32)
        assert np.array_equal(got, h) and np.array_equal(both[int(rc)][1].view(np.uint16), s)
        oh, os_ = fast_oracle_hits(bank, lens, codes, rc, 5000)
        assert np.array_equal(h, oh) and np.array_equal(s, os_)
        f1, s1 = ctx.pwm_scan(bank, lens, onehot.astype(np.float16), lib.DATA_ONEHOT_F16, N, L, rc)
        assert np.array_equal(f1, f) and np.array_equal(s1.view(np.uint16), both[int(rc)][1].view(np.uint16))
    bad = onehot.copy()
    bad[6000, 4 * 33 + 1] = 0.5
    with pytest.raises(lib.MotifsError) as e:
        ctx.pwm_scan_both(bank, lens, bad, lib.DATA_ONEHOT_F32, N, L)
    assert e.value.code == lib.ERR_NOT_ONEHOT


# ---- the four-reads-per-wave candidate kernel: quads that straddle ordering batches, ragged ends ----
@pytest.mark.parametrize("N,L,K,lo,hi,batch", [
    (11, 37, 130, 9, 12, 5),      # quad 1 = reads 4..7 lies across batches 0|1; two chunks, the second with 1 live tile
    (6, 31, 33, 12, 12, 3),       # every quad across a batch boundary; Lout = 20: a tail tile of 4 windows
    (9, 20, 8, 16, 16, 2),        # T = 4; Lout = 5: one partial tile only
    (13, 45, 257, 17, 20, 7),     # T = 5 (two waves per SIMD), three chunks
    (5, 300, 64, 12, 12, 5000),   # long enough reads for the one-read-per-wave kernel (LDS images too large for quads)
])
@pytest.mark.parametrize("rc", [False, True])
def test_quad_candidate_kernel_edges(torch_cuda, ctx, pkg, N, L, K, lo, hi, batch, rc):
    sy = pkg.synth
    codes = sy.gen_codes(N, L, 31000 + N * 7 + K, n_plant=3, k=min(hi, L))
    codes[N - 1, L - 1] = 4
    codes[0, 0] = 4
    pwms, lens = sy.gen_pwm_bank(K, 32000 + K, len_lo=lo, len_hi=hi, alpha=0.4)
    bank = sy.pad_bank(pwms, lens)
    h, s = dev_scan_hits(torch_cuda, ctx, pkg, bank, lens, codes, rc, batch)
    oh, os_ = oracle_hits(pkg, bank, lens, codes, rc, batch)
    assert len(oh) > 0
    assert np.array_equal(h, oh) and np.array_equal(s, os_)
