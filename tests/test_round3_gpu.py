"""GPU tests added in round 3 (through the C ABI): device memory for a host without a GPU array package, the
data-parallel step of one host thread driving several devices (gradient -> grouped all-reduce -> update, in that
order), the host-buffer multi-device entries, and the even ("fine") shard mode of the scan."""
import numpy as np
import pytest

from test_round2_gpu import fast_oracle_hits, same_step

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch


def _rec(f):
    return np.stack([f["m"], f["n"], f["l"]], axis=1).astype(np.uint32)


# ---- motifs_dev_alloc / _free / _upload / _download / _memset: a scan whose every device buffer comes from the ABI ----
def test_scan_on_buffers_the_abi_allocated(pkg):
    """What julia/MotifsHIP.jl's DeviceReads / gpu_scan_resident do: no torch, no hipMalloc of the caller's own."""
    sy, lib = pkg.synth, pkg._lib
    N, L, K = 900, 70, 40
    codes = sy.gen_codes(N, L, 808, n_plant=3, k=10)
    pwms, lens = sy.gen_pwm_bank(K, 809, len_lo=8, len_hi=12, alpha=0.35)
    bank = sy.pad_bank(pwms, lens)
    c = lib.Context(0)                                   # stays on its private stream: nothing else touches the device
    try:
        raw = c.dev_alloc(codes.nbytes)
        dcodes = c.dev_alloc(lib.Context.codes_bytes(N, L))
        bad = c.dev_alloc(4)
        c.dev_upload(raw, codes)
        c.dev_memset(bad, 0, 4)
        c.encode_dev(raw, lib.DATA_CODES_U8, N, L, dcodes, bad)
        assert int(c.dev_download(bad, 1, np.int32)[0]) == 0
        need = c.pwm_scan_hits_both_dev(bank, lens, dcodes, N, L, None, None, 0, batch=400)
        cap = max(need)
        hits = [c.dev_alloc(cap * 12) for _ in range(2)]
        scs = [c.dev_alloc(cap * 2) for _ in range(2)]
        cnt = c.dev_alloc(2 * K * 8)
        got = c.pwm_scan_hits_both_dev(bank, lens, dcodes, N, L, hits, scs, cap, batch=400, counts_ptr=cnt)
        assert got == need and min(got) > 100
        counts = c.dev_download(cnt, (2, K), np.int64)
        for rc in (0, 1):
            h = c.dev_download(hits[rc], (got[rc], 3), np.uint32)
            s = c.dev_download(scs[rc], got[rc], np.uint16)
            oh, os_ = fast_oracle_hits(bank, lens, codes, bool(rc), 400)
            assert np.array_equal(h, oh) and np.array_equal(s, os_)
            assert np.array_equal(counts[rc], np.bincount(oh[:, 0] - 1, minlength=K))
        # a large block goes down through the pinned ring (> 4 MiB) and comes back unchanged
        big = np.arange(3 << 20, dtype=np.uint32)
        p = c.dev_alloc(big.nbytes)
        c.dev_upload(p, big)
        assert np.array_equal(c.dev_download(p, big.shape, np.uint32), big)
        for q in [raw, dcodes, bad, cnt, p] + hits + scs:
            c.dev_free(q)
    finally:
        c.close()


def test_use_private_stream_returns_to_a_stream_of_the_librarys_own(torch_cuda, pkg):
    lib = pkg._lib
    c = lib.Context(0)
    try:
        first = c.get_stream()
        assert first != 0
        c.set_stream(0)
        assert c.get_stream() == 0
        c.use_private_stream()
        assert c.get_stream() != 0
        c.use_private_stream()                           # idempotent
    finally:
        c.close()


# ---- the ordering bug of round 2: AdaBelief enqueued before a grouped all-reduce ----
def _model_pair(pkg, c, seed=9):
    md = pkg.model
    hp = md.Hyperparam(filter_len=8, M=16, K=8, q=8, h=6)
    return hp, md.ucdl(hp, 60, ctx=c, seed=seed, arena_bytes=1 << 30), md.ucdl(hp, 60, ctx=c, seed=seed, arena_bytes=1 << 30)


def test_grouped_step_reduces_before_it_updates(torch_cuda, pkg):
    """One rank of ncclCommInitAll on one device.  The sum goes OUT OF PLACE into a buffer poisoned with NaN which the
    optimiser then reads: had AdaBelief been enqueued before the (deferred) all-reduce, as motifs_model_dp_train_step_dev
    did inside ncclGroupStart/End in round 2, the parameters would be NaN.  They must equal the ungrouped step's."""
    torch = torch_cuda
    sy, lib = pkg.synth, pkg._lib
    c = lib.Context(0)
    c.set_stream(0)
    try:
        hp, a, b = _model_pair(pkg, c)
        G, L = 3, 60
        codes = sy.gen_codes(G * hp.batch_size, L, 123, n_plant=2, k=8)
        raw = torch.from_numpy(codes).cuda()
        dcodes = torch.zeros(lib.Context.codes_bytes(codes.shape[0], L), dtype=torch.uint8, device="cuda")
        c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, codes.shape[0], L, dcodes.data_ptr())
        nP = a.model.nP
        loss_a, loss_b = (torch.zeros(G, dtype=torch.float32, device="cuda") for _ in range(2))
        grad_a, grad_b = (torch.zeros(nP, dtype=torch.float32, device="cuda") for _ in range(2))
        # the reference: one device, no communicator
        a.model.dp_train_step_dev(None, dcodes.data_ptr(), G, G, loss_a.data_ptr(), grad_a.data_ptr())
        comms = lib.Comm.create_all([c])
        assert comms[0].nranks == 1
        # (1) the one-call form, sum out of place into poison
        red = torch.full((nP,), float("nan"), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        lib.dp_train_step_all([b.model], comms, [dcodes.data_ptr()], [G], G, [loss_b.data_ptr()], [grad_b.data_ptr()], [red.data_ptr()])
        c.synchronize()
        pb = b.model.get_params()
        assert all(np.isfinite(x).all() for x in pb)
        assert torch.isfinite(red).all() and torch.allclose(red, grad_b, rtol=0, atol=0)   # one rank: the sum is the gradient
        assert np.allclose(loss_b.cpu().numpy(), loss_a.cpu().numpy(), rtol=2e-6)
        same_step(a.model.get_params(), pb)
        # (2) the three phases by hand, only the collective inside the group
        red.fill_(float("nan"))
        torch.cuda.synchronize()
        a.model.dp_train_step_dev(None, dcodes.data_ptr(), G, G, loss_a.data_ptr(), grad_a.data_ptr())
        b.model.dp_grad_dev(dcodes.data_ptr(), G, loss_b.data_ptr(), grad_b.data_ptr())
        lib.Comm.group_start()
        comms[0].allreduce_sum_f32_to(grad_b.data_ptr(), red.data_ptr(), nP)
        lib.Comm.group_end()
        b.model.dp_update_dev(red.data_ptr(), G)
        c.synchronize()
        pb = b.model.get_params()
        assert all(np.isfinite(x).all() for x in pb)
        same_step(a.model.get_params(), pb)
        # (3) the per-rank entry refuses to run inside an open group, and says why
        lib.Comm.group_start()
        try:
            with pytest.raises(lib.MotifsError) as e:
                b.model.dp_train_step_dev(comms[0], dcodes.data_ptr(), G, G, loss_b.data_ptr(), grad_b.data_ptr())
            assert e.value.code == lib.ERR_INVALID and "group" in str(e.value)
            with pytest.raises(lib.MotifsError):
                lib.dp_train_step_all([b.model], comms, [dcodes.data_ptr()], [G], G, [loss_b.data_ptr()], [grad_b.data_ptr()])
        finally:
            lib.Comm.group_end()
        with pytest.raises(lib.MotifsError):             # an _end without a _start is an error, not a crash
            lib.Comm.group_end()
        # outside a group the per-rank entry works with the communicator
        b.model.dp_train_step_dev(comms[0], dcodes.data_ptr(), G, G, loss_b.data_ptr(), grad_b.data_ptr())
        a.model.dp_train_step_dev(None, dcodes.data_ptr(), G, G, loss_a.data_ptr(), grad_a.data_ptr())
        c.synchronize()
        same_step(a.model.get_params(), b.model.get_params())
        c.synchronize()
        for cm in comms:
            cm.close()
        a.model.close()
        b.model.close()
    finally:
        c.close()


@pytest.mark.parametrize("null_stream", [False, True])
def test_step_graph_on_the_null_stream(torch_cuda, pkg, null_stream):
    """Found in round 3 by the test above: from its third call on a step of <= 8 mini-batches is replayed from a captured
    hipGraph, and with the context on HIP's legacy null stream (motifs_ctx_set_stream(ctx, NULL): torch's default stream)
    replays returned gradients of ~1e28.  The cause was the hipMemsetAsync / hipMemcpyAsync nodes of the captured step; it
    is all kernels now (dev_zero / dev_copy) and is replayed on any stream.  Six steps of two default models against an
    always-eager one, nothing synchronised in between, on either stream."""
    import os

    torch = torch_cuda
    sy, lib, md = pkg.synth, pkg._lib, pkg.model
    c = lib.Context(0)
    if null_stream:
        c.set_stream(0)
    try:
        hp = md.Hyperparam(filter_len=8, M=16, K=8, q=8, h=6)
        G, L = 3, 60
        os.environ["MOTIFS_NO_GRAPH"] = "1"
        try:
            eager = md.ucdl(hp, L, ctx=c, seed=9, arena_bytes=1 << 30)
        finally:
            del os.environ["MOTIFS_NO_GRAPH"]
        models = [eager, md.ucdl(hp, L, ctx=c, seed=9, arena_bytes=1 << 30), md.ucdl(hp, L, ctx=c, seed=9, arena_bytes=1 << 30)]
        codes = sy.gen_codes(G * hp.batch_size, L, 123, n_plant=2, k=8)
        raw = torch.from_numpy(codes).cuda()
        dcodes = torch.zeros(lib.Context.codes_bytes(codes.shape[0], L), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        c.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, codes.shape[0], L, dcodes.data_ptr())
        bufs = [(torch.zeros(G, dtype=torch.float32, device="cuda"), torch.zeros(eager.model.nP, dtype=torch.float32, device="cuda")) for _ in models]
        torch.cuda.synchronize()
        for it in range(6):
            for m, (lo, gr) in zip(models, bufs):
                m.model.dp_train_step_dev(None, dcodes.data_ptr(), G, G, lo.data_ptr(), gr.data_ptr())
            c.synchronize()
            g = [float(gr.abs().sum().item()) for _, gr in bufs]
            assert all(np.isfinite(x) and abs(x - g[0]) <= 1e-4 * g[0] for x in g), (it, g)
        for m in models[1:]:
            same_step(eager.model.get_params(), m.model.get_params())
        for m in models:
            m.model.close()
    finally:
        c.close()


def test_dp_train_step_host_equals_train_step(ctx, pkg):
    """motifs_model_dp_train_step_host on one device (no communicator) is train_step; on the reference's Float32 batches too."""
    md, sy, lib = pkg.model, pkg.synth, pkg._lib
    hp = md.Hyperparam(filter_len=8, M=16, K=8, q=8, h=6)
    L, G = 60, 5
    codes = sy.gen_codes(G * hp.batch_size, L, 321, n_plant=2, k=8)
    a = md.ucdl(hp, L, ctx=ctx, seed=4, arena_bytes=1 << 30)
    b = md.ucdl(hp, L, ctx=ctx, seed=4, arena_bytes=1 << 30)
    try:
        for step in range(2):
            la, l1a = a.model.train_step(codes, G)
            if step == 0:
                lb, l1b = lib.dp_train_step_host([b.model], None, codes, lib.DATA_CODES_U8, G)
            else:
                lb, l1b = lib.dp_train_step_host([b.model], None, sy.codes_to_onehot(codes), lib.DATA_ONEHOT_F32, G)
            assert np.allclose(la, lb, rtol=2e-5 if step else 2e-6)
            assert abs(l1a - l1b) <= 1e-3 * abs(l1a)
        same_step(a.model.get_params(), b.model.get_params())
        bad = sy.codes_to_onehot(codes)
        bad[3, 4 * 7 + 2] = 0.25
        with pytest.raises(lib.MotifsError) as e:
            lib.dp_train_step_host([b.model], None, bad, lib.DATA_ONEHOT_F32, G)
        assert e.value.code == lib.ERR_NOT_ONEHOT
    finally:
        a.model.close()
        b.model.close()


# ---- motifs_pwm_scan_both_sharded: host matrix, several contexts, one host thread each ----
@pytest.mark.parametrize("n_ctx", [1, 3])
def test_sharded_host_scan(torch_cuda, pkg, n_ctx):
    """Three contexts on the one device stand in for three devices (the histogram is then summed on the host; with a
    communicator of one rank the RCCL branch runs as well).  align = 500 (the ordering batch): the concatenation is the
    single-device record list bit for bit; align = 1: even shards, sequence-block-major global order, equal dictionaries."""
    sy, lib, par = pkg.synth, pkg._lib, pkg.parallel
    N, L, K = 2300, 80, 72
    codes = sy.gen_codes(N, L, 2024, n_plant=3, k=10)
    codes[5, 3] = 4
    pwms, lens = sy.gen_pwm_bank(K, 11, len_lo=8, len_hi=12, alpha=0.35)
    bank = sy.pad_bank(pwms, lens)
    onehot = sy.codes_to_onehot(codes)
    ctxs = [lib.Context(0) for _ in range(n_ctx)]
    try:
        one = ctxs[0].pwm_scan_both(bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L)      # ordering batches of 5000: one batch
        want_counts = np.stack([np.bincount(one[s][0]["m"].astype(np.int64) - 1, minlength=K) for s in (0, 1)])
        want_dicts = par.records_to_dicts(one[0], one[1], K)
        # whole ordering batches per shard: N < 5000, so every read lands on the first context and the lists are equal
        fwd, rcs, counts, shard = lib.pwm_scan_both_sharded(ctxs, None, bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L)
        assert np.array_equal(fwd[0], one[0][0]) and np.array_equal(rcs[0], one[1][0])
        assert np.array_equal(fwd[1].view(np.uint16), one[0][1].view(np.uint16))
        assert np.array_equal(counts, want_counts)
        assert shard[0].tolist() == [len(one[0][0]), len(one[1][0])] and shard[1:].sum() == 0
        # even shards
        for kind, data in ((lib.DATA_ONEHOT_F32, onehot), (lib.DATA_CODES_U8, codes)):
            fwd, rcs, counts, shard = lib.pwm_scan_both_sharded(ctxs, None, bank, lens, data, kind, N, L, shard_align=1)
            assert len(fwd[0]) == len(one[0][0]) and len(rcs[0]) == len(one[1][0])
            assert np.array_equal(counts, want_counts)
            assert par.records_to_dicts(fwd, rcs, K) == want_dicts
            if n_ctx > 1:
                assert (shard > 0).all() and not np.array_equal(fwd[0], one[0][0])
                # device d's records are exactly the single-strand scan of its block with n0 = its first read
                edges = [par.shard_range(N, r, n_ctx, align=1) for r in range(n_ctx)]
                at = 0
                for d, (a, b) in enumerate(edges):
                    oh, os_ = fast_oracle_hits(bank, lens, codes[a:b], False, 5000)
                    oh = oh.copy()
                    oh[:, 1] += a
                    assert np.array_equal(_rec(fwd[0][at:at + len(oh)]), oh)
                    assert np.array_equal(fwd[1][at:at + len(oh)].view(np.uint16), os_)
                    at += len(oh)
                assert at == len(fwd[0])
        # a buffer that is too small: the status, and the counts needed
        with pytest.raises(lib.MotifsError) as e:
            lib.pwm_scan_both_sharded(ctxs, None, bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L, shard_align=1, cap=10)
        assert e.value.code == lib.ERR_BUFFER_TOO_SMALL
        if n_ctx == 1:                                   # the RCCL branch of the histogram (a communicator of one rank)
            comms = lib.Comm.create_all(ctxs)
            fwd, rcs, counts, _ = lib.pwm_scan_both_sharded(ctxs, comms, bank, lens, onehot, lib.DATA_ONEHOT_F32, N, L, shard_align=1)
            assert np.array_equal(counts, want_counts) and np.array_equal(fwd[0], one[0][0])
            for cm in comms:
                cm.close()
    finally:
        for c in ctxs:
            c.close()


def test_rccl_reducer_orders_an_operand_from_another_stream(torch_cuda, pkg):
    """a context on its private stream, the operand filled by torch on ITS stream, a rank with no reads: nothing
    but the reducer itself orders the fill against the all-reduce.  One rank (RCCL cannot put two on one GPU): the value
    check is trivial, the path is the one an empty rank takes."""
    torch = torch_cuda
    lib, par, sy = pkg._lib, pkg.parallel, pkg.synth
    c = lib.Context(0)                                   # private non-blocking stream
    try:
        comm = lib.Comm(c, lib.Comm.unique_id(), 1, 0)
        red = par.RcclReducer(comm)
        for _ in range(20):
            big = torch.empty(1 << 24, dtype=torch.float32, device="cuda").normal_()   # keeps torch's stream busy ...
            t = torch.zeros((2, 300), dtype=torch.int64, device="cuda") + big[:600].reshape(2, 300).long() * 0 + 7   # ... in front of this fill
            red.sum_i64_(t)
            c.synchronize()
            assert int(t.min().item()) == 7 and int(t.max().item()) == 7
        # sharded_gpu_scan on a rank whose shard is empty (N = 0 reads) returns empty lists and zero counts
        pwms, lens = sy.gen_pwm_bank(8, 1, len_lo=8, len_hi=8)
        fwd, rcs, counts = par.sharded_gpu_scan(c, sy.pad_bank(pwms, lens), lens, np.zeros((0, 50), dtype=np.uint8), reducer=red)
        assert len(fwd[0]) == 0 and len(rcs[0]) == 0 and counts.sum() == 0
        comm.close()
    finally:
        c.close()


# ---- BASELINE configs[3] / configs[4] at the size of one rank's shard: properties that do not need the oracle at full size ----
def _full_size_shard(torch, pkg, ctx, n, L, K, len_lo, len_hi, ws_limit, sample):
    lib, sy = pkg._lib, pkg.synth
    pwms, lens = sy.gen_pwm_bank(K, 4711 + K, len_lo=len_lo, len_hi=len_hi, alpha=0.3)
    bank = sy.pad_bank(pwms, lens)
    codes = sy.gen_codes(n, L, 4712 + K, n_plant=5, k=len_hi)
    codes[7, 11] = 4
    raw = torch.from_numpy(codes).cuda()
    dcodes = torch.zeros(lib.Context.codes_bytes(n, L), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, n, L, dcodes.data_ptr())
    del raw
    ctx.set_workspace_limit(ws_limit)
    try:
        need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), n, L, None, None, 0)
        cap = max(need)
        hits = [torch.empty((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
        hsc = [torch.empty(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
        cnt = torch.zeros((2, K), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        got = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), n, L, [h.data_ptr() for h in hits], [s.data_ptr() for s in hsc], cap,
                                         counts_ptr=cnt.data_ptr())
        ctx.synchronize()
    finally:
        ctx.set_workspace_limit(0)
    assert got == need and min(got) > 100_000
    B = lib.SCAN_BATCH
    for rc in (0, 1):
        f = hits[rc][: got[rc]].to(torch.int64)
        m, nn, l = f[:, 0], f[:, 1], f[:, 2]
        assert int(m.min()) >= 1 and int(m.max()) <= K and int(nn.min()) >= 1 and int(nn.max()) <= n and int(l.min()) >= 1
        lens_t = torch.from_numpy(lens).cuda()
        assert bool((l <= L - lens_t[m - 1] + 1).all())                               # every window lies inside its read
        assert bool((hsc[rc][: got[rc]] > 0).all())                                   # binary16 bits of a positive score
        key = (((nn - 1) // B * (L + 1) + l) * B + (nn - 1) % B) * (K + 1) + m        # (batch, l, n, m): findall's order per batch
        assert bool((key[1:] > key[:-1]).all()), "records are not in the reference's order"
        assert torch.equal(torch.bincount(m - 1, minlength=K), cnt[rc])               # the histogram is the records'
        # sampled reads against the CPU port: every record of the read, scores included
        hs = hsc[rc][: got[rc]]
        for r0 in sample:
            sel = (nn > r0) & (nn <= r0 + 3)
            mine = torch.stack([m[sel], nn[sel], l[sel]], dim=1).cpu().numpy().astype(np.uint32)
            mys = hs[sel].cpu().numpy().view(np.uint16)
            oh, os_ = fast_oracle_hits(bank, lens, codes[r0:r0 + 3], bool(rc), B)
            oh = oh.copy()
            oh[:, 1] += r0
            # the oracle saw these three reads as a batch of their own: same records, (l, n, m) order within it
            order = np.lexsort((mine[:, 0], mine[:, 1], mine[:, 2]))
            assert np.array_equal(mine[order], oh) and np.array_equal(mys[order], os_), (rc, r0)
        del f, m, nn, l, key
    return got


def test_cfg3_shard_full_size_properties(torch_cuda, ctx, pkg):
    """62 500 reads x 500 bp vs 512 PWMs of 20 positions: one rank's shard of BASELINE configs[3] (13 ordering batches, one launch)."""
    _full_size_shard(torch_cuda, pkg, ctx, 62_500, 500, 512, 20, 20, 0, sample=(0, 4998, 31_249, 62_497))


def test_cfg4_shard_full_size_properties(torch_cuda, ctx, pkg):
    """25 000 reads x 1000 bp vs 2048 PWMs of 8-20 positions (a fifth of one rank's shard of BASELINE configs[4]) under a 4 GiB
    workspace bound: each strand crosses three super-batch launches (2 + 2 + 1 ordering batches), the running total chained on
    the device; samples sit on both sides of the launch edges."""
    _full_size_shard(torch_cuda, pkg, ctx, 25_000, 1000, 2048, 8, 20, 4 << 30, sample=(0, 9_998, 10_000, 19_999, 24_997))
