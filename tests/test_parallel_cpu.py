"""CPU tests of the data-parallel host side (motifs.jl_amd/parallel.py): shard layout, the reducer classes over a
world_size-2 and world_size-3 gloo group, and the control flow of a data-parallel step in which one rank has no
mini-batch.  The local compute here is a deterministic stand-in for the HIP kernels (they need a GPU); the
real-compute equivalents — sharded hit records == single-device records bit for bit, reduced gradient == sum of the
shard gradients — are tests/test_parallel_gpu.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_partitions(pkg):
    sr = pkg.parallel.shard_range
    for n, ws, al in [(100000, 8, 6), (100000, 8, 5000), (31, 2, 6), (5, 4, 6), (0, 3, 1), (12, 2, 6), (100000, 3, 5000),
                      (9999, 8, 5000), (12, 8, 6)]:
        edges = [sr(n, r, ws, al) for r in range(ws)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        for (a, b), (c, d) in zip(edges, edges[1:]):
            assert b == c and a <= b
        for a, b in edges:
            assert (a % al == 0 or a == n) and (b % al == 0 or b == n)   # mini-batches never straddle ranks
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 2 * al


def test_shard_range_can_leave_a_rank_empty(pkg):
    """More ranks than units: some blocks are empty — the callers must still join every collective (dp_train_step
    with n_groups_local = 0, sharded_gpu_scan with no reads)."""
    sr = pkg.parallel.shard_range
    edges = [sr(12, r, 8, 6) for r in range(8)]            # 2 mini-batches over 8 ranks
    assert sum(1 for a, b in edges if b > a) == 2 and sum(b - a for a, b in edges) == 12
    edges = [sr(9999, r, 8, 5000) for r in range(8)]       # 2 ordering batches over 8 ranks
    assert [b - a for a, b in edges] == [5000, 4999, 0, 0, 0, 0, 0, 0]


def fake_grad(lo, hi, n):
    """Deterministic stand-in for the summed gradient of groups [lo, hi)."""
    g = np.zeros(n, dtype=np.float32)
    for k in range(lo, hi):
        g += np.sin(np.arange(n, dtype=np.float32) * 0.01 + k)
    return g


class FakeModel:
    """The calls dp_train_step makes on a Model, on host tensors: gradient of the local groups, then a plain SGD step
    standing in for AdaBelief (the point is the control flow around the exchange, not the optimiser)."""

    def __init__(self, n, lo):
        self.n, self.lo = n, lo
        self.params = np.zeros(n, dtype=np.float32)
        self.ctx = self
        self.calls = []

    def synchronize(self):
        pass

    def loss_grad_dev(self, codes_ptr, n_groups, loss_ptr, grad_ptr):
        self.calls.append(("grad", n_groups))
        self._grad[:] = torch.from_numpy(fake_grad(self.lo, self.lo + n_groups, self.n))

    def adabelief_dev(self, grad_ptr, gscale):
        self.calls.append(("step", gscale))
        self.params -= 0.1 * gscale * self._grad.numpy()


def _worker(rank, ws, port, n_groups, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from _pkg import load_pkg

    par = load_pkg().parallel
    n = 257
    reducer, note = par.make_reducer(None, prefer_rccl=False)
    assert isinstance(reducer, par.HostReducer)
    lo, hi = par.shard_range(n_groups, rank, ws)
    # the two sums of the path through the reducer
    grad = torch.from_numpy(fake_grad(lo, hi, n))
    reducer.sum_f32_(grad)
    counts = torch.tensor([lo, hi, 7 * rank], dtype=torch.int64)
    reducer.sum_i64_(counts)
    t = par.host_all_reduce(torch.tensor([float(rank)], dtype=torch.float64), dist.ReduceOp.MAX)
    # one data-parallel step, the group total unknown to the caller (summed over the ranks); a rank may have no group
    model = FakeModel(n, lo)
    model._grad = torch.zeros(n, dtype=torch.float32)
    loss = torch.zeros(max(hi - lo, 1), dtype=torch.float32)
    par.dp_train_step(model, 0, hi - lo, loss, model._grad, None, reducer=reducer)
    hits = np.arange(lo, hi, dtype=np.uint32)
    allh, alls = par.gather_hits(hits, hits.astype(np.float16))
    ret[rank] = {"grad": grad.numpy().copy(), "counts": counts.numpy().copy(), "hits": allh, "max": float(t.item()),
                 "params": model.params.copy(), "calls": model.calls, "local": hi - lo}
    dist.destroy_process_group()


def _run(ws, n_groups):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(ws, port, n_groups, ret), nprocs=ws, join=True)
    return dict(ret)


def test_reducers_match_single_process():
    ret = _run(2, 10)
    want = fake_grad(0, 10, 257)
    for r in (0, 1):
        assert np.allclose(ret[r]["grad"], want, rtol=1e-6, atol=1e-5)
        assert ret[r]["counts"].tolist() == [5, 15, 7]
        assert ret[r]["hits"].tolist() == list(range(10))            # rank order == sequence-block order
        assert ret[r]["max"] == 1.0
        # the step used the mean over ALL groups, on every rank alike
        assert np.allclose(ret[r]["params"], -0.1 * want / 10, rtol=1e-6, atol=1e-6)
    assert np.array_equal(ret[0]["params"], ret[1]["params"])        # replicas stay bit-identical


def test_rank_without_a_mini_batch_still_joins_the_exchange():
    """2 mini-batches over 3 ranks (such a rank used to raise before the all-reduce and hang the others)."""
    ret = _run(3, 2)
    want = fake_grad(0, 2, 257)
    assert sorted(ret[r]["local"] for r in range(3)) == [0, 1, 1]
    for r in range(3):
        assert np.allclose(ret[r]["params"], -0.1 * want / 2, rtol=1e-6, atol=1e-6)
        kinds = [c[0] for c in ret[r]["calls"]]
        assert kinds == (["grad", "step"] if ret[r]["local"] else ["step"])
        assert ret[r]["calls"][-1][1] == 0.5
    assert np.array_equal(ret[0]["params"], ret[1]["params"]) and np.array_equal(ret[0]["params"], ret[2]["params"])


def test_fine_shards_keep_the_per_pwm_per_read_order(pkg):
    """shard edges inside an ordering batch (align = 1: even shards) change the GLOBAL record order to
    sequence-block-major, but for every (m, n) the records still come in ascending l per strand, so the dictionaries
    modify_w_found! builds (_h3_1_alignment.jl:38-52) are the single-device ones.  The per-shard scan here is the CPU
    oracle (the GPU equivalent is tests/test_parallel_gpu.py); what is under test is the shard layout and the
    concatenation rule."""
    from oracle import scan_oracle as so

    sy, par = pkg.synth, pkg.parallel
    N, L, K, batch = 230, 40, 12, 50
    codes = sy.gen_codes(N, L, 5, n_plant=2, k=8)
    pwms, lens = sy.gen_pwm_bank(K, 3, len_lo=6, len_hi=9, alpha=0.4)
    bank = sy.pad_bank(pwms, lens)
    onehot = sy.codes_to_onehot(codes)

    def scan(lo, hi, rc):
        f, s = so.get_pos_scores_arr(bank, lens, onehot[lo:hi], rc=rc, batch_size=batch)
        f = f.copy()
        f["n"] += lo                     # n0 of the shard (_h3_1_alignment.jl:83 adds the batch offset the same way)
        return f, s

    one = [scan(0, N, rc) for rc in (False, True)]
    want = par.records_to_dicts(one[0], one[1], K)
    for ws, align in [(8, 1), (3, 6), (8, batch), (5, 7)]:
        edges = [par.shard_range(N, r, ws, align=align) for r in range(ws)]
        if align == 1:
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1           # even: configs[2] splits 12500 x 8, not 3,3,3,3,2,2,2,2 batches
        parts = [[scan(a, b, rc) for a, b in edges if b > a] for rc in (False, True)]
        cat = [(np.concatenate([p[0] for p in parts[rc]]), np.concatenate([p[1] for p in parts[rc]])) for rc in (0, 1)]
        assert len(cat[0][0]) == len(one[0][0]) and len(cat[1][0]) == len(one[1][0])
        got = par.records_to_dicts(cat[0], cat[1], K)
        assert got == want                               # positions, scores and use_comp, list by list
        if align == batch:                               # whole ordering batches: the record lists themselves are equal
            for rc in (0, 1):
                assert np.array_equal(cat[rc][0], one[rc][0])
                assert np.array_equal(cat[rc][1].view(np.uint16), one[rc][1].view(np.uint16))
        elif ws > 1 and align == 1:
            assert not np.array_equal(cat[0][0], one[0][0])   # ... and otherwise they are NOT: only the per-(m, n) order holds
