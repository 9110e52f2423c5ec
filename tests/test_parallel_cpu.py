"""world_size-2 gloo tests of the data-parallel plumbing (motifs.jl_amd/parallel.py): shard layout and
the two all-reduces.  The local compute is a deterministic stand-in (the HIP path needs a GPU; its
single-process equivalence to per-group sums is covered by tests/test_model_gpu.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range_partitions(pkg):
    sr = pkg.parallel.shard_range
    for n, ws, al in [(100000, 8, 6), (100000, 8, 5000), (31, 2, 6), (5, 4, 6), (0, 3, 1), (12, 2, 6)]:
        edges = [sr(n, r, ws, al) for r in range(ws)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        for (a, b), (c, d) in zip(edges, edges[1:]):
            assert b == c and a <= b
        for a, b in edges:
            assert (a % al == 0 or a == n) and (b % al == 0 or b == n)   # mini-batches never straddle ranks
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 2 * al


def fake_grad(lo, hi, n):
    """Deterministic stand-in for the summed gradient of groups [lo, hi)."""
    g = np.zeros(n, dtype=np.float32)
    for k in range(lo, hi):
        g += np.sin(np.arange(n, dtype=np.float32) * 0.01 + k)
    return g


def _worker(rank, ws, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from _pkg import load_pkg

    par = load_pkg().parallel
    n_groups, n = 10, 257
    lo, hi = par.shard_range(n_groups, rank, ws)
    grad = torch.from_numpy(fake_grad(lo, hi, n))
    par.allreduce_sum_(grad)
    counts = torch.tensor([lo, hi, 7 * rank], dtype=torch.int64)
    par.allreduce_sum_(counts)
    hits = np.arange(lo, hi, dtype=np.uint32)
    allh, alls = par.gather_hits(hits, hits.astype(np.float16))
    if rank == 0:
        ret["grad"] = grad.numpy().copy()
        ret["counts"] = counts.numpy().copy()
        ret["hits"] = allh
    dist.destroy_process_group()


def test_allreduce_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    want = fake_grad(0, 10, 257)
    assert np.allclose(ret["grad"], want, rtol=1e-6, atol=1e-5)
    assert ret["counts"].tolist() == [5, 15, 7]
    assert ret["hits"].tolist() == list(range(10))            # rank order == sequence-block order
