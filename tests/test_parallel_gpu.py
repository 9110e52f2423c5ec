"""Real-compute tests of the sharded path (SURVEY.md §8e "Verify"), several ranks on ONE device:
  * the hit records of a scan sharded over the ranks, concatenated in rank order, are the single-device records bit
    for bit (forward and reverse strand), and the summed per-PWM histograms are the single-device histogram;
  * the reduced gradient of a data-parallel step is the sum of the shard gradients, every rank ends the step with the
    same parameters, and they are the parameters of a single-device step over all mini-batches;
  * a rank without a mini-batch joins the exchange with zeros.
RCCL cannot put two ranks on one GPU, so the sums ride torch.distributed/gloo here (parallel.HostReducer); the RCCL
communicator itself is exercised by tests/test_round2_gpu.py (one rank) and by bench.py --gpus N on a multi-GPU node.
At most 3 rank processes + the test runner touch the GPU at once."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(rank, ws, port):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    sys.path.insert(0, ROOT)
    from _pkg import load_pkg

    pkg = load_pkg()
    ctx = pkg._lib.Context(0)
    ctx.set_stream(0)
    reducer, _ = pkg.parallel.make_reducer(ctx, prefer_rccl=False)
    return pkg, ctx, reducer


def _scan_worker(rank, ws, port, align, ret):
    import torch
    import torch.distributed as dist

    pkg, ctx, reducer = _setup(rank, ws, port)
    sy, par, lib = pkg.synth, pkg.parallel, pkg._lib
    N, L, K, batch = 2300, 80, 72, 500                        # 5 ordering batches (the last short) over the ranks
    codes = sy.gen_codes(N, L, 2024, n_plant=3, k=10)
    pwms, lens = sy.gen_pwm_bank(K, 11, len_lo=8, len_hi=12, alpha=0.35)
    bank = sy.pad_bank(pwms, lens)
    fwd, rcs, counts = par.sharded_gpu_scan(ctx, bank, lens, codes, reducer=reducer, batch=batch, align=align)
    if rank == 0:                                             # the same scan on one device, in this process
        raw = torch.from_numpy(codes).cuda()
        dcodes = torch.zeros(lib.Context.codes_bytes(N, L), dtype=torch.uint8, device="cuda")
        ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, N, L, dcodes.data_ptr())
        need = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, None, None, 0, batch=batch)
        cap = max(need)
        hits = [torch.empty((cap, 3), dtype=torch.int32, device="cuda") for _ in range(2)]
        hsc = [torch.empty(cap, dtype=torch.int16, device="cuda") for _ in range(2)]
        c1 = torch.zeros((2, K), dtype=torch.int64, device="cuda")
        got = ctx.pwm_scan_hits_both_dev(bank, lens, dcodes.data_ptr(), N, L, [h.data_ptr() for h in hits], [s.data_ptr() for s in hsc],
                                         cap, batch=batch, counts_ptr=c1.data_ptr())
        ctx.synchronize()
        ok = True
        single = []
        for rc, (f, s) in enumerate((fwd, rcs)):
            one = hits[rc][: got[rc]].cpu().numpy().view(np.uint32)
            ok &= len(f) == got[rc] and got[rc] > 1000
            same = np.array_equal(np.stack([f["m"], f["n"], f["l"]], axis=1), one)
            same &= np.array_equal(s.view(np.uint16), hsc[rc][: got[rc]].cpu().numpy().view(np.uint16))
            # whole ordering batches per shard: the record lists are equal; shard edges inside a batch: they are not
            ok &= same if align in (None, batch) else not same
            rec = np.zeros(got[rc], dtype=lib.HIT_DTYPE)
            rec["m"], rec["n"], rec["l"] = one[:, 0], one[:, 1], one[:, 2]
            single.append((rec, hsc[rc][: got[rc]].cpu().numpy().view(np.float16)))
        # ... but what gpu_scan hands its callers - the per-(PWM, read) lists of modify_w_found! - is the same either way
        ok &= par.records_to_dicts(fwd, rcs, K) == par.records_to_dicts(single[0], single[1], K)
        ok &= np.array_equal(counts, c1.cpu().numpy())
        ret["scan_ok"] = bool(ok)
        ret["shards"] = [par.shard_range(N, r, ws, align=batch if align is None else align) for r in range(ws)]
    ctx.close()
    dist.destroy_process_group()


def _train_worker(rank, ws, port, n_groups, ret):
    import torch
    import torch.distributed as dist

    pkg, ctx, reducer = _setup(rank, ws, port)
    sy, par, lib, md = pkg.synth, pkg.parallel, pkg._lib, pkg.model
    hp = md.Hyperparam(filter_len=8, M=16, K=8, q=8, h=6)
    L, B = 60, hp.batch_size
    codes = sy.gen_codes(n_groups * B, L, 777, n_plant=2, k=8)

    def dev_codes(c):
        raw = torch.from_numpy(np.ascontiguousarray(c)).cuda()
        d = torch.zeros(lib.Context.codes_bytes(max(c.shape[0], 1), L), dtype=torch.uint8, device="cuda")
        if c.shape[0]:
            ctx.encode_dev(raw.data_ptr(), lib.DATA_CODES_U8, c.shape[0], L, d.data_ptr())
        return d

    lo, hi = par.shard_range(n_groups, rank, ws)
    cdl = md.ucdl(hp, L, ctx=ctx, seed=31, arena_bytes=1 << 30)
    nP = cdl.model.nP
    loss = torch.zeros(max(hi - lo, 1), dtype=torch.float32, device="cuda")
    grad = torch.zeros(nP, dtype=torch.float32, device="cuda")
    d_loc = dev_codes(codes[lo * B:hi * B])
    # the shard gradient on its own, before the step changes the parameters
    shard_grad = np.zeros(nP, dtype=np.float32)
    if hi > lo:
        cdl.model.loss_grad_dev(d_loc.data_ptr(), hi - lo, loss.data_ptr(), grad.data_ptr())
        ctx.synchronize()
        shard_grad = grad.cpu().numpy().copy()
    par.dp_train_step(cdl.model, d_loc.data_ptr(), hi - lo, loss, grad, n_groups, reducer=reducer)
    ctx.synchronize()
    D, F, _, v = cdl.model.get_params()
    params = np.concatenate([D, F, v])                        # the flat order of the gradient: [D | F | vecs]
    ret[f"shard_grad{rank}"] = shard_grad
    ret[f"params{rank}"] = params
    ret[f"reduced{rank}"] = grad.cpu().numpy().copy()
    ret[f"local{rank}"] = hi - lo
    if rank == 0:                                             # the whole step on one device
        one = md.ucdl(hp, L, ctx=ctx, seed=31, arena_bytes=1 << 30)
        l1 = torch.zeros(n_groups, dtype=torch.float32, device="cuda")
        g1 = torch.zeros(nP, dtype=torch.float32, device="cuda")
        d_all = dev_codes(codes)
        one.model.dp_train_step_dev(None, d_all.data_ptr(), n_groups, n_groups, l1.data_ptr(), g1.data_ptr())
        ctx.synchronize()
        ret["one_grad"] = g1.cpu().numpy().copy()
        D1, F1, _, v1 = one.model.get_params()
        ret["one_params"] = np.concatenate([D1, F1, v1])
        one.model.close()
    cdl.model.close()
    ctx.close()
    dist.destroy_process_group()


def _spawn(fn, ws, *args):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(fn, args=(ws, port) + args + (ret,), nprocs=ws, join=True)
    return dict(ret)


def test_sharded_scan_equals_single_device_bit_for_bit():
    ret = _spawn(_scan_worker, 2, None)
    assert ret["shards"] == [(0, 1500), (1500, 2300)]
    assert ret["scan_ok"]


def test_even_shards_give_the_single_device_dictionaries():
    """The fine shard mode (align = 1): BASELINE configs[2] then splits 12 500 x 8 instead of 3,3,3,3,2,2,2,2 ordering
    batches.  Global order = sequence-block-major; per (m, n): forward hits in ascending l, then reverse ones."""
    ret = _spawn(_scan_worker, 3, 1)
    assert ret["shards"] == [(0, 767), (767, 1534), (1534, 2300)]
    assert ret["scan_ok"]


@pytest.mark.parametrize("ws,n_groups", [(2, 5), (3, 2)])
def test_dp_train_step_equals_single_device_step(ws, n_groups):
    ret = _spawn(_train_worker, ws, n_groups)
    shard_sum = np.zeros_like(ret["shard_grad0"])
    for r in range(ws):
        shard_sum = shard_sum + ret[f"shard_grad{r}"]          # float32 adds in rank order, as gloo's ring of 2-3 does up to order
    if ws == 3:
        assert sorted(ret[f"local{r}"] for r in range(ws)) == [0, 1, 1]
    scale = float(np.abs(ret["one_grad"]).max())
    assert scale > 0
    for r in range(ws):
        # reduced gradient == sum of the shard gradients (float32 reassociation only)
        assert np.allclose(ret[f"reduced{r}"], shard_sum, rtol=0, atol=2e-6 * scale)
        # every rank holds the same parameters after the step, bit for bit
        assert np.array_equal(ret[f"params{r}"], ret["params0"])
    # ... and they are the single-device step's: same gradient up to summation order over the mini-batches
    assert np.allclose(ret["reduced0"], ret["one_grad"], rtol=0, atol=5e-6 * scale)
    # AdaBelief's first step is ~ eta * sign(g) wherever |g| >> sqrt(eps): compare the parameters where the gradient is
    # well away from zero (there a reassociation-sized change of g moves the update by < 1e-5), bound the rest by 2 * eta
    firm = np.abs(ret["one_grad"]) > 1e-3 * scale
    assert firm.sum() > 100
    assert np.allclose(ret["params0"][firm], ret["one_params"][firm], rtol=0, atol=2e-5)
    assert np.abs(ret["params0"] - ret["one_params"]).max() <= 2.3e-3
