"""Data-parallel host side: one process per GPU, reads sharded in contiguous blocks, parameters and
the PWM bank replicated.  The only exchanges are
  * training: one sum of the flat gradient [dD | dF | dvecs] per optimiser step, after which every rank applies the
    identical AdaBelief update (`motifs_model_dp_train_step_dev`);
  * scanning: one sum of the K-entry hit histogram per strand (`motifs_hist_allreduce`); hit records are
    concatenated in rank order, which is the single-device record order because shard edges fall on the
    5000-read ordering batches of `get_pos_scores_arr` (_h3_1_alignment.jl:71).
The reference has no counterpart (single GPU, `src/MOTIFs.jl:4-8` imports no communication package).

The collectives themselves live behind the C ABI (RCCL over xGMI, `csrc/comm_rccl.hip`) and run on the context's
stream, so they are ordered against the kernels that produce and consume their operands.  torch.distributed is
used for the rendezvous only (carrying the 128-byte RCCL id to the other ranks, barriers around timed regions);
`HostReducer` is the stand-in for boxes where RCCL cannot form the communicator — several ranks on ONE device in
the rehearsal tests, or CPU-only gloo runs."""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_range(n_items, rank, world_size, align=1):
    """Contiguous block [lo, hi) of `n_items` for `rank`; block edges fall on multiples of `align`
    (align = hp.batch_size keeps every 6-sequence mini-batch on one rank; align = 5000 keeps the
    scan's ordering batches whole).  Blocks differ by at most one aligned unit; the tail goes last.
    A rank may get an empty block (more ranks than units): callers must still join every collective."""
    units = (n_items + align - 1) // align
    base, extra = divmod(units, world_size)
    lo_u = rank * base + min(rank, extra)
    hi_u = lo_u + base + (1 if rank < extra else 0)
    return min(lo_u * align, n_items), min(hi_u * align, n_items)


class RcclReducer:
    """Sums over ranks through the library's own RCCL communicator (device pointers, the context's stream)."""

    kind = "rccl-c-abi"

    def __init__(self, comm):
        self.comm = comm

    def _order(self, t):
        """The collective runs on the context's stream.  An operand torch produced on ANOTHER stream (a fill on torch's
        current stream while the context keeps its private one) is not ordered against it by anything: wait for torch's
        stream first.  When the two are the same stream (bench.py, ctx.set_stream) the stream itself orders them."""
        if t.is_cuda:
            cur = torch.cuda.current_stream(t.device)
            if int(cur.cuda_stream) != int(self.comm.ctx.get_stream()):
                cur.synchronize()

    def sum_f32_(self, t):
        self._order(t)
        self.comm.allreduce_sum_f32(t.data_ptr(), t.numel())
        return t

    def sum_i64_(self, t):
        self._order(t)
        self.comm.allreduce_sum_i64(t.data_ptr(), t.numel())
        return t

    def hist_sum_(self, counts):
        """The (2, K) per-PWM hit counts of the both-strands scan just made on this context (motifs_hist_allreduce, on the context's stream)."""
        self._order(counts)
        self.comm.hist_allreduce(counts.data_ptr(), counts.shape[-1], counts.shape[0] if counts.dim() == 2 else 1)
        return counts


def host_all_reduce(t, op=None):
    """torch.distributed all-reduce of a small tensor wherever it lives: NCCL/RCCL groups only take device tensors, gloo
    groups are fed host copies.  Blocking; the result is in `t` on return."""
    op = op or dist.ReduceOp.SUM
    if world()[1] == 1:
        return t
    if dist.get_backend() == "nccl":
        d = t if t.is_cuda else t.cuda()
        dist.all_reduce(d, op=op)
        torch.cuda.synchronize(d.device)
        if d is not t:
            t.copy_(d.cpu())
    else:
        h = t.cpu() if t.is_cuda else t
        dist.all_reduce(h, op=op)
        if h is not t:
            t.copy_(h)
            torch.cuda.synchronize(t.device)
    return t


class HostReducer:
    """Sums over ranks with torch.distributed, blocking.  `ctx` (optional) is synchronised before the operand is read
    and the result is in place before the call returns, so no stream order is left to chance."""

    kind = "torch.distributed"

    def __init__(self, ctx=None):
        self.ctx = ctx

    def _sum(self, t):
        if world()[1] == 1:
            return t
        if self.ctx is not None:
            self.ctx.synchronize()
        if t.is_cuda:
            torch.cuda.synchronize(t.device)
        return host_all_reduce(t)

    sum_f32_ = _sum
    sum_i64_ = _sum
    hist_sum_ = _sum


def make_reducer(ctx, prefer_rccl=True):
    """The reducer of this process group: the C-ABI RCCL communicator when one rank owns one GPU, else the host
    stand-in.  Collective over the torch.distributed group (every rank must call it).  Returns (reducer, note)."""
    rank, ws = world()
    if ws == 1:
        return HostReducer(ctx), "single rank"
    if prefer_rccl:
        err = None
        try:
            box = [_lib.Comm.unique_id() if rank == 0 else None]
        except _lib.MotifsError as e:          # librccl missing on rank 0: everybody learns it from the broadcast
            box, err = [None], str(e)
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            try:
                comm = _lib.Comm(ctx, box[0], ws, rank)
            except _lib.MotifsError as e:
                comm, err = None, str(e)
            ok = host_all_reduce(torch.tensor([1 if comm is not None else 0]), dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                return RcclReducer(comm), "RCCL communicator created through the C ABI"
            if comm is not None:
                comm.close()
        return HostReducer(ctx), f"RCCL communicator unavailable ({err}); host-staged torch.distributed sums"
    return HostReducer(ctx), "host-staged torch.distributed sums (requested)"


def allreduce_sum_(t, reducer=None):
    """In-place sum over ranks (no-op for a single process)."""
    if world()[1] == 1:
        return t
    r = reducer or HostReducer()
    return r.sum_i64_(t) if t.dtype == torch.int64 else r.sum_f32_(t)


def dp_train_step(model, codes_ptr, n_groups_local, loss_t, grad_t, n_groups_total=None, reducer=None):
    """One data-parallel optimiser step: local summed gradient -> sum over ranks -> mean -> AdaBelief (identical on
    every rank, so the replicas stay bit-identical).  n_groups_local may be 0: the rank contributes zeros and still
    joins the exchange.  n_groups_total: mini-batches over all ranks; summed over ranks when not given."""
    _, ws = world()
    if n_groups_total is None:
        n_total = torch.tensor([n_groups_local], dtype=torch.int64)
        if ws > 1:
            HostReducer().sum_i64_(n_total)
        n_groups_total = int(n_total.item())
    if isinstance(reducer, RcclReducer) or ws == 1:
        comm = reducer.comm if isinstance(reducer, RcclReducer) else None
        model.dp_train_step_dev(comm, codes_ptr, n_groups_local, n_groups_total, loss_t.data_ptr(), grad_t.data_ptr())
        return loss_t
    reducer = reducer or HostReducer(model.ctx)
    if n_groups_local > 0:
        model.loss_grad_dev(codes_ptr, n_groups_local, loss_t.data_ptr(), grad_t.data_ptr())
    else:
        model.ctx.synchronize()
        grad_t.zero_()
    reducer.sum_f32_(grad_t)          # synchronises the context's stream before reading, the device after writing
    model.adabelief_dev(grad_t.data_ptr(), 1.0 / float(n_groups_total))
    return loss_t


def gather_hits(local_hits, local_scores):
    """Concatenate per-rank hit records in rank order (= sequence-block order).  Host arrays."""
    rank, ws = world()
    if ws == 1:
        return local_hits, local_scores
    objs = [None] * ws
    dist.all_gather_object(objs, (local_hits, local_scores))
    return np.concatenate([o[0] for o in objs]), np.concatenate([o[1] for o in objs])


def records_to_dicts(found_fwd, found_rc, num_motifs):
    """positions[m][n] / use_comp[m][n] as gpu_scan builds them (modify_w_found!, _h3_1_alignment.jl:38-52, forward records
    then reverse ones): what a consumer of the scan sees.  Two record lists that differ only in the global order of
    their (m, n) groups give equal dictionaries."""
    from .scan import modify_w_found

    pos = [dict() for _ in range(num_motifs)]
    sco = [dict() for _ in range(num_motifs)]
    comp = [dict() for _ in range(num_motifs)]
    modify_w_found(found_fwd[0], found_fwd[1], pos, sco, comp, rc=False)
    modify_w_found(found_rc[0], found_rc[1], pos, sco, comp, rc=True)
    return pos, sco, comp


def sharded_gpu_scan(ctx, pwms, lens, codes, reducer=None, batch=_lib.SCAN_BATCH, align=None):
    """gpu_scan (_h3_1_alignment.jl:89-99) over a read matrix sharded across the ranks: every rank scans its block
    with n0 = its first read, the per-PWM histograms are summed, the records concatenated in rank order.
    align = batch (the default): shard edges on whole ordering batches, the concatenation IS the single-device record
    list bit for bit, but 20 batches over 8 ranks split 3,3,3,3,2,2,2,2.  align = 1 (or the 6 of a mini-batch): even
    shards; the global order is then sequence-block-major (each shard numbers its ordering batches from its own first
    read), while for every (m, n) the records keep the order modify_w_found! depends on - ascending l per strand - so
    the dictionaries are the single-device ones (records_to_dicts).
    codes: (N, L) uint8 host array, the same on every rank.  Returns, on every rank,
    ((found_fwd, score_fwd), (found_rc, score_rc), counts[2, K])."""
    rank, ws = world()
    N, L = codes.shape
    K = len(lens)
    lo, hi = shard_range(N, rank, ws, align=batch if align is None else align)
    n_loc = hi - lo
    counts = torch.zeros((2, K), dtype=torch.int64, device=f"cuda:{ctx.device}")
    torch.cuda.synchronize(counts.device)      # the fill above is on torch's stream; an empty rank goes straight to the sum
    out = []
    if n_loc > 0:
        raw = torch.from_numpy(np.ascontiguousarray(codes[lo:hi])).to(counts.device)
        dcodes = torch.zeros(ctx.codes_bytes(n_loc, L), dtype=torch.uint8, device=counts.device)
        torch.cuda.synchronize(counts.device)
        ctx.encode_dev(raw.data_ptr(), _lib.DATA_CODES_U8, n_loc, L, dcodes.data_ptr())
        need = ctx.pwm_scan_hits_both_dev(pwms, lens, dcodes.data_ptr(), n_loc, L, None, None, 0, n0=lo, batch=batch)
        cap = max(max(need), 1)
        hits = [torch.empty((cap, 3), dtype=torch.int32, device=counts.device) for _ in range(2)]
        hsc = [torch.empty(cap, dtype=torch.int16, device=counts.device) for _ in range(2)]
        torch.cuda.synchronize(counts.device)
        got = ctx.pwm_scan_hits_both_dev(pwms, lens, dcodes.data_ptr(), n_loc, L, [h.data_ptr() for h in hits],
                                         [s.data_ptr() for s in hsc], cap, n0=lo, batch=batch, counts_ptr=counts.data_ptr())
        for rc in (0, 1):
            f = hits[rc][: got[rc]].cpu().numpy().view(np.uint32).reshape(-1, 3)
            rec = np.zeros(got[rc], dtype=_lib.HIT_DTYPE)
            rec["m"], rec["n"], rec["l"] = f[:, 0], f[:, 1], f[:, 2]
            out.append((rec, hsc[rc][: got[rc]].cpu().numpy().view(np.float16)))
    else:
        out = [(np.zeros(0, dtype=_lib.HIT_DTYPE), np.zeros(0, dtype=np.float16)) for _ in range(2)]
    if ws > 1:
        (reducer or HostReducer(ctx)).sum_i64_(counts)
    ctx.synchronize()
    fwd = gather_hits(*out[0])
    rcs = gather_hits(*out[1])
    return fwd, rcs, counts.cpu().numpy()
