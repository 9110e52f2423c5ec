"""Data-parallel plumbing: one process per GPU, sequences sharded in contiguous
blocks, parameters and the PWM bank replicated.  The only exchanges are
  * training: one all-reduce (sum) of the flat gradient [dD | dF | dvecs] per optimiser step, after
    which every rank applies the identical AdaBelief update;
  * scanning: one all-reduce (sum) of the K-entry hit histogram per strand.
The reference has no counterpart (single GPU, no collectives); torch.distributed is used as plumbing
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_range(n_items, rank, world_size, align=1):
    """Contiguous block [lo, hi) of `n_items` for `rank`; block edges fall on multiples of `align`
    (align = hp.batch_size keeps every 6-sequence mini-batch on one rank; align = 5000 keeps the
    scan's ordering batches whole).  Blocks differ by at most one aligned unit; the tail goes last."""
    units = (n_items + align - 1) // align
    base, extra = divmod(units, world_size)
    lo_u = rank * base + min(rank, extra)
    hi_u = lo_u + base + (1 if rank < extra else 0)
    return min(lo_u * align, n_items), min(hi_u * align, n_items)


def allreduce_sum_(t):
    """In-place sum over ranks (no-op for a single process)."""
    if world()[1] > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def dp_train_step(model, codes_ptr, n_groups_local, loss_t, grad_t, n_groups_total=None):
    """One data-parallel optimiser step: local summed gradient -> all-reduce -> mean -> AdaBelief
    (identical on every rank, so the replicas stay bit-identical).  n_groups_total: the number of mini-batches
    over all ranks when the caller knows it (equal shards); otherwise it is all-reduced too."""
    _, ws = world()
    model.loss_grad_dev(codes_ptr, n_groups_local, loss_t.data_ptr(), grad_t.data_ptr())
    allreduce_sum_(grad_t)
    if n_groups_total is None:
        n_total = torch.tensor([n_groups_local], dtype=torch.int64, device=grad_t.device)
        allreduce_sum_(n_total)
        n_groups_total = int(n_total.item())
    model.adabelief_dev(grad_t.data_ptr(), 1.0 / float(n_groups_total))
    return loss_t


def gather_hits(local_hits, local_scores):
    """Concatenate per-rank hit records in rank order (= sequence-block order).  Host arrays."""
    rank, ws = world()
    if ws == 1:
        return local_hits, local_scores
    objs = [None] * ws
    dist.all_gather_object(objs, (local_hits, local_scores))
    import numpy as np

    return np.concatenate([o[0] for o in objs]), np.concatenate([o[1] for o in objs])
