"""Data-parallel plumbing of the frame-pair path: one process per GPU under
torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).

Frame pairs are independent, so the per-pixel kernels need no collective: each
rank owns a contiguous shard of the pairs.  The only exchanges are the totals of
the reduced residuals (a few numbers per step) and, when the encoder trains, the
gradient mean, sent as two buckets so that the first all-reduce overlaps the rest
of the backward pass.
"""
import torch
import torch.distributed as dist


def initialised():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if initialised() else 1


def rank():
    return dist.get_rank() if initialised() else 0


def shard_range(n_items, rank_=None, world=None):
    """[begin, end) of the items rank owns: contiguous, sizes differ by at most one, the
    first n_items % world ranks hold the extra item."""
    r = rank() if rank_ is None else rank_
    w = world_size() if world is None else world
    base, extra = divmod(int(n_items), w)
    begin = r * base + min(r, extra)
    return begin, begin + base + (1 if r < extra else 0)


def start_bucket_allreduce(flat, begin, end):
    """Asynchronous sum of flat[begin:end] over the ranks; returns the work handle (None
    without a process group)."""
    if not initialised() or end <= begin:
        return None
    return dist.all_reduce(flat[begin:end], op=dist.ReduceOp.SUM, async_op=True)


def finish(handles):
    for h in handles:
        if h is not None:
            h.wait()


def total_pair_sums(sums, counts):
    """Sum the per-pair reductions of this rank's shard and add them up over the ranks:
    returns (float64 [4] sums, int64 [4] counts) of the whole job on every rank."""
    s = sums.reshape(-1, sums.shape[-1]).sum(0).to(torch.float64)
    c = counts.reshape(-1, counts.shape[-1]).to(torch.int64).sum(0)
    if initialised():
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return s, c


_stats_group = None


def average_over_ranks(t):
    """t (a small float32 tensor of per-rank sums) -> its mean over the ranks, in place.  Used for the
    batch-norm statistics kept over all ranks; runs on a process group of its own so that it does not
    queue behind the gradient buckets' all-reduce on the default group.  Every rank must call it the same
    number of times in the same order."""
    global _stats_group
    if not initialised() or world_size() == 1:
        return t
    if _stats_group is None:
        _stats_group = dist.new_group()     # (collective: every rank gets here at its first statistics exchange)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_stats_group)
    t.mul_(1.0 / world_size())
    return t


def agree_min(value, device=None):
    """The smallest `value` (an int) over the ranks: every rank must run the same number of training
    steps, because every step issues the gradient all-reduces."""
    if not initialised():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())
