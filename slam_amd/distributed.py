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


class BucketedAllReduce:
    """Gradient sum over the ranks in buckets that start while the backward pass still runs.  The flat gradient
    buffer is laid out in the order the gradients complete; the backward pass reports the offset up to which they
    are final (`ready`), and a bucket's all-reduce starts as soon as at least `min_elems` new elements are (or
    `force`), so only the last, small bucket is exposed after the last kernel.  `started` lists the [begin, end)
    ranges in the order they were issued (every rank issues the same ones: the model fixes them)."""

    def __init__(self, flat, min_elems=1 << 21):
        self.flat, self.min_elems = flat, int(min_elems)
        self.begin, self.handles, self.started = 0, [], []

    def ready(self, end, force=False):
        end = int(end)
        if end > self.begin and (force or end - self.begin >= self.min_elems):
            self.handles.append(start_bucket_allreduce(self.flat, self.begin, end))
            self.started.append((self.begin, end))
            self.begin = end

    def finish(self, end):
        self.ready(end, force=True)
        finish(self.handles)
        return self.started


def require_equal_over_ranks(value, what):
    """Raises on every rank unless `value` (an int) is the same on all of them."""
    if not initialised() or world_size() == 1:
        return
    t = torch.tensor([int(value), -int(value)], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    hi, lo = int(t[0].item()), -int(t[1].item())
    if hi != lo:
        raise RuntimeError("%s differs between the ranks (%d .. %d; this rank %d)" % (what, lo, hi, int(value)))


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


def all_ranks_ok(ok, device=None):
    """True when `ok` holds on every rank.  A step that allocates or builds something per rank (a model, a plan)
    and then enters a loop of collectives calls this in between - with ok=False from its except handler - so that
    a failure on one rank makes every rank leave together instead of the others blocking in their first all-reduce."""
    return bool(agree_min(1 if ok else 0, device=device))
