"""Two ranks on one GPU over gloo: all-reduces of slices of one GPU buffer.  A check of the rehearsal backend, not of the
product.  Observed on this image: slices that start at the buffer's first element work (120 M floats in 0.3 s); the first
all-reduce on a slice with a non-zero offset never returns - so the encoder's bucketed gradient all-reduce cannot be
rehearsed over gloo with GPU tensors (bench.py's BENCH_REHEARSAL_BACKEND covers the frame-pair legs only: run it with
--encoder-steps 0 --train-steps 0).  RCCL, which the real runs use, takes such views as every bucketed data-parallel
trainer passes them."""
import os, sys, time, torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
flat = torch.ones(134_000_000, device="cuda")
cuts = ((0, 120_000_000), (120_000_000, 127_000_000), (127_000_000, 132_000_000), (132_000_000, 134_000_000))
t0 = time.time()
for a, b in cuts:
    dist.all_reduce(flat[a:b], async_op=True).wait()
torch.cuda.synchronize()
print(rank, "four buckets one at a time %.2f s" % (time.time() - t0), flush=True)
busy = torch.randn(8192, 8192, device="cuda")
t0 = time.time()
for a, b in cuts:
    for _ in range(20):
        busy = (busy @ busy).clamp_(-1, 1)      # a queue of kernels ahead of every collective
    dist.all_reduce(flat[a:b], async_op=True).wait()
torch.cuda.synchronize()
print(rank, "the same behind a busy stream %.2f s" % (time.time() - t0), flush=True)
t0 = time.time()
hs = [dist.all_reduce(flat[a:b], async_op=True) for a, b in cuts[:2]]
for h in hs:
    h.wait()
torch.cuda.synchronize()
print(rank, "two in flight %.2f s" % (time.time() - t0), flush=True)
