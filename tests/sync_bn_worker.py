"""Child process of tests/test_sync_bn_gpu.py: one rank of a data-parallel forward + backward of the encoder
with the normalisation statistics kept over all ranks (a process group has to exist before the first GPU call,
and two ranks are two processes; both use GPU 0, the small statistics travel over gloo).

  python tests/sync_bn_worker.py RANK WORLD PORT SYNC OUT.npz     SYNC = 1: statistics over all ranks, 0: per rank
  python tests/sync_bn_worker.py full OUT.npz                     the whole batch in one process, no group
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B_TOTAL, HW, FC = 8, 64, 256


def batch():
    import torch
    g = torch.Generator().manual_seed(33)
    x = torch.randn(B_TOTAL, HW, HW, 4, generator=g)
    gt = 0.1 * torch.randn(B_TOTAL, 6, generator=g)
    return x, gt


def run(x, gt, sync, out, allreduce):
    import torch
    from slam_amd.network.cnn_model import VGG16Model
    dev = torch.device("cuda:0")
    model = VGG16Model(x.shape[0], x.to(dev), 6, 1e-3, fc_dim=FC, seed=5, sync_batch_norm=sync)
    model.build_graph()
    model.add_loss(torch.eye(6, device=dev), gt.to(dev))
    model.forward()
    model.compute_loss(with_gradient=True)
    if allreduce:
        model.backward_with_allreduce()     # gradient SUM over the ranks (the optimizer divides by the world size)
    else:
        model.backward()
    torch.cuda.synchronize()
    rec = {"loss": model.loss.cpu().numpy(), "output": model.network_output.cpu().numpy()}
    for a in model._act:
        rec["sums_" + a["name"]] = a["sums"].cpu().numpy()
    for name, gr in model.grads.items():
        rec["g_" + name.replace("/", "__")] = gr.float().cpu().numpy()
    np.savez(out, **rec)


def main():
    if sys.argv[1] == "full":
        x, gt = batch()
        run(x, gt, False, sys.argv[2], False)
        return
    rank, world, port, sync, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    x, gt = batch()
    per = B_TOTAL // world
    run(x[rank * per:(rank + 1) * per], gt[rank * per:(rank + 1) * per], bool(sync), out, True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
