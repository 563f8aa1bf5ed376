"""Child process of tests/test_encoder_fullsize_gpu.py (the library reads its dispatch switches once per
process, and a process group is set up before the first GPU call, so both need a process of their own).

  python tests/encoder_worker.py fullsize OUT.npz      forward + loss + backward at batch 128, 224x224, full width
  python tests/encoder_worker.py layers OUT.npz        every layer's kernels on seeded inputs of its real shape, one layer at a time
  python tests/encoder_worker.py benchdp OUT.npz       bench.py's multi-rank training-loop leg under an RCCL group of one
  python tests/encoder_worker.py nccl1 OUT.npz         train_step() under an RCCL process group of one
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sample_idx(n, count, seed):
    rs = np.random.RandomState(seed)
    return np.unique(rs.randint(0, n, size=count))


def fullsize(out):
    import torch
    from slam_amd.network.cnn_model import VGG16Model
    B = int(os.environ.get("WORKER_BATCH", "128"))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, 224, 224, 4, generator=g).to(dev)
    model = VGG16Model(B, x, 6, 1e-3, seed=5)
    model.build_graph()
    model.add_loss(torch.eye(6, device=dev), (0.1 * torch.randn(B, 6, generator=g)).to(dev))
    model.forward()
    model.compute_loss(with_gradient=True)
    model.backward()
    torch.cuda.synchronize()
    rec = {"loss": model.loss.cpu().numpy(), "output": model.network_output.cpu().numpy()}
    for i, a in enumerate(model._act):
        # (the first layer keeps no z by default: its normalised output stands for it)
        z = (a["y"] if i == 0 else a["z"]).reshape(-1)
        idx = torch.from_numpy(sample_idx(z.numel(), 8192, 100 + i)).to(dev)
        rec["z_" + a["name"]] = z[idx].float().cpu().numpy()
    for name, gr in model.grads.items():
        f = gr.reshape(-1)
        idx = torch.from_numpy(sample_idx(f.numel(), 20000, 7)).to(dev)
        rec["g_" + name.replace("/", "__")] = f[idx].float().cpu().numpy()
        rec["n_" + name.replace("/", "__")] = np.array([float(f.double().norm())])
    # exact integer checksums of the bit patterns (order-independent sums of the values read as integers): every
    # activation the backward pass reads and every gradient, all elements
    for i, a in enumerate(model._act):
        for key in ("z", "y", "pooled"):
            if key in a and not (key == "z" and i == 0) and not (key == "y" and a["pool"]):
                rec["ck_%s_%s" % (key, a["name"])] = np.array([int(a[key].view(torch.int16).to(torch.int64).sum())])
    rec["ck_grads"] = np.array([int(model.flat_grads.view(torch.int32).to(torch.int64).sum())])
    for name, gr in model.grads.items():
        rec["ck_g_" + name.replace("/", "__")] = np.array([int(gr.contiguous().view(torch.int32).to(torch.int64).sum())])
    np.savez(out, **rec)


def graphdet(out):
    """Training steps replayed from a HIP graph back to back, with no synchronisation in between and a learning
    rate that changes every step, against the same steps launched eagerly; run with SLAMHIP_DETERMINISTIC=1 the two
    must agree bit for bit."""
    import torch
    from slam_amd.network.cnn_model import VGG16Model

    def make():
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(3)
        x = torch.randn(8, 64, 64, 4, generator=g).to(dev)
        model = VGG16Model(8, x, 6, 1e-3, fc_dim=256, seed=1)
        model.build_graph()
        model.add_loss(torch.eye(6, device=dev), torch.randn(8, 6, generator=g).to(dev))
        model.add_optimizer(optimizer='AdamOptimizer')
        model._optimizer.learning_rate = lambda step: 1e-4 * (1.0 + (step % 3)) * 0.5 ** (step // 4)   # changes every step
        return model

    from slam_amd.network import encoder_ops as ops
    steps = 12
    eager = make()
    for _ in range(2):
        eager.train_step()
    lr_t_dev = torch.zeros(1, device=eager.device)
    for _ in range(steps):          # the recorded step's launches, one by one (the step size computed as the replay's)
        lr = eager._optimizer.learning_rate(eager.global_step)
        lr_t_dev.fill_(ops.adam_lr_t(lr, eager.global_step + 1))
        eager.forward()
        eager.compute_loss(with_gradient=True)
        eager.backward()
        ops.adam_step_dev(eager.flat_params, eager.flat_grads, eager._adam_m, eager._adam_v, lr_t_dev,
                          shadow=eager._fc_shadow)
        eager._operands_stale, eager._fc_shadow_fresh = True, True
        eager.global_step += 1
    torch.cuda.synchronize()
    replayed = make()
    step = replayed.capture_train_step(warmup=2)
    reads = os.environ.get("WORKER_READS", "0") == "1"
    for _ in range(steps):          # the host runs ahead: nothing waits for the GPU between the steps ...
        step()
        if reads:                   # ... or reads and allocates in between, as a training loop's logging does
            assert all(bool(torch.isfinite(t).all()) for t in list(replayed.grads.values()) + list(replayed.params.values()))
            float(replayed.loss)
    torch.cuda.synchronize()
    np.savez(out, p_eager=eager.flat_params.cpu().numpy(), p_graph=replayed.flat_params.cpu().numpy(),
             l_eager=eager.loss.cpu().numpy(), l_graph=replayed.loss.cpu().numpy(),
             steps=np.array([eager.global_step, replayed.global_step]))


def layers(out):
    """Each convolution layer by itself at its real shape on the same seeded operands in every process:
    forward with statistics, filter gradient, input gradient; the dense layers' products."""
    import torch
    from slam_amd.network import encoder_ops as ops
    from slam_amd.network.cnn_model import VGG16Model
    B = int(os.environ.get("WORKER_BATCH", "128"))
    dev = torch.device("cuda:0")
    x = torch.zeros(B, 224, 224, 4, device=dev)
    model = VGG16Model(B, x, 6, 1e-3, seed=5)
    model.build_graph()
    model._refresh_operands()        # the bfloat16 weight operands (forward() does this on its first call)
    torch.manual_seed(11)
    rec = {}

    def keep(key, t, count=16384, seed=3):
        f = t.reshape(-1)
        idx = torch.from_numpy(sample_idx(f.numel(), count, seed)).to(dev)
        rec[key] = f[idx].float().cpu().numpy()

    for i, a in enumerate(model._act):
        n = a["name"]
        if i == 0:
            continue          # the 4-channel layer has one kernel only
        prev = model._act[i - 1]
        h = prev["pooled"] if prev["pool"] else prev["y"]
        h.copy_(torch.randn(h.shape, device=dev).clamp_(min=0))          # post-ReLU-like input
        ops.conv3x3_stats(h, model._w_fwd[n], a["z"], a["sums"], model._stats_ws)
        keep("z_" + n, a["z"])
        rec["sums_" + n] = a["sums"].cpu().numpy()
        dz = torch.empty_like(a["z"]).copy_(torch.randn(a["z"].shape, device=dev))
        ops.wgrad(h, dz, model.grads[n + "/weights"])
        rec["dw_" + n] = model.grads[n + "/weights"].cpu().numpy()
        d_in = torch.empty_like(h)
        ops.conv3x3(dz, model._w_dgrad[n], out_bf16=d_in, flip=True)
        keep("din_" + n, d_in)
        del dz, d_in
    torch.cuda.synchronize()
    np.savez(out, **rec)


def nccl1(out):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", os.environ.get("WORKER_PORT", "29533"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))   # before any GPU call
    from slam_amd import distributed as dp
    from slam_amd.network.cnn_model import VGG16Model

    def make():
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(3)
        x = torch.randn(4, 64, 64, 4, generator=g).to(dev)
        model = VGG16Model(4, x, 6, 1e-3, fc_dim=256, seed=1)
        model.build_graph()
        model.add_loss(torch.eye(6, device=dev), torch.randn(4, 6, generator=g).to(dev))
        step = model.add_optimizer(optimizer='AdamOptimizer')
        model._optimizer.learning_rate = 1e-4
        return model, step

    assert dp.initialised() and dp.world_size() == 1
    # the gradients of one forward pass: through backward_with_allreduce() (two bucketed all-reduces over
    # RCCL) and through backward() alone, from the same activations
    model, step = make()
    model.forward()
    model.compute_loss(with_gradient=True)
    model.backward_with_allreduce()
    torch.cuda.synchronize()
    g_dp = model.flat_grads.cpu().numpy()
    buckets, fc_end, n_params = np.array(model.allreduce_buckets), model._fc_end, model._nparams
    model.backward()
    torch.cuda.synchronize()
    g_1 = model.flat_grads.cpu().numpy()
    # whole training steps under the group
    model, step = make()
    l_dp = [float(step()) for _ in range(3)]
    p_dp = model.flat_params.cpu().numpy()
    dist.destroy_process_group()
    assert not dp.initialised()
    model, step = make()
    l_1 = [float(step()) for _ in range(3)]
    np.savez(out, l_dp=np.array(l_dp), l_1=np.array(l_1), g_dp=g_dp, g_1=g_1, p_dp=p_dp, buckets=buckets,
             fc_end=fc_end, n_params=n_params)


def benchdp(out):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", os.environ.get("WORKER_PORT", "29534"))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    import bench
    from slam_amd import frame_pairs as fp
    from slam_amd.utils import synthetic as syn
    B, H, W = 8, 480, 640
    K = fp.intrinsics(1)
    gt = torch.from_numpy(syn.pose_pairs(B, seed=3)).cuda()
    rel = fp.relative_pose(gt[:, 0].contiguous(), gt[:, 1].contiguous()).reshape(B, 1, 4, 4)
    buf = [torch.from_numpy(syn.depth_frames(B, H, W, seed=1, kind="plane")).cuda(),
           torch.from_numpy(syn.rgb_frames(B, H, W, seed=1)).cuda(),
           torch.from_numpy(syn.depth_frames(B, H, W, seed=2, kind="plane")).cuda().reshape(B, 1, H, W),
           torch.from_numpy(syn.rgb_frames(B, H, W, seed=2)).cuda().reshape(B, 1, H, W, 3), rel]
    r = bench.train_loop(fp, buf, K, 1.0, 2, world=1, dp=True)
    dist.destroy_process_group()
    np.savez(out, pairs_per_s=r["pairs_per_s"], loss_finite=r["loss_finite"], batch=r["global_encoder_batch"],
             stages=np.array(list(r["stage_ms"].values())))


if __name__ == "__main__":
    {"fullsize": fullsize, "layers": layers, "nccl1": nccl1, "benchdp": benchdp, "graphdet": graphdet}[sys.argv[1]](sys.argv[2])
