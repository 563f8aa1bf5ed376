"""VGG16-style pose-regression encoder on the MI355X kernels, with the
reference's class and method names (slam/network/cnn_model.py:14-186).

The reference builds a TensorFlow graph and lets the session run it; here
`build_graph`, `add_loss` and `add_optimizer` lay out parameters, activations
and gradients in HBM once, and `apply_gradient_op()` (the object `add_optimizer`
returns, as in the reference) executes one training step as a fixed sequence of
kernel launches through the C ABI: forward, loss, hand-derived backward,
gradient all-reduce when torch.distributed is initialised, optimizer update.
torch holds the memory and the stream and nothing else; there is no autograd
and no CPU fallback.

Network (cnn_model.py:31-80): 12 conv3x3 SAME layers in five blocks
(64,64 | 128,128 | 256,256 | 512x3 | 512x3), each conv -> batch-norm with batch
statistics -> ReLU, 2x2 max-pool after every block, two ReLU fully connected
layers and a linear output layer of `output_dim`.
"""
import math
import os

import torch

from . import encoder_ops as ops
from .. import distributed as dp
from .model_config import get_config_provider
from ..utils.time_utils import time_it

# (name, output channels, max-pool after)   cnn_model.py:37-62
CONV_LAYERS = (
    ("conv1_1", 64, False), ("conv1_2", 64, True),
    ("conv2_1", 128, False), ("conv2_2", 128, True),
    ("conv3_1", 256, False), ("conv3_2", 256, True),
    ("conv4_1", 512, False), ("conv4_2", 512, False), ("conv4_3", 512, True),
    ("conv5_1", 512, False), ("conv5_2", 512, False), ("conv5_3", 512, True),
)
OUT_PAD = 64   # the output layer is computed 64 wide (zero columns beyond output_dim)


def _round_up(n, m):
    return (n + m - 1) // m * m


def model_flops(batch, height, width, in_channels=4, fc_dim=4096, output_dim=6):
    """(forward, forward + backward) floating-point operations of one step, a multiply-add
    counted as two; backward = data gradient + weight gradient of every layer, except that
    the first layer has no data gradient."""
    fwd, first = 0, None
    h, w, c = height, width, in_channels
    for _, cout, pool in CONV_LAYERS:
        f = 2 * batch * h * w * 9 * c * cout
        first = f if first is None else first
        fwd += f
        c = cout
        if pool:
            h, w = h // 2, w // 2
    fwd += 2 * batch * (h * w * c * fc_dim + fc_dim * fc_dim + fc_dim * output_dim)
    return fwd, 3 * fwd - first


def timed_training_steps(batch, steps, warmup=2, height=224, width=224, seed=0, data_seed=0, graph=None):
    """Random-initialised encoder on synthetic input: `steps` training steps (forward, loss,
    backward, gradient all-reduce when torch.distributed is initialised, Adam) timed with
    one event pair on the launching stream.  Returns a dict for benchmark reports."""
    from .. import distributed as dp
    dev = torch.device("cuda", torch.cuda.current_device())
    try:
        g = torch.Generator(device=dev).manual_seed(data_seed)     # data differ per rank, parameters do not
        x = torch.randn(batch, height, width, 4, device=dev, generator=g)
        model = VGG16Model(batch, x, 6, 1e-3, seed=seed)
        model.build_graph()
        model.add_loss(torch.eye(6, device=dev), 0.1 * torch.randn(batch, 6, device=dev, generator=g))
        step = model.add_optimizer()
        model._optimizer.learning_rate = 1e-5
    except Exception:
        dp.all_ranks_ok(False, device=dev)       # the other ranks must not enter the steps' all-reduces alone
        raise
    if not dp.all_ranks_ok(True, device=dev):
        raise RuntimeError("timed_training_steps: another rank failed to build its model")
    if graph is None:    # replay a recorded step wherever it is supported: one GPU, Adam
        graph = not (dp.initialised() and dp.world_size() > 1) and model._optimizer.kind == 'AdamOptimizer'
    if graph:
        step = model.capture_train_step(warmup=max(1, warmup))
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    step_ms = e0.elapsed_time(e1) / max(1, steps)

    def timed(fn):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / max(1, steps)

    def forward_backward():     # the step without the update (and without the all-reduce)
        model.forward()
        model.compute_loss(with_gradient=True)
        model.backward()

    fwd_bwd_ms = timed(forward_backward)
    forward_ms = timed(model.forward)
    _, flops = model_flops(batch, height, width, 4, model.fc_dim, 6)
    return dict(batch_per_gpu=batch, step_ms=step_ms, fwd_bwd_ms=fwd_bwd_ms, forward_ms=forward_ms, hip_graph=bool(graph),
                images_per_s_per_gpu=batch / (step_ms * 1e-3),
                tflop_per_step=flops / 1e12, achieved_tflops_per_gpu=flops / 1e12 / (step_ms * 1e-3),
                loss=float(model.loss), parameters=int(model.total_weights))


class _Optimizer:
    """Update rule + hyper-parameters of one tf.train optimizer (cnn_model.py:162-170)."""

    def __init__(self, kind, learning_rate):
        self.kind = kind
        self.learning_rate = learning_rate   # float or callable(global_step) -> float


class VGG16Model:

    def __init__(self, batch_size, rgbd_input_batch, output_dim, normalization_epsilon,
                 initial_params=None, fc_dim=4096, seed=0, batch_norm=True, sync_batch_norm=False):
        """rgbd_input_batch: float32 GPU tensor [batch, H, W, 4] that the caller refills in
        place before each step (the reference's placeholder, main/vgg16_rgbd.py:29).
        initial_params: optional {'conv1_2': {'weights': HWIO array, 'biases': ...}, ...}
        as the reference reads from its weights file (cnn_model.py:22-23, 129-137);
        layers without an entry (and conv1_1 always) start from a truncated normal.
        batch_norm=False gives the variant of slam/network/cnn_model_noBN.py:90-111:
        conv -> bias -> ReLU instead of conv -> batch-norm -> ReLU.
        sync_batch_norm: under torch.distributed the normalisation uses the statistics of the whole
        job's batch (the per-channel sums of the forward and of the backward reduction are averaged over
        the ranks), so that N ranks with batch b compute what one device computes on batch N * b, which
        is what the reference's single-device graph means by "the batch" (cnn_model.py:100-113).  Off
        (the default) every rank normalises over its own batch and no collective runs in the passes."""
        if not (rgbd_input_batch.is_cuda and rgbd_input_batch.dtype == torch.float32
                and rgbd_input_batch.dim() == 4 and rgbd_input_batch.shape[0] == batch_size):
            raise ValueError("rgbd_input_batch must be a float32 GPU tensor [batch, H, W, C]")
        _, H, W, _ = rgbd_input_batch.shape
        if H % 32 or W % 32:
            raise ValueError("input height and width must be multiples of 32 (five 2x2 pools)")
        self.network_input = rgbd_input_batch
        self.output_dim = int(output_dim)
        if not 0 < self.output_dim <= 8:
            raise ValueError("output_dim must be in 1..8")
        self.batch_size = int(batch_size)
        self.epsilon = float(normalization_epsilon)
        self.initial_params = initial_params or {}
        self.fc_dim = int(fc_dim)
        self.batch_norm = bool(batch_norm)
        self.sync_batch_norm = bool(sync_batch_norm)
        self.seed = seed
        self.total_weights = 0
        self.device = rgbd_input_batch.device
        self.global_step = 0
        self._built = False

    # ------------------------------------------------------------------ build
    def build_graph(self):
        """Lays out parameters and activations; returns the [batch, output_dim] float32
        tensor that every forward pass refreshes (cnn_model.py:31-80)."""
        B, H, W, cin = self.network_input.shape
        dev = self.device
        # ---- parameter table: fully connected first (their gradients are complete
        # first in the backward pass, so their all-reduce starts first)
        self._spec = []        # (name, kind, shape)
        feat = (H // 32) * (W // 32) * 512
        self._spec.append(("out_layer/weights", (self.fc_dim, OUT_PAD)))
        self._spec.append(("out_layer/bias", (OUT_PAD,)))
        self._spec.append(("fc7-conv/weights", (self.fc_dim, self.fc_dim)))
        self._spec.append(("fc7-conv/bias", (self.fc_dim,)))
        self._spec.append(("fc6-conv/weights", (feat, self.fc_dim)))
        self._spec.append(("fc6-conv/bias", (self.fc_dim,)))
        self._fc_end = None
        c_prev = cin
        conv_shapes = []
        for name, cout, _ in CONV_LAYERS:
            conv_shapes.append((name, c_prev, cout))
            c_prev = cout
        for name, ci, co in reversed(conv_shapes):
            self._spec.append((name + "/weights", (3, 3, ci, co)))
            if self.batch_norm:
                self._spec.append((name + "/bn", (2, co)))      # row 0 beta, row 1 scale
            else:
                self._spec.append((name + "/biases", (co,)))
        offs, total = {}, 0
        for name, shape in self._spec:
            if name == "conv5_3/weights":
                self._fc_end = total
            offs[name] = (total, shape)
            total += _round_up(math.prod(shape), 64)
        self._offsets, self._nparams = offs, total
        self.flat_params = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self.params = {n: self._view(self.flat_params, n) for n, _ in self._spec}
        self.grads = {n: self._view(self.flat_grads, n) for n, _ in self._spec}
        self._init_params(conv_shapes, feat)

        # ---- bfloat16 operand copies of the weights
        bf = dict(dtype=torch.bfloat16, device=dev)
        self._w_fwd, self._w_dgrad = {}, {}
        for name, ci, co in conv_shapes[1:]:
            self._w_fwd[name] = torch.empty(co, 9 * ci, **bf)
            self._w_dgrad[name] = torch.empty(ci, 9 * co, **bf)
        # the fully connected parameters lie first in the flat buffer: one bf16 shadow of that
        # range holds their [in][out] operands (written by the Adam kernel as it updates them)
        self._fc_shadow = torch.empty(self._fc_end, **bf)
        for name in ("fc6-conv", "fc7-conv", "out_layer"):
            k_in, n_out = self.params[name + "/weights"].shape
            self._w_fwd[name] = torch.empty(n_out, k_in, **bf)      # [out][in]
            self._w_dgrad[name] = self._view(self._fc_shadow, name + "/weights")    # [in][out]
        self.parameters_changed()

        # ---- activations
        f32 = dict(dtype=torch.float32, device=dev)
        self._x = torch.empty(B, H, W, cin, **bf)
        self._act = []     # per conv layer: dict(z, y, pooled, sums)
        h, w = H, W
        # the channel statistics of all layers lie in one buffer, cleared by one launch per forward pass
        self._sums_arena = torch.zeros(sum(2 * co for _, _, co in conv_shapes), **f32)
        sums_off = 0
        for (name, ci, co), (_, _, pool) in zip(conv_shapes, CONV_LAYERS):
            a = dict(name=name, cin=ci, cout=co, pool=pool,
                     z=torch.empty(B, h, w, co, **bf), y=torch.empty(B, h, w, co, **bf),
                     sums=self._sums_arena[sums_off:sums_off + 2 * co].view(2, co))
            sums_off += 2 * co
            if pool:
                h, w = h // 2, w // 2
                a["pooled"] = torch.empty(B, h, w, co, **bf)
            self._act.append(a)
        ws_bytes = max(ops.conv_stats_workspace_bytes(a["z"].numel() // a["cout"], a["cout"]) for a in self._act)
        self._stats_ws = torch.empty(max(ws_bytes, 16) // 4, **f32)
        self._fc = {}
        for name, k_in, n_out in (("fc6-conv", feat, self.fc_dim), ("fc7-conv", self.fc_dim, self.fc_dim),
                                  ("out_layer", self.fc_dim, OUT_PAD)):
            self._fc[name] = dict(z=torch.empty(B, n_out, **f32), a=torch.empty(B, n_out, **bf),
                                  dz=torch.empty(B, n_out, **bf), d_in=torch.empty(B, k_in, **f32))
        self._out_pad = torch.empty(B, OUT_PAD, **f32)
        self.network_output = torch.empty(B, self.output_dim, **f32)
        self._scratch = {}
        self.trace = None      # set to a dict to keep per-layer gradient copies (tests, debugging)
        # SLAM_FUSE_FIRST_BWD=0: the first layer's backward as two separate calls through dz
        self.fuse_first_backward = os.environ.get("SLAM_FUSE_FIRST_BWD", "1") not in ("", "0")
        # the first layer without a stored z: its forward writes y alone and its backward recomputes z from the
        # 4-channel input (SLAM_FIRST_RECOMPUTE=0: z is written and read as in every other layer)
        self.first_layer_recompute = (os.environ.get("SLAM_FIRST_RECOMPUTE", "1") not in ("", "0")
                                      and self.batch_norm and cin == 4 and not CONV_LAYERS[0][2])
        # one clearing launch per pass for all the accumulators the kernels add into (channel statistics in the
        # forward pass; filter gradients and batch-norm sums, i.e. the convolution part of flat_grads, in the
        # backward pass) instead of one per entry point (SLAM_MERGE_CLEARS=0: every entry point clears its own)
        self.merge_clears = os.environ.get("SLAM_MERGE_CLEARS", "1") not in ("", "0")
        self._conv_layout_table = [(self._offsets[n + "/weights"][0], ci, co, self._w_fwd[n], self._w_dgrad[n])
                                   for n, ci, co in conv_shapes[1:]]
        # True (or SLAM_KEEP_FULL_Y=1): pooled batch-norm layers also write their full-resolution output
        self.keep_full_y = os.environ.get("SLAM_KEEP_FULL_Y", "0") not in ("", "0")
        self._built = True
        return self.network_output

    def _view(self, flat, name):
        off, shape = self._offsets[name]
        return flat[off:off + math.prod(shape)].view(shape)

    def _init_params(self, conv_shapes, feat):
        """truncated normal (stddev 5e-2, cut at two stddev) unless initial values are given
        (cnn_model.py:129-137); batch-norm scale ones, beta zeros (cnn_model.py:105-106)."""
        g = torch.Generator().manual_seed(self.seed)

        def trunc(shape):
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, mean=0.0, std=5e-2, a=-0.1, b=0.1, generator=g)
            return t

        def initial(scope, key, shape, allow=True):
            init = self.initial_params.get(scope) if allow else None
            if init is not None and key in init:
                t = torch.as_tensor(init[key], dtype=torch.float32).reshape(shape)
                return t
            return trunc(shape)

        for i, (name, ci, co) in enumerate(conv_shapes):
            self.total_weights += 9 * ci * co
            self.params[name + "/weights"].copy_(initial(name, "weights", (3, 3, ci, co), allow=i > 0))
            if self.batch_norm:
                self.params[name + "/bn"][0].zero_()
                self.params[name + "/bn"][1].fill_(1.0)
            else:
                self.params[name + "/biases"].copy_(initial(name, "biases", (co,), allow=i > 0))
        for name, k_in, n_out in (("fc6-conv", feat, self.fc_dim), ("fc7-conv", self.fc_dim, self.fc_dim)):
            self.params[name + "/weights"].copy_(initial(name, "weights", (k_in, n_out)))
            self.params[name + "/bias"].copy_(initial(name, "biases", (n_out,)))
        D = self.output_dim
        self.params["out_layer/weights"][:, :D].copy_(trunc((self.fc_dim, D)))
        self.params["out_layer/bias"][:D].copy_(trunc((D,)))

    # the reference's builder methods, for callers that put layers of their own on a tensor
    def _extra_param(self, scope_name, key, shape, init_key, use_init=True):
        if not hasattr(self, "extra_params"):
            self.extra_params = {}
            self._extra_gen = torch.Generator().manual_seed(self.seed + 1)
        name = scope_name + "/" + key
        if name not in self.extra_params:
            init = self.initial_params.get(scope_name) if use_init else None
            if init is not None and init_key in init:
                t = torch.as_tensor(init[init_key], dtype=torch.float32).reshape(shape)
            else:
                t = torch.empty(shape)
                torch.nn.init.trunc_normal_(t, mean=0.0, std=5e-2, a=-0.1, b=0.1, generator=self._extra_gen)
            self.extra_params[name] = t.to(self.device)
        return self.extra_params[name]

    def add_conv_layer(self, scope_name, layer_input, filter_size, input_channels, output_channels,
                       padding='SAME', should_init_wb=True):
        """relu(batch_norm(conv2d(layer_input, weights))) with batch statistics, scale one and beta zero
        (cnn_model.py:87-115), applied at once to a GPU tensor [batch, H, W, input_channels] (float32 or
        bfloat16) on the same kernels as the encoder's own layers; returns bfloat16 [batch, H, W, output_channels].
        The weights live in `self.extra_params[scope_name + '/weights']` (float32 HWIO; taken from
        `initial_params[scope_name]` when should_init_wb and present, else truncated normal).  A layer made
        this way runs forward only: build_graph() fixes the layers that train_step() differentiates."""
        if list(filter_size) != [3, 3] or padding != 'SAME':
            raise ValueError("only 3x3 SAME convolutions are implemented")
        if not (layer_input.is_cuda and layer_input.dim() == 4 and layer_input.shape[3] == input_channels):
            raise ValueError("layer_input must be a GPU tensor [batch, H, W, input_channels]")
        if output_channels % 64:
            raise ValueError("output_channels must be a multiple of 64")
        w = self._extra_param(scope_name, "weights", (3, 3, input_channels, output_channels), "weights", should_init_wb)
        self.total_weights += 9 * input_channels * output_channels
        x = layer_input.contiguous()
        if x.dtype != torch.bfloat16:
            xb = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
            ops.cast_bf16(x.float(), xb)
            x = xb
        B, H, W, _ = x.shape
        z = torch.empty(B, H, W, output_channels, dtype=torch.bfloat16, device=x.device)
        if input_channels % 64 == 0:
            w_nk = torch.empty(output_channels, 9 * input_channels, dtype=torch.bfloat16, device=x.device)
            ops.conv_weight_layouts(w, w_nk, None)
            ops.conv3x3(x, w_nk, out_bf16=z)
        else:
            ops.conv3x3_small(x, w, z)
        sums = torch.empty(2, output_channels, dtype=torch.float32, device=x.device)
        ops.channel_stats(z, sums)
        scale = torch.ones(output_channels, dtype=torch.float32, device=x.device)
        beta = torch.zeros(output_channels, dtype=torch.float32, device=x.device)
        y = torch.empty_like(z)
        ops.bn_relu(z, sums, scale, beta, y, self.epsilon)
        return y

    def add_fc_layer(self, scope_name, layer_input, input_dim, out_dim):
        """relu(layer_input @ weights + bias) (cnn_model.py:117-124) applied at once to a GPU tensor
        [batch, input_dim]; returns float32 [batch, out_dim].  Forward only, parameters in
        `self.extra_params` like add_conv_layer."""
        if not (layer_input.is_cuda and layer_input.dim() == 2 and layer_input.shape[1] == input_dim):
            raise ValueError("layer_input must be a GPU tensor [batch, input_dim]")
        if input_dim % 64:
            raise ValueError("input_dim must be a multiple of 64")
        w = self._extra_param(scope_name, "weights", (input_dim, out_dim), "weights")
        b = self._extra_param(scope_name, "bias", (out_dim,), "biases")
        dev = layer_input.device
        x = torch.empty(layer_input.shape, dtype=torch.bfloat16, device=dev)
        ops.cast_bf16(layer_input.contiguous().float(), x)
        w_nk = torch.empty(out_dim, input_dim, dtype=torch.bfloat16, device=dev)
        ops.transpose_cast(w, w_nk)
        z = torch.empty(layer_input.shape[0], out_dim, dtype=torch.float32, device=dev)
        ops.gemm_nt(x, w_nk, out_f32=z)
        y = torch.empty_like(z)
        ops.bias_act(z, b, y_f32=y, relu=True)
        return y

    # ---------------------------------------------------------------- forward
    def parameters_changed(self):
        """To be called after the parameters were written from outside (initialisation, a restored
        checkpoint, a hand-made update): the bf16 operand copies are rebuilt before the next forward."""
        self._operands_stale = True
        self._fc_shadow_fresh = False
        if getattr(self, "_graph", None) is not None:
            # a recorded step cannot see this flag: bring the operand copies up to date now
            self._refresh_operands()
            self._operands_stale = True

    def _sync_stats(self):
        """The normalisation statistics are exchanged between the ranks in this pass."""
        on = self.sync_batch_norm and dp.initialised() and dp.world_size() > 1
        if on and not getattr(self, "_sync_checked", False):
            # the mean of the ranks' sums over ONE rank's row count is the whole batch's mean only if every rank
            # holds the same number of rows: agreed on once, at the first exchange (every rank gets here together)
            dp.require_equal_over_ranks(self.batch_size, "sync_batch_norm: the batch size per rank")
            self._sync_checked = True
        return on

    def _recompute_first(self):
        """The first layer runs without its z tensor in this pass (not under tracing, which keeps z, nor with
        statistics exchanged between ranks, whose backward runs in separate phases)."""
        return (self.first_layer_recompute and self.fuse_first_backward and self.trace is None
                and not self._sync_stats())

    def _refresh_operands(self):
        ops.conv_weight_layouts_multi(self.flat_params, self._conv_layout_table)     # every layer in one launch
        if not self._fc_shadow_fresh:
            ops.cast_bf16(self.flat_params[:self._fc_end], self._fc_shadow)
            self._fc_shadow_fresh = True
        for n in ("fc6-conv", "fc7-conv", "out_layer"):
            ops.transpose_bf16(self._w_dgrad[n], self._w_fwd[n])
        self._operands_stale = False

    def forward(self):
        """network_input -> network_output (batch statistics in the normalisation,
        as the reference does in training and inference alike)."""
        assert self._built, "call build_graph() first"
        if self._operands_stale:
            self._refresh_operands()
        ops.cast_bf16(self.network_input, self._x)
        h = self._x
        merged = self.merge_clears and self.batch_norm
        if merged:
            ops.zero_fill(self._sums_arena)
        before = ops.caller_clears_accumulators(True) if merged else None
        try:
            h = self._forward_convs(h)
        finally:
            if merged:
                ops.caller_clears_accumulators(before)
        return self._forward_dense(h)

    def _forward_convs(self, h):
        for i, a in enumerate(self._act):
            n = a["name"]
            if i == 0 and self._recompute_first():
                bn = self.params[n + "/bn"]
                ops.conv3x3_c4_bn_relu(h, self.params[n + "/weights"], bn[1], bn[0], a["y"], a["sums"], self._stats_ws,
                                       self.epsilon)
                h = a["y"]
                continue
            if self.batch_norm:     # the channel statistics come out of the convolution's epilogue
                if i == 0:
                    ops.conv3x3_small_stats(h, self.params[n + "/weights"], a["z"], a["sums"], self._stats_ws)
                else:
                    ops.conv3x3_stats(h, self._w_fwd[n], a["z"], a["sums"], self._stats_ws)
                if self._sync_stats():
                    # equal row counts on every rank: the mean of the ranks' sums over one rank's row count is
                    # the whole batch's mean (and mean square)
                    dp.average_over_ranks(a["sums"])
                bn = self.params[n + "/bn"]
                if a["pool"]:       # normalisation, ReLU and the block's max-pool in one pass
                    # the full-resolution y of a pooled layer has no reader (the next layer takes the
                    # pooled tensor, the backward pass recomputes the gate from z): written on request only
                    ops.bn_relu_pool(a["z"], a["sums"], bn[1], bn[0], a["y"] if self.keep_full_y else None,
                                     a["pooled"], self.epsilon)
                    h = a["pooled"]
                    continue
                ops.bn_relu(a["z"], a["sums"], bn[1], bn[0], a["y"], self.epsilon)
            elif i == 0:
                ops.conv3x3_small(h, self.params[n + "/weights"], a["z"])
            else:
                ops.conv3x3(h, self._w_fwd[n], out_bf16=a["z"])
            if not self.batch_norm:
                ops.bias_relu(a["z"], self.params[n + "/biases"], a["y"])
            h = a["y"]
            if a["pool"]:
                ops.maxpool2(a["y"], a["pooled"])
                h = a["pooled"]
        return h

    def _forward_dense(self, h):
        B = self.batch_size
        h = h.view(B, -1)
        for n in ("fc6-conv", "fc7-conv"):
            f = self._fc[n]
            ops.gemm_nt(h, self._w_fwd[n], out_f32=f["z"])
            ops.bias_act(f["z"], self.params[n + "/bias"], y_bf16=f["a"], relu=True)
            h = f["a"]
        f = self._fc["out_layer"]
        ops.gemm_nt(h, self._w_fwd["out_layer"], out_f32=f["z"])
        ops.bias_act(f["z"], self.params["out_layer/bias"], y_f32=self._out_pad, relu=False)
        self.network_output.copy_(self._out_pad[:, :self.output_dim])
        return self.network_output

    # ------------------------------------------------------------------- loss
    def add_loss(self, loss_weight, ground_truth):
        """loss = l2_loss((network_output - ground_truth) . loss_weight) (cnn_model.py:139-144).
        Both arguments are float32 GPU tensors the caller refills in place."""
        self._loss_weight, self._ground_truth = loss_weight, ground_truth
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._d_out = torch.zeros(self.batch_size, self.output_dim, dtype=torch.float32, device=self.device)
        self._d_out_pad = torch.zeros(self.batch_size, OUT_PAD, dtype=torch.float32, device=self.device)
        return self.loss

    def add_custom_loss(self, loss_and_gradient):
        """A loss defined on the graph's output outside this class, as the PoseNet trainer
        does (main/vgg16_posenet.py:16-25, 63): loss_and_gradient(network_output, loss, d_out)
        fills the [1] loss tensor and d loss / d network_output (d_out may be None)."""
        self._custom_loss = loss_and_gradient
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._d_out = torch.zeros(self.batch_size, self.output_dim, dtype=torch.float32, device=self.device)
        self._d_out_pad = torch.zeros(self.batch_size, OUT_PAD, dtype=torch.float32, device=self.device)
        return self.loss

    def compute_loss(self, with_gradient=True):
        if getattr(self, "_custom_loss", None) is not None:
            self._custom_loss(self.network_output, self.loss, self._d_out if with_gradient else None)
            return self.loss
        ops.l2loss(self.network_output, self._ground_truth, self._loss_weight, self.loss,
                   self._d_out if with_gradient else None)
        return self.loss

    # --------------------------------------------------------------- backward
    def _buf(self, key, like):
        t = self._scratch.get(key)
        if t is None or t.shape != like.shape:
            t = self._scratch[key] = torch.empty_like(like)
        return t

    def backward(self, on_fc_done=None, on_conv_block_done=None):
        """Gradients of `loss` with respect to every parameter, into `grads` (views of
        `flat_grads`).  on_fc_done() is called once the fully connected gradients are
        complete, on_conv_block_done(end) after every block of convolution layers with the offset
        up to which `flat_grads` is final (the data-parallel step starts all-reduces there)."""
        self._d_out_pad[:, :self.output_dim].copy_(self._d_out)
        d = self._d_out_pad
        order = ("out_layer", "fc7-conv", "fc6-conv")
        last_conv = self._act[-1]
        feat_in = (last_conv["pooled"] if last_conv["pool"] else last_conv["y"]).view(self.batch_size, -1)
        inputs = {"out_layer": self._fc["fc7-conv"]["a"], "fc7-conv": self._fc["fc6-conv"]["a"], "fc6-conv": feat_in}
        d_feat = self._buf("d_feat", feat_in)
        for n in order:
            f = self._fc[n]
            relu = n != "out_layer"
            ops.bias_act_bwd(d, f["a"] if relu else None, f["dz"], self.grads[n + "/bias"], relu=relu)
            ops.wgrad(inputs[n], f["dz"], self.grads[n + "/weights"])
            if n == "fc6-conv":
                ops.gemm_nt(f["dz"], self._w_dgrad[n], out_bf16=d_feat)
            else:
                ops.gemm_nt(f["dz"], self._w_dgrad[n], out_f32=f["d_in"])
                d = f["d_in"]
        if on_fc_done is not None:
            on_fc_done()
        g = d_feat.view_as(last_conv["pooled"] if last_conv["pool"] else last_conv["y"])
        if self.trace is not None:
            self.trace["d_feat"] = g.clone()
        if self.merge_clears:      # filter gradients, batch-norm sums and bias gradients of every convolution layer
            ops.zero_fill(self.flat_grads[self._fc_end:])
        before = ops.caller_clears_accumulators(True) if self.merge_clears else None
        try:
            self._backward_convs(g, on_conv_block_done)
        finally:
            if self.merge_clears:
                ops.caller_clears_accumulators(before)

    def _block_done(self, i, on_conv_block_done):
        """Layer i's gradients are written: at the first layer of a block, report how far flat_grads is final
        (the layers lie in the buffer in the order the backward pass finishes them)."""
        if on_conv_block_done is not None and (i == 0 or self._act[i - 1]["pool"]):
            on_conv_block_done(self._offsets[self._act[i - 1]["name"] + "/weights"][0] if i else self._nparams)

    def _backward_convs(self, g, on_conv_block_done=None):
        for i in range(len(self._act) - 1, -1, -1):
            a = self._act[i]
            n = a["name"]
            g_out = g.clone() if self.trace is not None else None
            dz = self._buf(("dz", a["z"].shape), a["z"])
            sync = self.batch_norm and self._sync_stats()
            if (i == 0 and self.fuse_first_backward and self.batch_norm and not a["pool"] and self.trace is None
                    and a["z"].shape[-1] == 64 and self._x.shape[-1] == 4 and not sync):
                # first layer: its dz has one reader, the filter gradient, which computes it on the way
                bn = self.params[n + "/bn"]
                if self._recompute_first():
                    ops.bn_relu_bwd_first_layer_recompute(g, self._x, self.params[n + "/weights"], a["sums"], bn[1],
                                                          bn[0], self.grads[n + "/bn"], self.grads[n + "/weights"],
                                                          self._stats_ws, self.epsilon)
                else:
                    ops.bn_relu_bwd_first_layer(g, a["z"], a["sums"], bn[1], bn[0], self.grads[n + "/bn"], self._x,
                                                self.grads[n + "/weights"], self.epsilon)
                self._block_done(i, on_conv_block_done)
                continue
            if self.batch_norm:
                bn = self.params[n + "/bn"]
                bwd = ops.bn_relu_pool_bwd if a["pool"] else ops.bn_relu_bwd
                # (pooled layers: pool backward + ReLU gate + batch-norm backward, no upsampled gradient tensor)
                if sync:    # the two reduced sums of the whole batch: reduce, average over the ranks, apply
                    bwd(g, a["z"], a["sums"], bn[1], bn[0], self.grads[n + "/bn"], dz, self.epsilon, phases=1)
                    dp.average_over_ranks(self.grads[n + "/bn"])
                    bwd(g, a["z"], a["sums"], bn[1], bn[0], self.grads[n + "/bn"], dz, self.epsilon, phases=2)
                else:
                    bwd(g, a["z"], a["sums"], bn[1], bn[0], self.grads[n + "/bn"], dz, self.epsilon)
            else:
                if a["pool"]:
                    dy = self._buf(("dy", a["y"].shape), a["y"])
                    ops.maxpool2_bwd(a["y"], a["pooled"], g, dy)
                    g = dy
                ops.bias_relu_bwd(g, a["y"], dz, self.grads[n + "/biases"])
            if i == 0:
                ops.conv3x3_small_wgrad(self._x, dz, self.grads[n + "/weights"])
            else:
                prev = self._act[i - 1]
                h_in = prev["pooled"] if prev["pool"] else prev["y"]
                ops.wgrad(h_in, dz, self.grads[n + "/weights"])
                d_in = self._buf(("din", h_in.shape), h_in)
                ops.conv3x3(dz, self._w_dgrad[n], out_bf16=d_in, flip=True)
                g = d_in
            if self.trace is not None:
                self.trace[n] = dict(g_out=g_out, dz=dz.clone(), d_in=g.clone() if i else None)
            self._block_done(i, on_conv_block_done)

    # -------------------------------------------------------------- optimizer
    def add_optimizer(self, optimizer=None):
        """Adam (or the optimizer the configuration names) on a learning rate of
        0.01 * 0.1^(global_step // 5) (cnn_model.py:146-160).  Returns the callable that
        runs one training step and gives back the loss tensor.  `optimizer` names the
        update rule directly, for trainers that do not read it from the configuration
        (main/vgg16_posenet.py:27-41)."""
        self.global_step = 0
        learning_rate = lambda step: 0.01 * 0.1 ** (step // 5)   # exponential_decay, staircase
        self._optimizer = self.get_optimizer(learning_rate) if optimizer is None else _Optimizer(optimizer, learning_rate)
        if self._optimizer.kind == 'AdamOptimizer':
            self._adam_m = torch.zeros_like(self.flat_params)
            self._adam_v = torch.zeros_like(self.flat_params)
        elif self._optimizer.kind == 'RMSPropOptimizer':
            self._rms = torch.ones_like(self.flat_params)    # TensorFlow starts the mean square at one
        self.apply_gradient_op = self.train_step
        return self.apply_gradient_op

    def get_optimizer(self, learning_rate):
        opt = get_config_provider().optimizer()
        if opt not in ('AdamOptimizer', 'GradientDescentOptimizer', 'RMSPropOptimizer'):
            raise ValueError("unknown optimizer %r" % (opt,))
        return _Optimizer(opt, learning_rate)

    def _apply_gradients(self, grad_scale):
        o = self._optimizer
        lr = o.learning_rate(self.global_step) if callable(o.learning_rate) else o.learning_rate
        if o.kind == 'AdamOptimizer':
            ops.adam_step(self.flat_params, self.flat_grads, self._adam_m, self._adam_v, lr,
                          step=self.global_step + 1, grad_scale=grad_scale, shadow=self._fc_shadow)
            self._operands_stale, self._fc_shadow_fresh = True, True
        elif o.kind == 'GradientDescentOptimizer':
            ops.sgd_step(self.flat_params, self.flat_grads, lr, grad_scale=grad_scale)
            self.parameters_changed()
        else:
            ops.rmsprop_step(self.flat_params, self.flat_grads, self._rms, lr, grad_scale=grad_scale)
            self.parameters_changed()
        self.global_step += 1

    def set_output_gradient(self, d_output):
        """d loss / d network_output from a head built on this encoder's output
        (main/lstm_rgbd.py:71-77), in place of the gradient of the encoder's own loss."""
        if getattr(self, "_d_out", None) is None:
            self._d_out = torch.zeros(self.batch_size, self.output_dim, dtype=torch.float32, device=self.device)
            self._d_out_pad = torch.zeros(self.batch_size, OUT_PAD, dtype=torch.float32, device=self.device)
        self._d_out.copy_(d_output)

    def backward_with_allreduce(self):
        """backward(); under torch.distributed the gradient sum over the ranks runs in buckets that start as the
        backward pass completes them: the fully connected layers first (they overlap the whole convolution
        backward), then every block of convolution layers of at least 2 Mi elements (blocks 5 and 4; the three
        wide-image blocks, 1.1 M elements together, go last).  Only that last bucket, 4.6 MB, is exposed after the
        last kernel.  The issued ranges are kept in `allreduce_buckets`."""
        from .. import distributed as dp
        if dp.initialised():
            buckets = dp.BucketedAllReduce(self.flat_grads)
            self.backward(on_fc_done=lambda: buckets.ready(self._fc_end, force=True), on_conv_block_done=buckets.ready)
            self.allreduce_buckets = buckets.finish(self._nparams)
        else:
            self.backward()

    def apply_sgd(self, learning_rate, grad_scale=1.0):
        """Plain gradient descent on every parameter (the update a head's optimizer applies
        to the encoder, slam/network/lstm_model.py:63-66)."""
        ops.sgd_step(self.flat_params, self.flat_grads, learning_rate, grad_scale=grad_scale)
        self.parameters_changed()

    def train_step(self):
        """forward, loss, backward, gradient mean over the data-parallel ranks, update."""
        from .. import distributed as dp
        self.forward()
        self.compute_loss(with_gradient=True)
        self.backward_with_allreduce()
        self._apply_gradients(1.0 / dp.world_size())
        return self.loss

    def capture_train_step(self, warmup=2):
        """The training step recorded once in a HIP graph and replayed: one launch from the host instead of
        some 250 ctypes calls.  The GPU time of the step is the same within a percent (17.0 ms against
        17.1 ms at batch 128: the eager launches already keep the queue full); what the replay removes is
        the host's share, for loops whose host also decodes and assembles batches.
        Single GPU and Adam only (the gradient all-reduce and the other optimizers stay on the eager
        path).  The first `warmup` steps run eagerly (they are ordinary training steps and allocate every
        buffer the recording needs).  Returns the callable that runs one step; the caller refills
        `network_input`, the ground truth and the loss weight in place as before.  The learning rate of
        every step is written to device memory by a fill kernel queued ahead of the replay (kernel arguments are
        frozen in a graph; the value rides in that kernel's own argument, so the host may run ahead).  Every node of the graph is a kernel: the library clears its accumulators with a kernel,
        because replays with memset nodes now and then ran ahead of their zero fill."""
        from .. import distributed as dp
        if dp.initialised() and dp.world_size() > 1:
            raise RuntimeError("capture_train_step: the data-parallel step is not captured; use train_step()")
        if getattr(self, "_optimizer", None) is None or self._optimizer.kind != 'AdamOptimizer':
            raise RuntimeError("capture_train_step: call add_optimizer() with Adam first")
        if self.trace is not None:
            raise RuntimeError("capture_train_step: tracing keeps per-step copies; switch it off")
        for _ in range(max(1, warmup)):      # every lazily allocated buffer exists before the capture
            self.train_step()
        self._lr_t_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._operands_stale = True          # the operand refresh after an update belongs to every step

        def body():
            self.forward()
            self.compute_loss(with_gradient=True)
            self.backward()
            ops.adam_step_dev(self.flat_params, self.flat_grads, self._adam_m, self._adam_v, self._lr_t_dev,
                              shadow=self._fc_shadow)
            self._operands_stale, self._fc_shadow_fresh = True, True

        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=self.device)
        torch.cuda.synchronize(self.device)
        with torch.cuda.graph(graph, stream=side):
            body()
        self._graph = graph

        def step():
            o = self._optimizer
            lr = o.learning_rate(self.global_step) if callable(o.learning_rate) else o.learning_rate
            # the value travels as an argument of a fill kernel queued ahead of the replay: no host buffer that a
            # host running several steps ahead of the GPU could overwrite before an earlier copy has read it
            self._lr_t_dev.fill_(ops.adam_lr_t(lr, self.global_step + 1))
            graph.replay()
            self.global_step += 1
            return self.loss

        self.apply_gradient_op = step
        return step

    @time_it(drain_gpu=True)
    def start_training(self, max_steps):
        """cnn_model.py:175-181."""
        for _ in range(max_steps):
            self.train_step()
        return self.loss
