"""Recurrent head of the CNN-LSTM trainer with the reference's class and method
names (slam/network/lstm_model.py:10-75): `layers` stacked LSTM cells of
`layer_size` units applied for one time step to the encoder output, a linear
output layer, the loss sum(((out - gt) W)^2) / batch and plain gradient descent
on a learning rate of 0.01 * 0.1^(global_step // 50) over every trainable
variable, the encoder's included.

As in the reference the recurrent state is a constant of zeros: the constructor
ignores its `init_state` argument (lstm_model.py:23), so the value the trainer
feeds for it has no effect; after build_graph() `init_state` is the tensor that
receives the cells' new state (lstm_model.py:29).

Initial values: TensorFlow 0.x defaults that the reference relies on - cell
weights uniform in +-sqrt(3 / (2 * layer_size)) (uniform_unit_scaling), cell
biases zero, output layer weights and biases standard normal.  The generator
differs from TensorFlow's, so individual values do; the distributions do not.
"""
import math

import torch

from . import encoder_ops as ops


class LSTMmodel:

    def __init__(self, model_input, layer_size, layers, output_dim, ground_truth, batch_size, init_state,
                 upstream=None, seed=0):
        """model_input: float32 GPU tensor [batch, layer_size] refreshed before every step -
        the encoder's network_output when `upstream` is that VGG16Model, whose parameters
        then train with the head's (main/lstm_rgbd.py:71-77).  ground_truth: float32 GPU
        tensor [batch, output_dim] the caller refills in place."""
        if not (model_input.is_cuda and model_input.dtype == torch.float32
                and tuple(model_input.shape) == (batch_size, layer_size)):
            raise ValueError("model_input must be a float32 GPU tensor [batch_size, layer_size]")
        self.model_input = model_input
        self.batch_size = int(batch_size)
        self.layer_size = int(layer_size)
        self.layers = int(layers)
        self.output_dim = int(output_dim)
        self.forget_bias = 1.0
        self.ground_truth = ground_truth
        self.upstream = upstream
        self.seed = seed
        self.device = model_input.device
        self.init_state = self.__init_state()
        self.global_step = 0

    def __init_state(self):
        return torch.zeros(self.batch_size, 2 * self.layer_size * self.layers, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ build
    def build_graph(self):
        n, L, D, B = self.layer_size, self.layers, self.output_dim, self.batch_size
        spec = (("lstm/MultiRNNCell/W", (L, 2 * n, 4 * n)), ("lstm/MultiRNNCell/B", (L, 4 * n)),
                ("lstm/fc_out/weights", (n, D)), ("lstm/fc_out/biases", (D,)))
        offs, total = {}, 0
        for name, shape in spec:
            offs[name] = (total, shape)
            total += math.prod(shape)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.flat_params = torch.zeros(total, **f32)
        self.flat_grads = torch.zeros(total, **f32)
        view = lambda flat, name: flat[offs[name][0]:offs[name][0] + math.prod(offs[name][1])].view(offs[name][1])
        self.params = {name: view(self.flat_params, name) for name, _ in spec}
        self.grads = {name: view(self.flat_grads, name) for name, _ in spec}
        g = torch.Generator().manual_seed(self.seed)
        bound = math.sqrt(3.0 / (2 * n))
        self.params["lstm/MultiRNNCell/W"].copy_((torch.rand(L, 2 * n, 4 * n, generator=g) * 2 - 1) * bound)
        self.params["lstm/fc_out/weights"].copy_(torch.randn(n, D, generator=g))
        self.params["lstm/fc_out/biases"].copy_(torch.randn(D, generator=g))

        self._state_in = self.init_state
        self._state_out = torch.zeros_like(self.init_state)
        self._gates = torch.empty(B, L, 4 * n, **f32)
        self._d_pre = torch.empty(B, L, 4 * n, **f32)
        self._d_x = torch.zeros(B, n, **f32)
        self.output_layer = torch.zeros(B, D, **f32)
        self.init_state = self._state_out       # the cells' new state (lstm_model.py:29)
        return self.output_layer

    def forward(self):
        if self.upstream is not None:
            self.upstream.forward()
        p = self.params
        ops.lstm_head_fwd(self.model_input, self._state_in, p["lstm/MultiRNNCell/W"], p["lstm/MultiRNNCell/B"],
                          p["lstm/fc_out/weights"], p["lstm/fc_out/biases"], self._gates, self._state_out,
                          self.output_layer, forget_bias=self.forget_bias)
        return self.output_layer

    # ------------------------------------------------------------------- loss
    def add_loss(self, loss_weight):
        """loss = sum(((output - ground_truth) . loss_weight)^2) / batch_size (lstm_model.py:52-55)."""
        self._loss_weight = loss_weight
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._d_out = torch.zeros_like(self.output_layer)
        return self.loss

    def compute_loss(self):
        ops.weighted_sq_loss(self.output_layer, self.ground_truth, self._loss_weight, 1.0 / self.batch_size,
                             self.loss, self._d_out)
        return self.loss

    def backward(self):
        """Gradients of `loss` into `grads`, and through the encoder when there is one."""
        p, g = self.params, self.grads
        ops.lstm_head_bwd(self._d_out, self.model_input, self._state_in, p["lstm/MultiRNNCell/W"],
                          p["lstm/fc_out/weights"], self._gates, self._state_out, self._d_pre, self._d_x,
                          g["lstm/MultiRNNCell/W"], g["lstm/MultiRNNCell/B"], g["lstm/fc_out/weights"],
                          g["lstm/fc_out/biases"])
        if self.upstream is not None:
            self.upstream.set_output_gradient(self._d_x)
            self.upstream.backward_with_allreduce()

    # -------------------------------------------------------------- optimizer
    def learning_rate(self, step):
        return 0.01 * 0.1 ** (step // 50)       # exponential_decay(0.01, step, 50, 0.1, staircase)

    def add_optimizer(self):
        """lstm_model.py:57-75.  Returns the callable that runs one training step."""
        self.global_step = 0
        self.apply_gradient_op = self.train_step
        return self.apply_gradient_op

    def train_step(self):
        from .. import distributed as dp
        self.forward()
        self.compute_loss()
        self.backward()
        world = dp.world_size()
        if dp.initialised():
            dp.finish([dp.start_bucket_allreduce(self.flat_grads, 0, self.flat_grads.numel())])
        lr = self.learning_rate(self.global_step)
        ops.sgd_step(self.flat_params, self.flat_grads, lr, grad_scale=1.0 / world)
        if self.upstream is not None:
            self.upstream.apply_sgd(lr, 1.0 / world)
        self.global_step += 1
        return self.loss

    # ------------------------------------------------------------ checkpoints
    def named_variables(self):
        out = dict(self.params)
        if self.upstream is not None:
            out.update(self.upstream.params)
        return out
