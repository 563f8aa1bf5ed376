"""GoogLeNet / PoseNet encoder with three regression heads on the MI355X kernels:
the reference's slam/network/google_net.py:8-266 (class and method names kept).

Graph (google_net.py:14-215): 7x7/2 convolution, 3x3/2 max-pool, LRN, 1x1 and 3x3
convolutions, LRN, max-pool, then nine inception modules (1x1 | 1x1-3x3 | 1x1-5x5 |
3x3 max-pool-1x1, concatenated) with max-pools after 3b and 4e; auxiliary heads on
the outputs of 4b and 4d (5x5/3 average pool, 1x1 convolution to 128, two ReLU fc
layers of 1024, linear output) and the main head on 5b (7x7 VALID average pool, two
linear fc layers of 1024, linear output).  The module widths below are the
reference's.
"""
import torch

from . import encoder_ops as ops
from .network import Network

# name: (1x1, 3x3 reduce, 3x3, 5x5 reduce, 5x5, pool projection)
INCEPTION = (
    ('3a', (64, 96, 128, 16, 32, 32)), ('3b', (128, 128, 192, 32, 96, 64)),
    ('4a', (192, 96, 208, 16, 48, 64)), ('4b', (160, 112, 224, 24, 64, 64)), ('4c', (128, 128, 256, 24, 64, 64)),
    ('4d', (112, 144, 288, 32, 64, 64)), ('4e', (256, 160, 320, 32, 128, 128)),
    ('5a', (256, 160, 320, 32, 128, 128)), ('5b', (384, 192, 384, 48, 128, 128)),
)
POOL_AFTER = {'3b': 'pool3_3x3_s2', '4e': 'pool4_3x3_s2'}
AUX_HEAD_AFTER = {'4b': 'output1', '4d': 'output2'}
HEAD_WEIGHTS = (0.3, 0.3, 1.0)          # google_net.py:223


class GoogleNet(Network):

    use_lrn = True

    def __init__(self, inputs, output_dim, trainable=True, seed=0):
        self.output_dim = output_dim
        super(GoogleNet, self).__init__(inputs, trainable, seed=seed)

    def _inception(self, tag, source, widths):
        c1, c3r, c3, c5r, c5, cp = widths
        p = 'inception_%s_' % tag
        self.feed(source).conv(1, 1, c1, 1, 1, name=p + '1x1')
        self.feed(source).conv(1, 1, c3r, 1, 1, name=p + '3x3_reduce').conv(3, 3, c3, 1, 1, name=p + '3x3')
        self.feed(source).conv(1, 1, c5r, 1, 1, name=p + '5x5_reduce').conv(5, 5, c5, 1, 1, name=p + '5x5')
        self.feed(source).max_pool(3, 3, 1, 1, name=p + 'pool').conv(1, 1, cp, 1, 1, name=p + 'pool_proj')
        self.feed(p + '1x1', p + '3x3', p + '5x5', p + 'pool_proj').concat(3, name=p + 'output')
        return p + 'output'

    def setup(self):
        stem = self.feed('data').conv(7, 7, 64, 2, 2, name='conv1_7x7_s2').max_pool(3, 3, 2, 2, name='pool1_3x3_s2')
        if self.use_lrn:
            stem = stem.lrn(2, 2e-05, 0.75, name='pool1_norm1')
        stem = stem.conv(1, 1, 64, 1, 1, name='conv2_3x3_reduce').conv(3, 3, 192, 1, 1, name='conv2_3x3')
        if self.use_lrn:
            stem = stem.lrn(2, 2e-05, 0.75, name='conv2_norm2')
        stem.max_pool(3, 3, 2, 2, name='pool2_3x3_s2')
        source = 'pool2_3x3_s2'
        for tag, widths in INCEPTION:
            source = self._inception(tag, source, widths)
            if tag in AUX_HEAD_AFTER:
                p = 'inception_%s_' % tag
                (self.feed(source).avg_pool(5, 5, 3, 3, name=p + 'avg_pool').conv(1, 1, 128, 1, 1, name=p + 'conv1x1')
                 .fc(1024, name=p + 'fc1').fc(1024, name=p + 'fc2')
                 .fc(self.output_dim, name=AUX_HEAD_AFTER[tag], relu=False))
            if tag in POOL_AFTER:
                self.feed(source).max_pool(3, 3, 2, 2, name=POOL_AFTER[tag])
                source = POOL_AFTER[tag]
        (self.feed(source).avg_pool(7, 7, 1, 1, padding='VALID', name='pool5_7x7_s1')
         .fc(1024, relu=False, name='inception_5b_fc1').fc(1024, relu=False, name='inception_5b_fc2')
         .fc(self.output_dim, relu=False, name='output3'))

    # --------------------------------------------------------------------- losses
    def _heads(self):
        return [self.layers[n] for n in ('output1', 'output2', 'output3')]

    def _add(self, groundtruth, kind):
        self._groundtruth, self._loss_kind = groundtruth, kind
        dev = self.device
        self.total_loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.head_losses = [torch.zeros(1, dtype=torch.float32, device=dev) for _ in range(3)]
        self._identity = torch.eye(self.output_dim, dtype=torch.float32, device=dev)
        return self.total_loss

    def add_loss(self, groundtruth):
        """0.3 loss1 + 0.3 loss2 + loss3 with get_loss per head (google_net.py:217-225)."""
        return self._add(groundtruth, 'posenet')

    def add_l2_loss(self, groundtruth):
        """The same weighting of tf.nn.l2_loss(output - groundtruth) per head
        (google_net.py:227-240; the add_n call there is written for this sum)."""
        return self._add(groundtruth, 'l2')

    def get_loss(self, output, groundtruth, loss=None, d_out=None):
        """sum((x^ - x)^2) + 1100 sum((q^ - q)^2) (google_net.py:242-250)."""
        loss = torch.zeros(1, dtype=torch.float32, device=output.device) if loss is None else loss
        ops.posenet_loss(output, groundtruth, loss, d_out, beta=1100.0)
        return loss

    def get_l2_loss(self, output, groundtruth, loss=None, d_out=None):
        loss = torch.zeros(1, dtype=torch.float32, device=output.device) if loss is None else loss
        ops.l2loss(output, groundtruth, self._identity, loss, d_out)
        return loss

    def compute_loss(self):
        fn = self.get_loss if self._loss_kind == 'posenet' else self.get_l2_loss
        self.total_loss.zero_()
        for head, weight, loss in zip(self._heads(), HEAD_WEIGHTS, self.head_losses):
            fn(head.out_f32, self._groundtruth, loss, head.d_out)
            head.d_out.mul_(weight)
            self.total_loss.add_(loss, alpha=weight)
        return self.total_loss

    # ------------------------------------------------------------------ optimizer
    def learning_rate(self, step):
        return 0.00001 * 0.1 ** (step // 10)        # exponential_decay(1e-5, step, 10, 0.1, staircase)

    def add_optimizer(self):
        """Adam on the decayed learning rate (google_net.py:253-266).  Returns the callable
        that runs one training step and gives back the total loss tensor."""
        self.global_step = 0
        self._adam_m = torch.zeros_like(self.flat_params)
        self._adam_v = torch.zeros_like(self.flat_params)
        self.apply_gradient_op = self.train_step
        return self.apply_gradient_op

    def train_step(self):
        from .. import distributed as dp
        self.forward()
        self.compute_loss()
        self.backward()
        if dp.initialised():
            dp.finish([dp.start_bucket_allreduce(self.flat_grads, 0, self.flat_grads.numel())])
        ops.adam_step(self.flat_params, self.flat_grads, self._adam_m, self._adam_v,
                      self.learning_rate(self.global_step), step=self.global_step + 1,
                      grad_scale=1.0 / dp.world_size())
        self.global_step += 1
        self._operands_stale = True
        return self.total_loss
