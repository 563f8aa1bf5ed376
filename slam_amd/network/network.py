"""Layer-graph builder and executor behind the GoogLeNet encoder: the role of
kaffe.tensorflow.network.Network, the un-vendored base class of the reference's
slam/network/google_net.py:2,8.  A subclass describes its graph in setup() with the
same chained calls (feed / conv / max_pool / avg_pool / lrn / concat / fc, same
argument orders and defaults); here the calls lay out buffers in HBM and record an
execution plan, and forward() / backward() run it as kernel launches through the
C ABI.  There is no autograd and no CPU fallback.

Semantics (restated from the library's published helpers; parity unpinned, the
library and TensorFlow are not available): conv = SAME convolution + bias + ReLU,
pools SAME unless padding='VALID', lrn = tf.nn.local_response_normalization, fc
flattens NHWC and computes x.W + b (+ ReLU).  Variables start from TensorFlow 0.x's
default initializer, uniform in +-sqrt(3 / fan_in) with fan_in the product of all
but the last dimension (1 for a bias vector).

Data layout: activations NHWC bfloat16; the members of a concat write straight
into channel slices of the concat's buffer; convolutions run as im2col + the
matrix-core GEMMs of the VGG encoder, so every filter is kept as a float32
[Kp][Np] matrix (K = kh*kw*c_in rows in (ky, kx, ci) order, both sides zero
padded to multiples of 64) inside one flat parameter buffer; gradients of
activations live in one arena that backward() zeroes once, and every layer adds
its input gradient into it (fan-out sums come for free).
"""
import math

import torch

from . import encoder_ops as ops
from . import layer_ops as lops

IN_PAD = 8      # the network input is widened to 8 channels (16-byte im2col chunks)


def _round_up(n, m):
    return (n + m - 1) // m * m


class _Node:
    """One tensor of the graph: logical shape [B, H, W, C] stored in `base` with row
    pitch `ld` at channel offset `off` (a concat member lives inside its concat)."""

    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)
        self.base = self.grad = self.out_f32 = self.d_out = None
        self.ld, self.off = shape[3], 0
        self.consumers = 0
        self.member_of = None

    @property
    def C(self):
        return self.shape[3]

    @property
    def rows(self):
        return self.shape[0] * self.shape[1] * self.shape[2]

    def whole(self):
        return self.ld == self.C and self.off == 0


class Network:

    def __init__(self, inputs, trainable=True, seed=0):
        """inputs: {'data': float32 GPU tensor [B, H, W, C]} that the caller refills in
        place before each step (the reference's placeholder)."""
        self.inputs = inputs
        self.trainable = trainable
        self.seed = seed
        self.layers = {}
        self.terminals = []
        self.spec = []          # plain description of the graph, in execution order
        data = inputs['data']
        if not (data.is_cuda and data.dtype == torch.float32 and data.dim() == 4):
            raise ValueError("inputs['data'] must be a float32 GPU tensor [B, H, W, C]")
        self.device = data.device
        B, H, W, C = data.shape
        if C > IN_PAD:
            raise ValueError("at most %d input channels" % IN_PAD)
        self.layers['data'] = _Node('data', (B, H, W, IN_PAD))
        self.setup()
        self._allocate()

    def setup(self):
        raise NotImplementedError('Must be subclassed.')

    # ------------------------------------------------------------ graph building
    def feed(self, *args):
        assert len(args) != 0
        self.terminals = []
        for name in args:
            if name not in self.layers:
                raise KeyError('Unknown layer name fed: %s' % name)
            self.terminals.append(self.layers[name])
        return self

    def get_output(self):
        return self.terminals[-1]

    def _single_input(self):
        assert len(self.terminals) == 1, "this layer takes one input"
        node = self.terminals[0]
        if node.member_of is not None:
            raise ValueError("%s is a concat member and cannot be read on its own" % node.name)
        node.consumers += 1
        return node

    def _register(self, entry, node):
        if node.name in self.layers:
            raise ValueError("duplicate layer name %s" % node.name)
        self.layers[node.name] = node
        self.spec.append(entry)
        self.terminals = [node]
        return self

    def conv(self, k_h, k_w, c_o, s_h, s_w, name, relu=True, padding='SAME', group=1, biased=True):
        assert padding == 'SAME' and group == 1 and biased and s_h == s_w, "only what the encoder uses"
        src = self._single_input()
        B, H, W, C = src.shape
        ho, wo = lops.conv_out_shape(H, W, k_h, k_w, s_h)
        return self._register(dict(op='conv', name=name, input=src.name, k_h=k_h, k_w=k_w, c_o=c_o, stride=s_h,
                                   relu=relu), _Node(name, (B, ho, wo, c_o)))

    def _pool(self, kind, k_h, k_w, s_h, s_w, name, padding):
        assert k_h == k_w and s_h == s_w and padding in ('SAME', 'VALID')
        src = self._single_input()
        B, H, W, C = src.shape
        ho, wo = lops.conv_out_shape(H, W, k_h, k_w, s_h, padding == 'SAME')
        return self._register(dict(op=kind, name=name, input=src.name, k=k_h, stride=s_h, same=padding == 'SAME'),
                              _Node(name, (B, ho, wo, C)))

    def max_pool(self, k_h, k_w, s_h, s_w, name, padding='SAME'):
        return self._pool('max_pool', k_h, k_w, s_h, s_w, name, padding)

    def avg_pool(self, k_h, k_w, s_h, s_w, name, padding='SAME'):
        return self._pool('avg_pool', k_h, k_w, s_h, s_w, name, padding)

    def lrn(self, radius, alpha, beta, name, bias=1.0):
        src = self._single_input()
        return self._register(dict(op='lrn', name=name, input=src.name, radius=radius, alpha=alpha, beta=beta, bias=bias),
                              _Node(name, src.shape))

    def concat(self, axis, name):
        assert axis == 3, "channel axis"
        members = list(self.terminals)
        B, H, W, _ = members[0].shape
        assert all(m.shape[:3] == (B, H, W) for m in members)
        node = _Node(name, (B, H, W, sum(m.C for m in members)))
        off = 0
        for m in members:
            if m.member_of is not None or m.consumers:
                raise ValueError("%s: a concat member may have no other consumer" % m.name)
            m.consumers += 1
            m.member_of, m.ld, m.off = node, node.C, off
            off += m.C
        return self._register(dict(op='concat', name=name, inputs=[m.name for m in members]), node)

    def fc(self, num_out, name, relu=True):
        src = self._single_input()
        B = src.shape[0]
        return self._register(dict(op='fc', name=name, input=src.name, c_o=num_out, relu=relu),
                              _Node(name, (B, 1, 1, num_out)))

    # ---------------------------------------------------------------- allocation
    def _allocate(self):
        dev = self.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        # activations: one buffer per tensor that is not a concat member
        grad_sizes = []
        for node in self.layers.values():
            if node.member_of is None:
                node.base = torch.zeros(node.shape, **bf)
                grad_sizes.append((node, node.base.numel()))
        self._grad_arena = torch.zeros(sum(_round_up(n, 128) for _, n in grad_sizes), **bf)
        pos = 0
        for node, n in grad_sizes:
            node.grad = self._grad_arena[pos:pos + n].view(node.shape)
            pos += _round_up(n, 128)
        for node in self.layers.values():
            if node.member_of is not None:
                node.base, node.grad = node.member_of.base, node.member_of.grad
            if node.consumers == 0 and node.name != 'data':       # graph output
                node.out_f32 = torch.zeros(node.shape[0], node.C, **f32)
                node.d_out = torch.zeros(node.shape[0], node.C, **f32)
        # parameters + scratch sizes
        self._plan, table, total = [], [], 0
        max_col = max_z = 1
        for e in self.spec:
            if e['op'] not in ('conv', 'fc'):
                self._plan.append((e, None))
                continue
            src, dst = self.layers[e['input']], self.layers[e['name']]
            if e['op'] == 'fc':
                B, H, W, C = src.shape
                geom = dict(in_view=(B, 1, 1, H * W * C), k_h=1, k_w=1, stride=1)
            else:
                geom = dict(in_view=src.shape, k_h=e['k_h'], k_w=e['k_w'], stride=e['stride'])
            c_in = geom['in_view'][3]
            if c_in % 8:
                raise ValueError("%s: %d input channels (must be a multiple of 8)" % (e['name'], c_in))
            geom['K'] = geom['k_h'] * geom['k_w'] * c_in
            geom['Kp'], geom['Np'] = _round_up(geom['K'], 64), _round_up(e['c_o'], 64)
            geom['M'] = dst.rows
            geom['direct'] = geom['k_h'] == 1 and geom['k_w'] == 1 and geom['stride'] == 1 and c_in % 64 == 0
            geom['w_off'], geom['b_off'] = total, total + geom['Kp'] * geom['Np']
            total = geom['b_off'] + geom['Np']
            max_col = max(max_col, geom['M'] * geom['Kp'])
            max_z = max(max_z, geom['M'] * geom['Np'])
            table.append((e, geom))
            self._plan.append((e, geom))
        self.flat_params = torch.zeros(total, **f32)
        self.flat_grads = torch.zeros(total, **f32)
        self._w_nk = torch.zeros(sum(g['Kp'] * g['Np'] for _, g in table), **bf)     # [Np][Kp] forward operands
        self._w_kn = torch.zeros_like(self._w_nk)                                      # [Kp][Np] data-gradient operands
        self._col = torch.zeros(max_col, **bf)
        self._dcol = torch.zeros(max_col, **bf)
        self._z = torch.zeros(max_z, **f32)
        self._dz = torch.zeros(max_z, **bf)
        self._x8 = self.layers['data'].base
        pos = 0
        gen = torch.Generator().manual_seed(self.seed)
        self.total_weights = 0
        for e, g in table:
            g['op_off'] = pos
            pos += g['Kp'] * g['Np']
            W = self._matrix(self.flat_params, g)
            fan_in = g['k_h'] * g['k_w'] * self._real_c_in(e, g)
            bound = math.sqrt(3.0 / fan_in)
            w = (torch.rand(g['k_h'], g['k_w'], self._real_c_in(e, g), e['c_o'], generator=gen) * 2 - 1) * bound
            self._filter_view(W, e, g).copy_(w)
            self.flat_params[g['b_off']:g['b_off'] + e['c_o']].copy_(
                (torch.rand(e['c_o'], generator=gen) * 2 - 1) * math.sqrt(3.0))
            self.total_weights += w.numel() + e['c_o']
        self._operands_stale = True

    def parameters_changed(self):
        """To be called after the parameters were written from outside (a restored checkpoint, a
        hand-made update): the bf16 operand copies are rebuilt before the next forward."""
        self._operands_stale = True

    def _real_c_in(self, e, g):
        """Channels of the layer's input that carry data (the padded network input has fewer)."""
        if e['input'] == 'data' and e['op'] == 'conv':
            return self.inputs['data'].shape[3]
        return g['in_view'][3]

    def _matrix(self, flat, g):
        return flat[g['w_off']:g['w_off'] + g['Kp'] * g['Np']].view(g['Kp'], g['Np'])

    def _filter_view(self, W, e, g):
        """[k_h, k_w, c_in, c_o] view of the data-carrying part of a [Kp][Np] filter matrix."""
        c_in = g['in_view'][3]
        return W[:g['K']].view(g['k_h'], g['k_w'], c_in, g['Np'])[:, :, :self._real_c_in(e, g), :e['c_o']]

    def named_variables(self):
        """{'<layer>/weights': [k_h, k_w, c_in, c_o] (fc: [1, 1, in, out]), '<layer>/biases': [c_o]}
        views of the flat parameter buffer, for weights files."""
        out = {}
        for e, g in self._plan:
            if g is None:
                continue
            out[e['name'] + '/weights'] = self._filter_view(self._matrix(self.flat_params, g), e, g)
            out[e['name'] + '/biases'] = self.flat_params[g['b_off']:g['b_off'] + e['c_o']]
        return out

    def named_gradients(self):
        out = {}
        for e, g in self._plan:
            if g is None:
                continue
            out[e['name'] + '/weights'] = self._filter_view(self._matrix(self.flat_grads, g), e, g)
            out[e['name'] + '/biases'] = self.flat_grads[g['b_off']:g['b_off'] + e['c_o']]
        return out

    # ------------------------------------------------------------------- forward
    def _refresh_operands(self):
        for e, g in self._plan:
            if g is None:
                continue
            W = self._matrix(self.flat_params, g)
            n = g['Kp'] * g['Np']
            ops.transpose_cast(W, self._w_nk[g['op_off']:g['op_off'] + n].view(g['Np'], g['Kp']))
            ops.cast_bf16(W, self._w_kn[g['op_off']:g['op_off'] + n].view(g['Kp'], g['Np']))
        self._operands_stale = False

    def _columns(self, src, g):
        """The [M][Kp] im2col matrix of a layer's input (the input itself for 1x1 / fc layers
        whose width is already a multiple of 64)."""
        assert src.whole(), "%s is read as a whole tensor" % src.name
        x = src.base.view(g['in_view'])
        if g['direct']:
            return x.view(g['M'], g['Kp'])
        col = self._col[:g['M'] * g['Kp']].view(g['M'], g['Kp'])
        lops.im2col(x, col, g['k_h'], g['k_w'], g['stride'])
        return col

    def forward(self):
        if self._operands_stale:
            self._refresh_operands()
        lops.pad_cast(self.inputs['data'], self._x8)
        for e, g in self._plan:
            op = e['op']
            if op == 'concat':
                continue                                   # the members wrote into it
            src, dst = self.layers[e['input']], self.layers[e['name']]
            if op in ('conv', 'fc'):
                col = self._columns(src, g)
                n = g['Kp'] * g['Np']
                z = self._z[:g['M'] * g['Np']].view(g['M'], g['Np'])
                ops.gemm_nt(col, self._w_nk[g['op_off']:g['op_off'] + n].view(g['Np'], g['Kp']), out_f32=z)
                bias = self.flat_params[g['b_off']:g['b_off'] + g['Np']]
                lops.bias_act_slice(z, bias, e['c_o'], y_bf16=dst.base, ld=dst.ld, off=dst.off, relu=e['relu'])
                if dst.out_f32 is not None:
                    lops.bias_act_slice(z, bias, e['c_o'], y_f32=dst.out_f32, relu=e['relu'])
            elif op in ('max_pool', 'avg_pool'):
                assert src.whole() and dst.whole()
                lops.pool(src.base, dst.base, e['k'], e['stride'], e['same'], op == 'max_pool')
            elif op == 'lrn':
                assert src.whole() and dst.whole()
                lops.lrn(src.base, dst.base, e['radius'], e['alpha'], e['beta'], e['bias'])
            else:
                raise ValueError(op)

    # ------------------------------------------------------------------ backward
    def backward(self):
        """Gradients of the loss whose d loss / d output sits in every output node's `d_out`
        into flat_grads."""
        self._grad_arena.zero_()
        for e, g in reversed(self._plan):
            op = e['op']
            if op == 'concat':
                continue
            src, dst = self.layers[e['input']], self.layers[e['name']]
            if op in ('conv', 'fc'):
                n = g['Kp'] * g['Np']
                dz = self._dz[:g['M'] * g['Np']].view(g['M'], g['Np'])
                gb = self.flat_grads[g['b_off']:g['b_off'] + g['Np']]
                if dst.d_out is not None:
                    lops.bias_act_slice_bwd(dst.d_out, dst.base, dz, gb, e['c_o'], relu=e['relu'])
                else:
                    lops.bias_act_slice_bwd(dst.grad, dst.base, dz, gb, e['c_o'], ld=dst.ld, off=dst.off, relu=e['relu'])
                col = self._columns(src, g)
                ops.wgrad(col, dz, self._matrix(self.flat_grads, g))
                if src.name != 'data':
                    dcol = self._dcol[:g['M'] * g['Kp']].view(g['M'], g['Kp'])
                    ops.gemm_nt(dz, self._w_kn[g['op_off']:g['op_off'] + n].view(g['Kp'], g['Np']), out_bf16=dcol)
                    lops.col2im_add(dcol, src.grad.view(g['in_view']), g['k_h'], g['k_w'], g['stride'])
            elif op in ('max_pool', 'avg_pool'):
                lops.pool_bwd_add(src.base, dst.grad, src.grad, e['k'], e['stride'], e['same'], op == 'max_pool')
            elif op == 'lrn':
                lops.lrn_bwd_add(src.base, dst.grad, src.grad, e['radius'], e['alpha'], e['beta'], e['bias'])

    def outputs(self):
        """{name: float32 [B, width]} of the graph's output layers."""
        return {n.name: n.out_f32 for n in self.layers.values() if n.out_f32 is not None}
