"""TEST INFRASTRUCTURE - CPU restatement (PyTorch float32, differentiable) of the layer
graph the reference's GoogLeNet is made of (slam/network/google_net.py:14-215 on top of
kaffe.tensorflow.network.Network, an un-vendored and unpinned dependency, itself on
TensorFlow 0.x).  Layer semantics from the library's published helpers:
  conv      tf.nn.conv2d(padding='SAME') + bias (+ ReLU)
  max_pool  tf.nn.max_pool, avg_pool tf.nn.avg_pool (SAME: padding cells do not count)
  lrn       x / (bias + alpha * sum_{|j-c|<=r} x_j^2)^beta
  concat    channel axis;  fc: flatten NHWC, x.W + b (+ ReLU)
SAME padding: out = ceil(in / stride), total = max((out-1)*stride + k - in, 0), the
smaller half in front.  Parity unpinned: neither library is installed and the
reference holds no fixture for this path.

`run(spec, variables, x)` executes the plain graph description that
slam_amd.network.network.Network records (a list of dicts), so the checker shares no
code with the GPU executor.  With storage='bf16' every stored tensor and every GEMM
operand is rounded to bfloat16 (straight-through for gradients), as the GPU path stores
them.
"""
import torch
import torch.nn.functional as F


def _round_bf16(t):
    return t + (t.to(torch.bfloat16).to(torch.float32) - t).detach()


def _same_pad(size, k, stride):
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


def _pad_hw(x, k_h, k_w, stride, value=0.0):
    pt, pb = _same_pad(x.shape[2], k_h, stride)
    pl, pr = _same_pad(x.shape[3], k_w, stride)
    return F.pad(x, (pl, pr, pt, pb), value=value)


def run(spec, variables, x_nhwc, storage='bf16'):
    """variables: {'<layer>/weights': [k_h,k_w,c_in,c_o], '<layer>/biases': [c_o]} float32
    tensors (may require grad).  Returns {layer name: NHWC tensor} of every layer."""
    q = _round_bf16 if storage == 'bf16' else (lambda t: t)
    t = {'data': q(x_nhwc).permute(0, 3, 1, 2)}
    for e in spec:
        op = e['op']
        if op == 'concat':
            t[e['name']] = torch.cat([t[n] for n in e['inputs']], 1)
            continue
        x = t[e['input']]
        if op == 'conv':
            w = q(variables[e['name'] + '/weights']).permute(3, 2, 0, 1)       # HWIO -> OIHW
            y = F.conv2d(_pad_hw(x, e['k_h'], e['k_w'], e['stride']), w, variables[e['name'] + '/biases'],
                         stride=e['stride'])
            y = F.relu(y) if e['relu'] else y
        elif op == 'fc':
            flat = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)               # NHWC flatten
            w = q(variables[e['name'] + '/weights']).reshape(flat.shape[1], -1)
            y = flat @ w + variables[e['name'] + '/biases']
            y = (F.relu(y) if e['relu'] else y)[:, :, None, None]
        elif op == 'max_pool':
            xp = _pad_hw(x, e['k'], e['k'], e['stride'], float('-inf')) if e['same'] else x
            y = F.max_pool2d(xp, e['k'], e['stride'])
        elif op == 'avg_pool':
            if e['same']:
                ones = torch.ones_like(x[:1, :1])
                num = F.avg_pool2d(_pad_hw(x, e['k'], e['k'], e['stride']), e['k'], e['stride'])
                den = F.avg_pool2d(_pad_hw(ones, e['k'], e['k'], e['stride']), e['k'], e['stride'])
                y = num / den
            else:
                y = F.avg_pool2d(x, e['k'], e['stride'])
        elif op == 'lrn':
            r = e['radius']
            sq = F.pad(x * x, (0, 0, 0, 0, r, r))
            s = sum(sq[:, j:j + x.shape[1]] for j in range(2 * r + 1))
            y = x / (e['bias'] + e['alpha'] * s) ** e['beta']
        else:
            raise ValueError(op)
        t[e['name']] = q(y)
    return {k: v.permute(0, 2, 3, 1) for k, v in t.items()}
