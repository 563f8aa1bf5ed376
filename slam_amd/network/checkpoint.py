"""Weights files of the trainers: the role of the reference's tf.train.Saver
(slam/main/vgg16_rgbd.py:45,61-63; slam/main/lstm_rgbd.py:48,65,101-103).  The
TensorFlow checkpoint format is not reproduced: a file is a dict of named float32
tensors written with torch.save and read back with weights_only=True."""
import os

import torch


def save(filename, named_variables, global_step=None):
    os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
    payload = {k: v.detach().cpu().clone() for k, v in named_variables.items()}
    if global_step is not None:
        payload["global_step"] = torch.tensor(int(global_step))
    tmp = filename + ".tmp"
    torch.save(payload, tmp)
    os.replace(tmp, filename)


def restore(filename, named_variables, models=()):
    """Copies every saved tensor into the variable of the same name; a variable without
    a saved value, or a shape mismatch, is an error (as Saver.restore).  `models`: the
    networks that own the variables; each is told that its parameters changed, so that
    derived copies (bf16 operands) are rebuilt before its next forward.  Returns the saved
    global step or None."""
    payload = torch.load(filename, map_location="cpu", weights_only=True)
    for name, var in named_variables.items():
        if name not in payload:
            raise KeyError("%s: no saved value for variable %r" % (filename, name))
        if tuple(payload[name].shape) != tuple(var.shape):
            raise ValueError("%s: variable %r has shape %s, saved %s" %
                             (filename, name, tuple(var.shape), tuple(payload[name].shape)))
        var.copy_(payload[name])
    for m in models:
        m.parameters_changed()
    step = payload.get("global_step")
    return None if step is None else int(step)
