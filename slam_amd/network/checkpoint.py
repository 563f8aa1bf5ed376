"""Weights files of the trainers: the role of the reference's tf.train.Saver
(slam/main/vgg16_rgbd.py:45,61-63; slam/main/lstm_rgbd.py:48,65,101-103).  The
TensorFlow checkpoint format is not reproduced: a file is a dict of named float32
tensors written with torch.save and read back with weights_only=True."""
import os

import torch


def optimizer_slots(model):
    """The optimizer's per-parameter state of a VGG16Model as named tensors (tf.train.Saver stores the
    slot variables with the weights): Adam's two moments or RMSProp's mean square, flat buffers in the
    layout of model.flat_params.  Pass the result to save() / restore() merged with model.params to make
    a run resumable."""
    slots = {}
    for name, attr in (("optimizer/adam_m", "_adam_m"), ("optimizer/adam_v", "_adam_v"), ("optimizer/rms", "_rms")):
        t = getattr(model, attr, None)
        if t is not None:
            slots[name] = t
    return slots


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
