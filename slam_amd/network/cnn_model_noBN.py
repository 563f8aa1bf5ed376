"""The encoder without batch normalisation (slam/network/cnn_model_noBN.py:14-190):
same class name and methods as cnn_model.VGG16Model, every convolution followed by
bias + ReLU instead of batch-norm + ReLU."""
from . import cnn_model


class VGG16Model(cnn_model.VGG16Model):

    def __init__(self, batch_size, rgbd_input_batch, output_dim, normalization_epsilon=0.0, **kwargs):
        kwargs["batch_norm"] = False
        super().__init__(batch_size, rgbd_input_batch, output_dim, normalization_epsilon, **kwargs)
