"""VGG16 without batch normalisation on the RGB-D dataset: the reference's
slam/main/vgg16_noBN_rgbd.py:13-69.  The flow is that of slam_amd.main.vgg16_rgbd
(same configuration, provider, loss weights and saving schedule) with the encoder
class of slam_amd.network.cnn_model_noBN."""
from slam_amd.main import vgg16_rgbd
from slam_amd.network.cnn_model_noBN import VGG16Model


def main(argv=None):
    return vgg16_rgbd.main(argv, model_class=VGG16Model)


if __name__ == '__main__':
    main()
