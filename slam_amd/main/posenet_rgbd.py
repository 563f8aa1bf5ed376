"""slam/main/posenet_rgbd.py:15-61: GoogLeNet on the RGB-D sequences (4-channel input,
6-wide twist output, sequence_batch_itr(200, 20)); see slam_amd.main.posenet."""
from slam_amd.main import posenet
from slam_amd.network.google_net import GoogleNet


def main(argv=None):
    return posenet.run(argv, GoogleNet, rgbd=True, default_batch_size=20, default_base_dir='/usr/prakt/s085/google-net/')


if __name__ == '__main__':
    main()
