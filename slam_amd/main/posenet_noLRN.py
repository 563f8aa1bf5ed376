"""slam/main/posenet_noLRN.py: the PoseNet trainer on the GoogLeNet without LRN layers,
batch size 1 (see slam_amd.main.posenet)."""
from slam_amd.main import posenet
from slam_amd.network.google_net_noLRN import GoogleNet


def main(argv=None):
    return posenet.run(argv, GoogleNet, rgbd=False, default_batch_size=1, default_base_dir='/usr/prakt/s085/posenet/')


if __name__ == '__main__':
    main()
