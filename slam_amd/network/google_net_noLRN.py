"""The GoogLeNet encoder without the two local response normalisations
(slam/network/google_net_noLRN.py): same class name and methods."""
from . import google_net


class GoogleNet(google_net.GoogleNet):
    use_lrn = False
