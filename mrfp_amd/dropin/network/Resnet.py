from mrfp_amd.network.Resnet import *  # noqa: F401,F403
