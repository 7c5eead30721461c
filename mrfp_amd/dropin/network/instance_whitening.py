from mrfp_amd.network.instance_whitening import *  # noqa: F401,F403
