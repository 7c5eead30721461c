from mrfp_amd.network.mynn import *  # noqa: F401,F403
