from mrfp_amd.network.wider_resnet import *  # noqa: F401,F403
from mrfp_amd.network.wider_resnet import IdentityResidualBlock, WiderResNetA2, bnrelu  # noqa: F401
