from mrfp_amd.network.cov_settings import *  # noqa: F401,F403
from mrfp_amd.network.cov_settings import CovMatrix_ISW, CovMatrix_IRW, make_cov_index_matrix  # noqa: F401
