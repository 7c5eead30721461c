"""Drop-in for the reference's `config.py`."""
from mrfp_amd.config import *  # noqa: F401,F403
from mrfp_amd.config import cfg, assert_and_infer_cfg  # noqa: F401
