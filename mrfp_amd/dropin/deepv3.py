"""Drop-in for the reference's top-level `deepv3.py` (main.py:31 does `from deepv3 import *`)."""
from mrfp_amd.deepv3 import *  # noqa: F401,F403
from mrfp_amd.deepv3 import _AtrousSpatialPyramidPoolingModule  # noqa: F401
