"""Drop-in for the reference's `metrics.py`."""
from mrfp_amd.metrics import *  # noqa: F401,F403
