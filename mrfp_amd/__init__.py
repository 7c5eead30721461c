"""MI355X-native MRFP+ training hot path."""
import os as _os

# Several HIP streams carry one training step (compute, weight gradients, the all-reduce side stream, RCCL's own); give
# them distinct hardware queues.  Only effective when this package is imported before the HIP runtime starts (bench.py
# sets it itself); see DESIGN.md section 5.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
