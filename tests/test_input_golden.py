"""Input transform against COMMITTED outputs of the reference's PIL calls (tests/golden/input_pipeline.npz, written by
tests/golden/make_golden_input.py in the build container): needs no Pillow at test time."""
import numpy as np
import pytest
import torch

from oracle import input_oracle as io

DEV = "cuda:0"


def test_numpy_pipeline_equals_golden_pil_outputs():
    """oracle/input_oracle.py::transform_numpy (restated arithmetic only) against tests/golden/input_pipeline.npz, the outputs
    of the reference's PIL calls recorded by tests/golden/make_golden_input.py -- needs no PIL."""
    import os
    import sys
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gdir)
    import make_golden_input as mg
    G = np.load(os.path.join(gdir, "input_pipeline.npz"))
    for i, d in enumerate(mg.DRAWS):
        im, lb = io.transform_numpy(G["img"], G["lab"], flip=d["flip"], scaled_size=d["scaled"], pad=d["pad"], crop_xy=d["crop"],
                                    crop_size=mg.CROP, blur=d["blur"], jitter=d["jitter"])
        assert np.array_equal(im, G["img_%d" % i]) and np.array_equal(lb, G["lab_%d" % i]), i


@pytest.mark.gpu
def test_train_transform_equals_golden_pil_outputs():
    """device output against the committed PIL outputs (tests/golden/input_pipeline.npz): independent of the local Pillow."""
    import os
    import sys
    from mrfp_amd.input_pipeline import Draw, TrainTransform
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gdir)
    import make_golden_input as mg
    G = np.load(os.path.join(gdir, "input_pipeline.npz"))
    tt = TrainTransform(mg.CROP, 0.5, 2.0, 255)
    xi, xl = torch.from_numpy(G["img"]).to(DEV), torch.from_numpy(G["lab"]).to(DEV)
    for i, d in enumerate(mg.DRAWS):
        im, lb = tt(xi, xl, Draw(d["flip"], d["jitter"], d["scaled"], d["pad"], d["crop"], d["blur"]))
        assert np.array_equal(im.cpu().numpy(), G["img_%d" % i].astype(np.float32)), i
        assert np.array_equal(lb.cpu().numpy(), G["lab_%d" % i].astype(np.int64)), i
