"""Input pipeline on the GPU (csrc/input.hip, mrfp_amd/input_pipeline.py) against the reference's PIL calls
(oracle/input_oracle.py::transform_pil): byte-exact image and label for flipped / unflipped, colour-jittered, up- and
down-scaled, padded and unpadded, blurred draws."""
import random

import numpy as np
import pytest
import torch

from oracle import input_oracle as io

pytestmark = pytest.mark.gpu
Image = pytest.importorskip("PIL.Image")
DEV = "cuda:0"


@pytest.mark.parametrize("H,W,crop", [(96, 128, 64), (60, 90, 96), (128, 256, 128), (75, 75, 75)])
def test_train_transform_equals_pil(H, W, crop):
    from mrfp_amd.input_pipeline import Draw, TrainTransform
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    lab = rng.integers(0, 19, (H, W), dtype=np.uint8)
    lab[rng.random((H, W)) < 0.03] = 255
    tt = TrainTransform(crop, 0.5, 2.0, 255)
    r, nr = random.Random(7), np.random.RandomState(7)
    draws = [tt.draw(W, H, r, nr) for _ in range(12)]
    assert any(d.jitter for d in draws) and any(d.blur is not None for d in draws) and any(d.flip for d in draws)
    draws.append(Draw(True, [("hue", -0.3), ("contrast", 1.2), ("brightness", 1.5), ("saturation", 0.8)], (W, H), ((crop - W) // 2 + 1 if crop > W else 0, (crop - H) // 2 + 1 if crop > H else 0),
                      (0, 0), 0.37))                                  # scale exactly 1: mirrored copy only; blurred
    xi, xl = torch.from_numpy(img).to(DEV), torch.from_numpy(lab).to(DEV)
    for d in draws:
        want_im, want_lab = io.transform_pil(Image.fromarray(img), Image.fromarray(lab), flip=d.flip, scaled_size=d.scaled,
                                             pad=d.pad, crop_xy=d.crop, crop_size=crop, blur=d.blur, jitter=d.jitter)
        got_im, got_lab = tt(xi, xl, d)
        assert np.array_equal(got_im.cpu().numpy(), want_im), d
        assert np.array_equal(got_lab.cpu().numpy(), want_lab.astype(np.int64)), d


def test_refuses_cpu_tensors():
    from mrfp_amd import _lib
    from mrfp_amd.input_pipeline import Draw, TrainTransform
    tt = TrainTransform(32)
    with pytest.raises(_lib.MrfpHipError):
        tt(torch.zeros(8, 8, 3, dtype=torch.uint8), torch.zeros(8, 8, dtype=torch.uint8), Draw(False, None, (8, 8), (13, 13), (0, 0), None))

