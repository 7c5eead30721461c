"""Golden vectors of the input transform: outputs of the reference's own sequence of PIL calls (dataloaders.py transforms as
main.py:409-419 composes them; Pillow 12.2.0 in the build container) for a small seeded image and fixed draws.

    python tests/golden/make_golden_input.py        ->  tests/golden/input_pipeline.npz  (inputs + expected outputs, ~20 KB)

The fixture keeps tests/test_input_*.py meaningful on a machine whose Pillow differs or is absent."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import input_oracle as io  # noqa: E402

DRAWS = [
    dict(flip=False, jitter=None, scaled=(58, 43), pad=(0, 0), crop=(7, 3), blur=None),
    dict(flip=True, jitter=[("hue", -0.21), ("contrast", 1.13), ("brightness", 0.62), ("saturation", 0.9)], scaled=(30, 22),
         pad=(2, 6), crop=(1, 0), blur=0.61),
    dict(flip=True, jitter=[("saturation", 1.2), ("brightness", 1.5), ("hue", 0.3), ("contrast", 0.8)], scaled=(40, 30),
         pad=(0, 2), crop=(5, 1), blur=None),
    dict(flip=False, jitter=None, scaled=(77, 58), pad=(0, 0), crop=(20, 11), blur=0.05),
]
CROP = 32


def main():
    from PIL import Image
    rng = np.random.default_rng(2024)
    img = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    img[:4] = img[:4, :, :1]                   # grey pixels
    lab = rng.integers(0, 19, (30, 40), dtype=np.uint8)
    lab[rng.random((30, 40)) < 0.05] = 255
    out = {"img": img, "lab": lab}
    for i, d in enumerate(DRAWS):
        im, lb = io.transform_pil(Image.fromarray(img), Image.fromarray(lab), flip=d["flip"], scaled_size=d["scaled"], pad=d["pad"],
                                  crop_xy=d["crop"], crop_size=CROP, blur=d["blur"], jitter=d["jitter"])
        out["img_%d" % i] = im.astype(np.uint8)          # exact: the values are integers 0..255
        out["lab_%d" % i] = lb.astype(np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "input_pipeline.npz"), **out)
    print("written", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
