"""Input pipeline, CPU part: the restatement of Pillow's resampler in oracle/input_oracle.py
is pinned against PIL itself (bit-exact), and the host-side tables / random draws of mrfp_amd/input_pipeline.py agree
with it."""
import random

import numpy as np
import pytest

from oracle import input_oracle as io

Image = pytest.importorskip("PIL.Image")

CASES = [(37, 53, 74, 80), (64, 96, 40, 61), (50, 50, 50, 77), (120, 200, 173, 289), (101, 77, 55, 39), (33, 41, 16, 20),
         (48, 64, 48, 64), (30, 45, 131, 17)]


@pytest.mark.parametrize("H,W,oh,ow", CASES)
def test_resampler_restatement_equals_pil(H, W, oh, ow):
    rng = np.random.default_rng(H * 1000 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    lab = rng.integers(0, 20, (H, W), dtype=np.uint8)
    for filt, pf in (("bicubic", Image.BICUBIC), ("bilinear", Image.BILINEAR)):
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), pf))
        assert np.array_equal(io.resample_u8(img, ow, oh, filt), ref), filt
    refl = np.asarray(Image.fromarray(lab).resize((ow, oh), Image.NEAREST))
    assert np.array_equal(lab[io.nearest_table(H, oh)][:, io.nearest_table(W, ow)], refl)


def test_pipeline_tables_equal_oracle_tables():
    from mrfp_amd import input_pipeline as ip
    rng = random.Random(3)
    pairs = [(53, 80), (96, 61), (200, 289), (64, 64), (45, 17), (2048, 1311), (1024, 2047), (1914, 957)]
    pairs += [(rng.randint(8, 700), rng.randint(8, 700)) for _ in range(40)]
    for a, b in pairs:
        bo, ko = io.resample_tables(a, b, "bicubic")
        bp, kp = ip._bicubic_tables(a, b)
        assert np.array_equal(bo, bp) and np.array_equal(ko, kp)
        assert np.array_equal(io.nearest_table(a, b), ip._nearest_table(a, b))


def test_draw_follows_the_reference_order():
    """flip gate, jitter gate, scale, x1, y1, blur gate (+ radius): dataloaders.py:145, 655, 421, 327-331, 172-174."""
    from mrfp_amd.input_pipeline import TrainTransform
    tt = TrainTransform(64, 0.5, 2.0, 255)
    for seed in range(20):
        r1, r2 = random.Random(seed), random.Random(seed)
        n1, n2 = np.random.RandomState(seed), np.random.RandomState(seed)
        d = tt.draw(100, 80, r1, n1)
        flip = r2.random() < 0.5
        jit = None
        if r2.random() < 0.5:                              # ColorJitter.__call__ gate, then get_params (:622-643)
            jit = [("brightness", n2.uniform(0.5, 1.5)), ("contrast", n2.uniform(0.8, 1.2)), ("saturation", n2.uniform(0.8, 1.2)),
                   ("hue", n2.uniform(-0.3, 0.3))]
            n2.shuffle(jit)
        s = 1.0 * r2.uniform(0.5, 2.0)
        w, h = int(100 * s), int(80 * s)
        pad_h = (64 - h) // 2 + 1 if 64 > h else 0
        pad_w = (64 - w) // 2 + 1 if 64 > w else 0
        W2, H2 = w + 2 * pad_w, h + 2 * pad_h
        x1 = 0 if W2 == 64 else r2.randint(0, W2 - 64)
        y1 = 0 if H2 == 64 else r2.randint(0, H2 - 64)
        blur = r2.random() if r2.random() < 0.5 else None
        assert (d.flip, d.jitter, d.scaled, d.pad, d.crop, d.blur) == (flip, jit, (w, h), (pad_w, pad_h), (x1, y1), blur)
        assert r1.random() == r2.random() and n1.uniform() == n2.uniform()      # all streams are at the same position


def test_transform_pil_shapes_and_padding():
    rng = np.random.default_rng(1)
    img = Image.fromarray(rng.integers(0, 256, (40, 60, 3), dtype=np.uint8))
    mask = Image.fromarray(rng.integers(0, 19, (40, 60), dtype=np.uint8))
    im, lab = io.transform_pil(img, mask, flip=True, scaled_size=(45, 30), pad=(10, 18), crop_xy=(1, 2), crop_size=64)
    assert im.shape == (3, 64, 64) and lab.shape == (64, 64) and im.dtype == np.float32
    assert (lab == 255).any() and (im[:, lab == 255] == 0).all()        # the border: ignore label, black pixels


def test_gaussian_blur_restatement_equals_pil():
    from PIL import ImageFilter
    from mrfp_amd import input_pipeline as ip
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (29, 37, 3), dtype=np.uint8)
    r = random.Random(2)
    for radius in [r.random() for _ in range(12)] + [1e-3, 0.999]:
        ref = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=radius)))
        assert np.array_equal(io.gaussian_blur_u8(img, radius), ref), radius
        f32 = np.float32
        fr = io.gaussian_box_radius(radius)
        ww = int(f32(f32(1 << 24) / f32(fr * f32(2) + f32(1))))
        assert ip._blur_weights(radius) == (ww, ((1 << 24) - ww) // 2)


def test_color_jitter_restatement_equals_pil():
    """Blend.c / Convert.c restated: every adjust_* call of dataloaders.py:491-594 against PIL (the two colour-space
    conversions also on every one of the 2^24 triples)."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (45, 61, 3), dtype=np.uint8)
    img[:6] = img[:6, :, :1]                               # grey pixels: the s == 0 branches
    pim = Image.fromarray(img)
    r = random.Random(4)
    for op, lo, hi in (("brightness", 0.5, 1.5), ("contrast", 0.8, 1.2), ("saturation", 0.8, 1.2), ("hue", -0.3, 0.3)):
        for f in [lo, hi, 1.0 if op != "hue" else 0.0] + [r.uniform(lo, hi) for _ in range(6)]:
            assert np.array_equal(io.jitter_u8(img, op, f), np.asarray(io.jitter_pil(pim, op, f))), (op, f)
    a = np.arange(1 << 24, dtype=np.uint32)
    tri = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], -1).astype(np.uint8).reshape(4096, 4096, 3)
    assert np.array_equal(io.rgb2hsv(tri), np.asarray(Image.fromarray(tri).convert("HSV")))
    assert np.array_equal(io.hsv2rgb(tri), np.asarray(Image.fromarray(tri, "HSV").convert("RGB")))

