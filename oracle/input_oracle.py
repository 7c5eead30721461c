"""CPU oracle of the training input pipeline's geometric part -- TEST INFRASTRUCTURE ONLY (imported by tests/ only).

Two layers:
* `transform_pil`: the reference's own sequence of PIL calls for one sample (main.py:409-419 transform_tr), restated call
  for call -- RandomHorizontalFlip (dataloaders.py:139-150), RandomSizeAndCrop (398-435: img.resize((w, h), BICUBIC),
  mask.resize((w, h), NEAREST)), RandomCrop (257-337: ImageOps.expand borders, crop), Resize (467-482: identity here, PIL
  returns a copy when the size is unchanged), ToTensor (118-136: float32, NO division by 255) -- with the random draws
  passed in, RandomGaussianBlur (168-177) when its gate fired.  ColorJitter (596-660) is NOT part of the GPU path (DESIGN.md
  section 8).
* `resample_tables` / `resample_u8` / `nearest_table`: the arithmetic INSIDE those PIL calls, restated from the published
  algorithm of the third-party dependency Pillow (pinned here: 12.2.0; src/libImaging/Resample.c precompute_coeffs,
  normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; Geometry.c ImagingScaleAffine for NEAREST;
  BoxBlur.c _gaussian_blur_radius, ImagingLineBoxBlur8 for GaussianBlur).
  Pinned against PIL itself in tests/test_input_cpu.py (bit-exact on every case).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x: float) -> float:
    x = -x if x < 0 else x
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x: float) -> float:
    a = -0.5
    x = -x if x < 0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


def resample_tables(in_size: int, out_size: int, filt: str):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc: bounds [out,2] = (first source index, count), coefs [out,ksize]
    int32 in 22-bit fixed point.  Python floats are C doubles and the operations run in Pillow's order."""
    f, sup = FILTERS[filt]
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = sup * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coefs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [f((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            coefs[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coefs


def _resample_axis(img: np.ndarray, out_size: int, filt: str, axis: int) -> np.ndarray:
    bounds, coefs = resample_tables(img.shape[axis], out_size, filt)
    a = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx in range(out_size):
        lo, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(coefs[xx, :n].astype(np.int64), a[lo:lo + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resample_u8(img: np.ndarray, w: int, h: int, filt: str) -> np.ndarray:
    """img.resize((w, h), filt) of an 8-bit [H,W,C] image: horizontal pass, 8-bit intermediate, vertical pass."""
    out = img
    if w != img.shape[1]:
        out = _resample_axis(out, w, filt, 1)
    if h != img.shape[0]:
        out = _resample_axis(out, h, filt, 0)
    return out


def nearest_table(in_size: int, out_size: int) -> np.ndarray:
    """Source index of every destination pixel of resize(NEAREST): ImagingScaleAffine accumulates xo += a in double."""
    a = float(in_size) / out_size
    xo = 0.0 + a * 0.5
    tab = np.zeros(out_size, np.int32)
    for x in range(out_size):
        tab[x] = -1 if xo < 0.0 else int(xo)
        xo += a
    return tab


def gaussian_box_radius(radius: float, passes: int = 3) -> np.float32:
    """Pillow BoxBlur.c _gaussian_blur_radius: the box radius whose `passes` repetitions approximate the Gaussian (C float)."""
    f32 = np.float32
    r = f32(radius)
    sigma2 = f32(f32(r * r) / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32(f32(f32(2) * l + f32(1)) * f32(f32(l * f32(l + f32(1))) - f32(f32(3) * sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - f32(f32(l + f32(1)) * f32(l + f32(1))))))
    return f32(l + a)


def gaussian_blur_u8(img: np.ndarray, radius: float) -> np.ndarray:
    """img.filter(ImageFilter.GaussianBlur(radius)) of an 8-bit [H,W,C] image for a box radius below 1 (ImagingBoxBlur:
    three passes of ImagingLineBoxBlur8 along x, then three along y; every pass rounds to 8 bits, edges replicated)."""
    f32 = np.float32
    fr = gaussian_box_radius(radius)
    assert int(fr) == 0, "restated for box radii < 1 only (radius = random.random() in the reference)"
    ww = int(f32(f32(1 << 24) / f32(fr * f32(2) + f32(1))))
    fw = ((1 << 24) - ww) // 2
    out = img
    for axis in (1, 1, 1, 0, 0, 0):
        x = np.moveaxis(out, axis, 0).astype(np.int64)
        left, right = np.concatenate([x[:1], x[:-1]]), np.concatenate([x[1:], x[-1:]])
        y = ((x * ww + (left + right) * fw + (1 << 23)) >> 24).astype(np.uint8)
        out = np.moveaxis(y, 0, axis)
    return out


def transform_pil(img, mask, *, flip: bool, scaled_size, pad, crop_xy, crop_size: int, ignore_index: int = 255, blur=None):
    """The reference's PIL calls for one sample with the draws given: img / mask are PIL images ('RGB' / 'L').
    -> (float32 [3,Hc,Wc] in 0..255, float32 [Hc,Wc]) as dataloaders.py ToTensor returns them."""
    from PIL import Image, ImageOps
    if flip:                                                             # dataloaders.py:145-147
        img, mask = img.transpose(Image.FLIP_LEFT_RIGHT), mask.transpose(Image.FLIP_LEFT_RIGHT)
    w, h = scaled_size
    img, mask = img.resize((w, h), Image.BICUBIC), mask.resize((w, h), Image.NEAREST)      # :427
    pad_w, pad_h = pad
    if not (w == crop_size and h == crop_size):                          # RandomCrop.__call__ :283-337
        if pad_h or pad_w:
            border = (pad_w, pad_h, pad_w, pad_h)
            img = ImageOps.expand(img, border=border, fill=(0, 0, 0))
            mask = ImageOps.expand(mask, border=border, fill=ignore_index)
        x1, y1 = crop_xy
        img = img.crop((x1, y1, x1 + crop_size, y1 + crop_size))
        mask = mask.crop((x1, y1, x1 + crop_size, y1 + crop_size))
    img, mask = img.resize((crop_size, crop_size), Image.BILINEAR), mask.resize((crop_size, crop_size), Image.NEAREST)  # :479-480
    if blur is not None:                                                 # RandomGaussianBlur :172-174
        from PIL import ImageFilter
        img = img.filter(ImageFilter.GaussianBlur(radius=blur))
    im = np.array(img).astype(np.float32).transpose((2, 0, 1))          # ToTensor :128-133
    return im, np.array(mask).astype(np.float32)
