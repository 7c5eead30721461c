"""CPU oracle of the training input pipeline (transform_tr) -- TEST INFRASTRUCTURE ONLY (imported by tests/ only).

Two layers:
* `transform_pil`: the reference's own sequence of PIL calls for one sample (main.py:409-419 transform_tr), restated call
  for call -- RandomHorizontalFlip (dataloaders.py:139-150), RandomSizeAndCrop (398-435: img.resize((w, h), BICUBIC),
  mask.resize((w, h), NEAREST)), RandomCrop (257-337: ImageOps.expand borders, crop), Resize (467-482: identity here, PIL
  returns a copy when the size is unchanged), ToTensor (118-136: float32, NO division by 255) -- with the random draws
  passed in: ColorJitter (596-660: the drawn list of (op, factor)), RandomGaussianBlur (168-177) when its gate fired.
* `resample_tables` / `resample_u8` / `nearest_table`: the arithmetic INSIDE those PIL calls, restated from the published
  algorithm of the third-party dependency Pillow (pinned here: 12.2.0; src/libImaging/Resample.c precompute_coeffs,
  normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc; Geometry.c ImagingScaleAffine for NEAREST;
  BoxBlur.c _gaussian_blur_radius, ImagingLineBoxBlur8 for GaussianBlur; Blend.c ImagingBlend, Convert.c rgb2l /
  rgb2hsv / hsv2rgb for ColorJitter).
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


# ---- ColorJitter (dataloaders.py:491-660) --------------------------------------------------------------------------
def pil_blend(d: np.ndarray, p: np.ndarray, alpha: float) -> np.ndarray:
    """Image.blend(degenerate, image, alpha) per byte (Pillow Blend.c, C float arithmetic, clipped)."""
    a = np.float32(alpha)
    t = d.astype(np.float32) + a * (p.astype(np.int32) - d.astype(np.int32)).astype(np.float32)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def pil_l(img: np.ndarray) -> np.ndarray:
    r, g, b = [img[..., i].astype(np.int64) for i in range(3)]
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)       # Convert.c rgb2l


def rgb2hsv(img: np.ndarray) -> np.ndarray:
    """Pillow Convert.c rgb2hsv_row (float variables, double constants as in the C source)."""
    f32 = np.float32
    r, g, b = [img[..., i] for i in range(3)]
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    gray = maxc == minc
    cr = (maxc.astype(np.int32) - minc.astype(np.int32)).astype(np.float32)
    crs = np.where(gray, f32(1), cr)
    s = cr / np.where(gray, f32(1), maxc.astype(np.float32))
    rc, gc, bc = [(maxc.astype(np.int32) - c).astype(np.float32) / crs for c in (r, g, b)]
    h = np.where(r == maxc, (bc - gc).astype(np.float64),
                 np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc.astype(np.float64),
                          4.0 + gc.astype(np.float64) - rc.astype(np.float64))).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv2rgb(hsv: np.ndarray) -> np.ndarray:
    """Pillow Convert.c hsv2rgb (following colorsys.py; float arithmetic, round())."""
    f32 = np.float32
    h, s, v = hsv[..., 0].astype(np.float64), hsv[..., 1], hsv[..., 2]
    fs = s.astype(np.float32) / f32(255.0)
    hh = h * 6.0 / 255.0
    i = np.floor(hh).astype(np.int64)
    f = (hh - i).astype(np.float32)
    vf = v.astype(np.float32)

    def rnd(x):
        return np.clip(np.round(x.astype(np.float64)), 0, 255).astype(np.uint8)
    p, q, t = rnd(vf * (f32(1.0) - fs)), rnd(vf * (f32(1.0) - fs * f)), rnd(vf * (f32(1.0) - fs * (f32(1.0) - f)))
    im = i % 6
    sel = [im == k for k in range(6)]
    r = np.select(sel, [v, q, p, p, t, v])
    g = np.select(sel, [t, v, v, q, p, p])
    b = np.select(sel, [p, p, t, v, v, q])
    gray = s == 0
    return np.stack([np.where(gray, v, r), np.where(gray, v, g), np.where(gray, v, b)], -1).astype(np.uint8)


def hue_shift(hue_factor: float) -> int:
    """np.uint8(hue_factor * 255) of dataloaders.py:588 with the wrap-around of the numpy the reference pins (1.x: a C
    cast through int): truncation toward zero, modulo 256."""
    return int(hue_factor * 255) & 255


def jitter_u8(img: np.ndarray, op: str, factor: float) -> np.ndarray:
    """One adjust_* call of dataloaders.py:491-594 on an 8-bit RGB array."""
    if op == "brightness":                                   # ImageEnhance.Brightness: degenerate = black
        return pil_blend(np.zeros_like(img), img, factor)
    if op == "contrast":                                     # ImageEnhance.Contrast: degenerate = int(mean(L) + 0.5)
        L = pil_l(img)
        mean = int(int(L.astype(np.int64).sum()) / L.size + 0.5)
        return pil_blend(np.full_like(img, mean), img, factor)
    if op == "saturation":                                   # ImageEnhance.Color: degenerate = L replicated
        return pil_blend(np.repeat(pil_l(img)[..., None], 3, -1), img, factor)
    if op == "hue":
        hsv = rgb2hsv(img)
        hsv[..., 0] = (hsv[..., 0].astype(np.int64) + hue_shift(factor)) & 255
        return hsv2rgb(hsv)
    raise ValueError(op)


def jitter_pil(img, op: str, factor: float):
    """The same call through PIL, as the reference makes it."""
    from PIL import Image, ImageEnhance
    if op == "brightness":
        return ImageEnhance.Brightness(img).enhance(factor)
    if op == "contrast":
        return ImageEnhance.Contrast(img).enhance(factor)
    if op == "saturation":
        return ImageEnhance.Color(img).enhance(factor)
    h, s, v = img.convert("HSV").split()                     # adjust_hue :584-591
    np_h = np.array(h, dtype=np.uint8)
    np_h = ((np_h.astype(np.int64) + hue_shift(factor)) & 255).astype(np.uint8)
    return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")


def transform_pil(img, mask, *, flip: bool, scaled_size, pad, crop_xy, crop_size: int, ignore_index: int = 255, blur=None,
                  jitter=None):
    """The reference's PIL calls for one sample with the draws given: img / mask are PIL images ('RGB' / 'L').
    -> (float32 [3,Hc,Wc] in 0..255, float32 [Hc,Wc]) as dataloaders.py ToTensor returns them."""
    from PIL import Image, ImageOps
    if flip:                                                             # dataloaders.py:145-147
        img, mask = img.transpose(Image.FLIP_LEFT_RIGHT), mask.transpose(Image.FLIP_LEFT_RIGHT)
    for op, factor in (jitter or []):                                    # ColorJitter :596-660 (ops in the drawn order)
        img = jitter_pil(img, op, factor)
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


def transform_numpy(img: np.ndarray, lab: np.ndarray, *, flip: bool, scaled_size, pad, crop_xy, crop_size: int,
                    ignore_index: int = 255, blur=None, jitter=None):
    """The same pipeline as `transform_pil` built only from the restated arithmetic above (no PIL): uint8 [H,W,3] and [H,W]
    in -> (uint8 [3,Hc,Wc], uint8 [Hc,Wc]).  Checked against tests/golden/input_pipeline.npz (PIL's outputs)."""
    if flip:
        img, lab = img[:, ::-1], lab[:, ::-1]
    for op, factor in (jitter or []):
        img = jitter_u8(np.ascontiguousarray(img), op, factor)
    w, h = scaled_size
    H, W = lab.shape
    img = resample_u8(np.ascontiguousarray(img), w, h, "bicubic")
    lab = lab[nearest_table(H, h)][:, nearest_table(W, w)]
    pw, ph = pad
    if not (w == crop_size and h == crop_size):
        if pw or ph:
            img = np.pad(img, ((ph, ph), (pw, pw), (0, 0)), constant_values=0)
            lab = np.pad(lab, ((ph, ph), (pw, pw)), constant_values=ignore_index)
        x1, y1 = crop_xy
        img, lab = img[y1:y1 + crop_size, x1:x1 + crop_size], lab[y1:y1 + crop_size, x1:x1 + crop_size]
    if blur is not None and blur != 0:
        img = gaussian_blur_u8(np.ascontiguousarray(img), blur)
    return np.ascontiguousarray(img.transpose(2, 0, 1)), np.ascontiguousarray(lab)
