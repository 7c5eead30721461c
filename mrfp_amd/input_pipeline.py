"""The reference's training transform (main.py:409-419 transform_tr) on the GPU, bit-exact with its PIL calls.

Reference: main.py:409-419 `transform_tr` = RandomHorizontalFlip -> ColorJitter -> RandomSizeAndCrop(crop_size,
crop_nopad=False, ignore_index=255) -> Resize(crop_size) -> RandomGaussianBlur -> ToTensor (dataloaders.py).  This module
does flip, ColorJitter, the BICUBIC / NEAREST rescale, the ImageOps.expand padding, the crop, the Gaussian blur and ToTensor on the device: uint8 image
and label map in, float32 [3,H,W] (0..255) and int64 [H,W] out, byte for byte what PIL produces (tests/test_input_gpu.py).
RandomGaussianBlur (:168-177) is included: its radius is random.random() < 1, for which ImageFilter.GaussianBlur is three
horizontal + three vertical passes of a 3-tap fixed-point box blur.  ColorJitter (dataloaders.py:596-660) is included: PIL's
ImageEnhance blends (Blend.c float arithmetic) and the RGB -> HSV -> RGB round trip of adjust_hue (Convert.c), per pixel,
the contrast mean reduced on the device; the Resize step is the identity here (the crop already has crop_size) and PIL
returns a copy for it.  One deviation is stated in oracle/input_oracle.py::hue_shift: `np.uint8(hue_factor * 255)` of a
negative factor is taken with the wrap-around of the numpy 1.x the reference pins.

The fixed-point coefficient tables of Pillow's resampler are built on the host exactly as Pillow builds them
(src/libImaging/Resample.c, double precision) and cached per (source size, destination size); the kernels
(csrc/input.hip) do the integer arithmetic."""
from __future__ import annotations

import math
import random as _random
from dataclasses import dataclass
from functools import lru_cache
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream

PRECISION_BITS = 32 - 8 - 2


def _bicubic_vec(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@lru_cache(maxsize=256)
def _bicubic_tables(in_size: int, out_size: int):
    """Pillow precompute_coeffs(BICUBIC) + normalize_coeffs_8bpc -> (bounds int32 [out,2], coefs int32 [out,ksize]).
    Vectorised over the destination index with the same IEEE double operations in the same order as Pillow's scalar loop
    (the weight sum is a sequential cumsum, not numpy's pairwise sum); checked entry by entry against the scalar
    restatement in oracle/input_oracle.py (tests/test_input_cpu.py)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)            # (int) truncates; the argument is > -1
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    t = np.arange(ksize, dtype=np.float64)[None, :]
    live = np.arange(ksize)[None, :] < xmax[:, None]
    w = np.where(live, _bicubic_vec((t + xmin[:, None] - center[:, None] + 0.5) * ss), 0.0)
    ww = np.cumsum(w, axis=1)[:, -1:]                                           # sequential adds, trailing zeros change nothing
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    one = float(1 << PRECISION_BITS)
    coefs = np.where(w < 0, -0.5 + w * one, 0.5 + w * one).astype(np.int32)     # C (int) cast: truncation
    coefs = np.where(live, coefs, 0).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, np.ascontiguousarray(coefs)


@lru_cache(maxsize=256)
def _nearest_table(in_size: int, out_size: int) -> np.ndarray:
    """ImagingScaleAffine: xo = a/2, xin = (int)xo, xo += a (accumulated in double: a sequential cumsum)."""
    a = float(in_size) / out_size
    steps = np.full(out_size, a, dtype=np.float64)
    steps[0] = 0.0 + a * 0.5
    xo = np.cumsum(steps)
    return np.where(xo < 0.0, -1, xo.astype(np.int64)).astype(np.int32)


def _blur_weights(radius: float):
    """ImageFilter.GaussianBlur(radius) -> the (ww, fw) 24-bit weights of its three box-blur passes per axis, derived as
    Pillow derives them (BoxBlur.c _gaussian_blur_radius in C float arithmetic, then ww = (UINT32)(2^24 / (2 r + 1)),
    fw = (2^24 - (2 int(r) + 1) ww) / 2).  Only box radii below 1 (every radius = random.random() gives one)."""
    f32 = np.float32
    r = f32(radius)
    sigma2 = f32(f32(r * r) / f32(3))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32(f32(f32(2) * l + f32(1)) * f32(f32(l * f32(l + f32(1))) - f32(f32(3) * sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - f32(f32(l + f32(1)) * f32(l + f32(1))))))
    fr = f32(l + a)
    if int(fr) != 0:
        raise _lib.MrfpHipError("GaussianBlur radius %r gives a box radius >= 1: only the reference's range [0, 1) is built" % radius)
    ww = int(f32(f32(1 << 24) / f32(fr * f32(2) + f32(1))))
    fw = ((1 << 24) - ww) // 2
    return ww, fw


_JITTER_OPS = {"brightness": 0, "contrast": 1, "saturation": 2, "hue": 3}


@dataclass
class Draw:
    flip: bool
    jitter: Optional[list]             # ColorJitter: [(op, factor), ...] in application order when its gate fired, else None
    scaled: Tuple[int, int]            # (w, h) after RandomSizeAndCrop's rescale
    pad: Tuple[int, int]               # (pad_w, pad_h) of ImageOps.expand on every side
    crop: Tuple[int, int]              # (x1, y1) in the padded image
    blur: Optional[float]              # GaussianBlur radius when its gate fired


class TrainTransform:
    """transform_tr of the reference (main.py:409-419) on the device."""

    def __init__(self, crop_size: int, scale_min: float = 0.5, scale_max: float = 2.0, ignore_index: int = 255):
        self.crop_size, self.scale_min, self.scale_max, self.ignore_index = int(crop_size), scale_min, scale_max, ignore_index
        self._dev_tables = {}

    JITTER = dict(brightness=0.5, hue=0.3, contrast=0.2, saturation=0.2)      # main.py:412

    def draw(self, w: int, h: int, rng=_random, np_rng=np.random) -> Draw:
        """Consumes python's `random` stream in the reference's order (dataloaders.py:145, 655, 421, 327-331, 172-174) and,
        when the ColorJitter gate fires, numpy's global stream as get_params does (:622-643: four uniform factors, then
        np.random.shuffle of the four transforms)."""
        flip = rng.random() < 0.5
        jitter = None
        if rng.random() < 0.5:
            j = self.JITTER
            ops = [("brightness", float(np_rng.uniform(max(0, 1 - j["brightness"]), 1 + j["brightness"]))),
                   ("contrast", float(np_rng.uniform(max(0, 1 - j["contrast"]), 1 + j["contrast"]))),
                   ("saturation", float(np_rng.uniform(max(0, 1 - j["saturation"]), 1 + j["saturation"]))),
                   ("hue", float(np_rng.uniform(-j["hue"], j["hue"])))]
            np_rng.shuffle(ops)                            # consumes the stream as shuffling the four Lambdas does
            jitter = ops
        scale_amt = 1.0 * rng.uniform(self.scale_min, self.scale_max)
        sw, sh = int(w * scale_amt), int(h * scale_amt)
        t = self.crop_size
        pad_w = pad_h = 0
        x1 = y1 = 0
        if not (sw == t and sh == t):
            pad_h = (t - sh) // 2 + 1 if t > sh else 0
            pad_w = (t - sw) // 2 + 1 if t > sw else 0
            W2, H2 = sw + 2 * pad_w, sh + 2 * pad_h
            x1 = 0 if W2 == t else rng.randint(0, W2 - t)
            y1 = 0 if H2 == t else rng.randint(0, H2 - t)
        blur = rng.random() if rng.random() < 0.5 else None
        return Draw(flip, jitter, (sw, sh), (pad_w, pad_h), (x1, y1), blur)

    def _tables(self, dev, H, W, sh, sw):
        key = (str(dev), H, W, sh, sw)
        t = self._dev_tables.get(key)
        if t is None:
            bx, kx = _bicubic_tables(W, sw)
            by, ky = _bicubic_tables(H, sh)
            t = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in
                      (bx, kx, by, ky, _nearest_table(W, sw), _nearest_table(H, sh))) + (kx.shape[1], ky.shape[1])
            if len(self._dev_tables) > 64:
                self._dev_tables.clear()
            self._dev_tables[key] = t
        return t

    def __call__(self, img_u8: torch.Tensor, lab_u8: torch.Tensor, d: Draw, out_img: Optional[torch.Tensor] = None,
                 out_lab: Optional[torch.Tensor] = None):
        """img_u8: uint8 [H,W,3], lab_u8: uint8 [H,W], both on the GPU -> (float32 [3,T,T], int64 [T,T])."""
        if not (img_u8.is_cuda and lab_u8.is_cuda and img_u8.dtype == torch.uint8 and lab_u8.dtype == torch.uint8):
            raise _lib.MrfpHipError("TrainTransform: uint8 CUDA tensors expected (there is no CPU path)")
        H, W, C = img_u8.shape
        if C != 3 or tuple(lab_u8.shape) != (H, W):
            raise _lib.MrfpHipError("TrainTransform: image [H,W,3] and label [H,W] expected")
        img_u8, lab_u8 = img_u8.contiguous(), lab_u8.contiguous()
        dev, T = img_u8.device, self.crop_size
        sw, sh = d.scaled
        bx, kx, by, ky, tx, ty, ksx, ksy = self._tables(dev, H, W, sh, sw)
        cur = img_u8
        if d.jitter:                                       # ColorJitter on the original-size image (per-pixel: commutes with the flip)
            ws = torch.empty(16, dtype=torch.uint8, device=dev)
            for op, factor in d.jitter:
                nxt = torch.empty_like(cur)
                shift = int(factor * 255) & 255 if op == "hue" else 0
                call("mrfp_jitter_u8", ptr(cur), ptr(nxt), H * W, _JITTER_OPS[op], float(factor), shift, ptr(ws), stream())
                cur = nxt
        if sw != W or d.flip:       # horizontal pass first (Pillow ImagingResample), reading the source mirrored when flipped
            # (at sw == W the coefficients are exactly (0, 1, 0): the pass is then a plain mirrored copy)
            tmp = torch.empty((H, sw, 3), dtype=torch.uint8, device=dev)
            call("mrfp_resample_u8", ptr(cur), ptr(tmp), H, W, H, sw, 3, ptr(bx), ptr(kx), ksx, 0, int(d.flip), stream())
            cur = tmp
        if sh != H:
            tmp = torch.empty((sh, sw, 3), dtype=torch.uint8, device=dev)
            call("mrfp_resample_u8", ptr(cur), ptr(tmp), H, sw, sh, sw, 3, ptr(by), ptr(ky), ksy, 1, 0, stream())
            cur = tmp
        if out_img is None:
            out_img = torch.empty((3, T, T), dtype=torch.float32, device=dev)
        if out_lab is None:
            out_lab = torch.empty((T, T), dtype=torch.int64, device=dev)
        blur = d.blur is not None and d.blur != 0.0          # PIL returns a copy for radius 0
        crop_u8 = torch.empty((T, T, 3), dtype=torch.uint8, device=dev) if blur else None
        call("mrfp_input_assemble", ptr(cur), ptr(lab_u8), ptr(ty), ptr(tx), sh, sw, H, W, int(d.flip), d.pad[0], d.pad[1],
             d.crop[0], d.crop[1], T, T, int(self.ignore_index), ptr(out_img), ptr(crop_u8), ptr(out_lab), stream())
        if blur:                                              # RandomGaussianBlur (dataloaders.py:168-177), then ToTensor
            ww, fw = _blur_weights(d.blur)
            a, b = crop_u8, torch.empty_like(crop_u8)
            for vertical in (0, 0, 0, 1, 1, 1):               # ImagingBoxBlur: three passes along x, then three along y
                call("mrfp_box_blur3_u8", ptr(a), ptr(b), T, T, 3, ww, fw, vertical, stream())
                a, b = b, a
            call("mrfp_u8hwc_to_f32chw", ptr(a), ptr(out_img), T, T, stream())
        return out_img, out_lab
