"""Timing of the GPU input transform (the reference's transform_tr, byte-exact with PIL) against the same PIL calls
on one host core.    python tools/input_micro.py [H W crop reps]"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd.input_pipeline import TrainTransform  # noqa: E402


def pil_calls(img, mask, d, crop, ignore=255):
    """The reference's PIL calls for one draw (dataloaders.py: flip, ColorJitter, resize, expand, crop, blur, ToTensor)."""
    from PIL import Image, ImageEnhance, ImageFilter, ImageOps
    if d.flip:
        img, mask = img.transpose(Image.FLIP_LEFT_RIGHT), mask.transpose(Image.FLIP_LEFT_RIGHT)
    for op, f in (d.jitter or []):
        if op == "hue":
            h, s, v = img.convert("HSV").split()
            nh = ((np.array(h, dtype=np.int64) + (int(f * 255) & 255)) & 255).astype(np.uint8)
            img = Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")
        else:
            enh = {"brightness": ImageEnhance.Brightness, "contrast": ImageEnhance.Contrast, "saturation": ImageEnhance.Color}[op]
            img = enh(img).enhance(f)
    img, mask = img.resize(d.scaled, Image.BICUBIC), mask.resize(d.scaled, Image.NEAREST)
    if d.pad[0] or d.pad[1]:
        b = (d.pad[0], d.pad[1], d.pad[0], d.pad[1])
        img, mask = ImageOps.expand(img, border=b, fill=(0, 0, 0)), ImageOps.expand(mask, border=b, fill=ignore)
    x1, y1 = d.crop
    img, mask = img.crop((x1, y1, x1 + crop, y1 + crop)), mask.crop((x1, y1, x1 + crop, y1 + crop))
    if d.blur is not None:
        img = img.filter(ImageFilter.GaussianBlur(radius=d.blur))
    return np.array(img).astype(np.float32).transpose((2, 0, 1)), np.array(mask).astype(np.float32)


def main():
    a = [int(v) for v in sys.argv[1:]]
    H, W, crop, reps = (a + [1024, 2048, 768, 20][len(a):])[:4]
    from PIL import Image
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    lab = rng.integers(0, 19, (H, W), dtype=np.uint8)
    tt = TrainTransform(crop)
    r = random.Random(0)
    draws = [tt.draw(W, H, r) for _ in range(reps)]
    xi, xl = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    out_i = torch.empty(3, crop, crop, device="cuda")
    out_l = torch.empty(crop, crop, dtype=torch.int64, device="cuda")
    for d in draws[:3]:
        tt(xi, xl, d, out_i, out_l)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for d in draws[3:]:                  # every draw has a new scaled size: host tables built + uploaded inside the timing
        tt(xi, xl, d, out_i, out_l)
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t0) / max(1, reps - 3) * 1e3
    t0 = time.perf_counter()
    for d in draws:                      # tables cached
        tt(xi, xl, d, out_i, out_l)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    pi, pl = Image.fromarray(img), Image.fromarray(lab)
    t0 = time.perf_counter()
    for d in draws:
        pil_calls(pi, pl, d, crop)
    cpu_ms = (time.perf_counter() - t0) / reps * 1e3
    mean_scale = float(np.mean([d.scaled[0] / W for d in draws]))
    print(json.dumps({"op": "transform_tr (flip, ColorJitter and Gaussian blur on half the draws each, bicubic rescale, pad, crop, ToTensor)", "source": [H, W], "crop": crop,
                      "mean_scale": round(mean_scale, 3), "gpu_ms_per_image": round(gpu_ms, 3), "gpu_ms_per_image_new_tables": round(cold_ms, 3),
                      "pil_ms_per_image_one_core": round(cpu_ms, 2), "gpu_images_per_s": round(1e3 / gpu_ms, 1)}))


if __name__ == "__main__":
    main()
