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
from oracle import input_oracle as io  # noqa: E402  (timed here as the CPU baseline only)


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
        io.transform_pil(pi, pl, flip=d.flip, scaled_size=d.scaled, pad=d.pad, crop_xy=d.crop, crop_size=crop, blur=d.blur, jitter=d.jitter)
    cpu_ms = (time.perf_counter() - t0) / reps * 1e3
    mean_scale = float(np.mean([d.scaled[0] / W for d in draws]))
    print(json.dumps({"op": "transform_tr (flip, ColorJitter and Gaussian blur on half the draws each, bicubic rescale, pad, crop, ToTensor)", "source": [H, W], "crop": crop,
                      "mean_scale": round(mean_scale, 3), "gpu_ms_per_image": round(gpu_ms, 3), "gpu_ms_per_image_new_tables": round(cold_ms, 3),
                      "pil_ms_per_image_one_core": round(cpu_ms, 2), "gpu_images_per_s": round(1e3 / gpu_ms, 1)}))


if __name__ == "__main__":
    main()
