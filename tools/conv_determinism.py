"""Every forward-type convolution launch shape of the bench workload (tools/bench_conv_shapes.json, from `bench.py
--dump-convs`), run three times on the same operands: bit-identical outputs, and close to torch's fp32 convolution of the
same bf16 operands.  python tools/conv_determinism.py [reps]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrfp_amd import conv                   # noqa: E402
from mrfp_amd.config import cfg             # noqa: E402

cfg.MODEL.ACT_DTYPE = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # 0: the batch of the dump (16)
shapes = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_conv_shapes.json")))
bad = 0
g = torch.Generator(device="cuda").manual_seed(0)
for name, a in shapes:
    if name != "mrfp_conv_fwd":
        continue
    B, H, W, C, N, ldy, R, S, Ho, Wo, stride, ph, pw, dil, sstride = a
    if sstride != 1 or ldy != N or ph != pw or R != S:
        continue
    B = batch or B
    x = torch.empty(B, C, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x.normal_(generator=g)
    w = torch.empty(N, C, R, S, device="cuda").normal_(generator=g) * 0.05
    with torch.no_grad():
        ys, sts = [], []
        for _ in range(reps):
            y = conv.conv2d(x, w, None, stride, ph, dil)
            st = getattr(y, "_mrfp_colstats", None)
            ys.append(y.clone())
            sts.append(None if st is None else st[0].clone())
        same = all(torch.equal(ys[0], y) for y in ys[1:]) and all(s is None or torch.equal(sts[0], s) for s in sts[1:])
        ref = F.conv2d(x.float(), w.bfloat16().float(), None, stride, ph, dil)
        rel = ((ys[0].float() - ref).abs().max() / ref.abs().max()).item()
    flag = "" if same and rel < 1.5e-2 else "   <<<<<<"
    bad += bool(flag)
    print("%-70s same=%s rel=%.2e%s" % (str(a), same, rel, flag), flush=True)
    del x, w, ys, ref, sts
print("BAD" if bad else "OK", bad)
