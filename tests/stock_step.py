"""Yardstick, not a test and not product code (it lives under tests/ because only tests/, smoke() and bench.py's cpu_baseline leg may
import the oracle): the WHOLE training step of the bench workload on stock PyTorch-ROCm -- the oracle's functional model
(oracle/mrfp_oracle.py, the reference's arithmetic in plain torch ops) moved to cuda:0, channels_last, torch.autocast(bfloat16),
MIOpen find mode (torch.backends.cudnn.benchmark = True), torch.optim.SGD(foreach) -- timed beside this library's step on the same box.

    python tests/stock_step.py [--trunk resnet-101] [--size 768] [--batch 16] [--warmup 5] [--steps 20] [--out gpurun_out/stock_step.json]

BASELINE.json configs[1] names "stock ROCm convs"; VERDICT r4 item 3(b) asks for this line in DESIGN section 6.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trunk", default="resnet-101")
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-find", action="store_true")
    ap.add_argument("--out", default="gpurun_out/stock_step.json")
    args = ap.parse_args()
    from mrfp_amd import deepv3, synth
    from oracle import mrfp_oracle as orc
    torch.backends.cudnn.benchmark = not args.no_find
    dev = torch.device("cuda:0")
    m = deepv3.MRFPPlus(19, trunk=args.trunk)
    spec = synth.spec_of(m.state_dict())
    del m
    sd = synth.synth_state_dict(spec, seed=0)
    CL = torch.channels_last
    sd = {k: (v.to(dev).contiguous(memory_format=CL) if v.dim() == 4 else v.to(dev)) for k, v in sd.items()}
    B, S = args.batch, args.size
    x, y = synth.synth_batch(B, S, S, seed=1)
    x = x.to(dev).contiguous(memory_format=CL)
    y = y.to(dev)
    noise = synth.synth_noise(B, seed=2, channels=(64 if args.trunk == "resnet-50" else 128, 256))
    noise = {k: v.to(dev) for k, v in noise.items()}
    keys = orc.trainable_keys(sd)
    leaf = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = dict(sd)
    work.update(leaf)
    opt = torch.optim.SGD([leaf[k] for k in keys], lr=1e-6, momentum=0.9, weight_decay=5e-4, foreach=True)
    it = [0]

    def step():
        new_stats = {}
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
            loss = orc.mrfp_forward(work, x, y, training=True, toggles=(True, True, True), noise=noise, new_stats=new_stats)
        loss.backward()
        for g in opt.param_groups:
            g["lr"] = 1e-6 * orc.poly_lr_factor(it[0])
        opt.step()
        with torch.no_grad():
            for k, v in new_stats.items():
                work[k].copy_(v)
        it[0] += 1
        return loss

    t0 = time.time()
    # heartbeat: the first step runs MIOpen's find for every distinct convolution (forward, data and weight gradients) -- minutes
    # without output, and the GPU pool takes a silent command for a hung one
    import threading
    stop = threading.Event()

    def beat():
        while not stop.wait(30.0):
            print("[stock_step] ... %.0f s, iteration %d" % (time.time() - t0, it[0]), file=sys.stderr, flush=True)
    threading.Thread(target=beat, daemon=True).start()
    for i in range(args.warmup):
        loss = step()
        torch.cuda.synchronize()
        print("[stock_step] warm-up %d at %.1f s, loss %.4f" % (i, time.time() - t0, float(loss)), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.steps):
        step()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / args.steps
    stop.set()
    out = {"what": "stock PyTorch-ROCm step: oracle model on cuda:0, channels_last, autocast(%s), MIOpen find mode %s, SGD(foreach)"
                   % (args.dtype, "off" if args.no_find else "on"),
           "trunk": args.trunk, "batch": B, "size": S, "ms_per_step": round(ms, 3), "images_per_s": round(B / ms * 1e3, 2),
           "steps": args.steps, "warmup": args.warmup, "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
           "torch": torch.__version__}
    print(json.dumps(out))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
