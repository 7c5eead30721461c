#!/usr/bin/env python3
"""Benchmark of the MRFP+ training hot path (one step = forward + backward + fused SGD on one synthetic
batch, inputs resident in HBM).  Contract: prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json's metric is quoted on ResNet-101 DeepLabV3+ + MRFP+ at 768x768,
16 images per GPU, bf16 activations (configs[2]; configs[3] = the same per rank on 8 GPUs, weak scaling).
All three perturbation toggles are forced ON (worst case).
"""
import argparse
import ctypes
import json
import os
import sys
import time

# Before the HIP runtime starts: the training step uses several streams (compute, weight gradients, the gradient
# all-reduce side stream, RCCL's own).  With the runtime's default of 4 hardware queues they alias once a process group
# exists and the overlap turns into serialisation + barrier packets: 65.6 instead of 61.4 ms per step with the
# data-parallel machinery on (measured at one rank, tools/host_time_sync.py); at N = 1 without a process group it is neutral.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X (guide: ~2.5 PF dense)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--trunk", default="resnet-101", choices=["resnet-50", "resnet-101", "wider_resnet38_a2"])
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--width", type=int, default=0, help="input width when not square (configs[4]: --size 1024 --width 2048)")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay zero_grad + forward + backward as a hipGraph (single GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--dump-convs", default=None, help="write per-launch conv shapes/timings (JSON) here")
    ap.add_argument("--fourier", action="store_true", help="variant: also attach the build-defined multi-resolution Fourier "
                    "amplitude perturbation (perturb.MultiResolutionFourier, every step) -- NOT the reference path, "
                    "reported as its own workload")
    return ap.parse_args()


class HipTimer:
    """hipEvents on a given stream through libamdhip64 (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so.7")   # the copy torch already loaded (same SONAME)
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]

    def event(self):
        e = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.hip.hipEventRecord(e, ctypes.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        self.hip.hipEventSynchronize(b)
        ms = ctypes.c_float()
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return ms.value


def cpu_baseline(args, seconds):
    """The CPU oracle (a port of the reference arithmetic, validated bit-exact against the reference in the
    build container) timed on this host's cores on a bounded sample of the same workload."""
    import json as js
    from mrfp_amd import synth
    from oracle import mrfp_oracle as orc
    # the GPU box gives one GPU's share of the host (16 cores); more threads than that only contend
    cores = int(os.environ.get("MRFP_CPU_THREADS", min(os.cpu_count() or 1, 16)))
    torch.set_num_threads(cores)
    print("[bench] cpu_baseline: oracle on %d host threads ..." % cores, file=sys.stderr, flush=True)
    from mrfp_amd import deepv3
    from mrfp_amd.config import cfg
    m = deepv3.MRFPPlus(19, trunk=args.trunk)
    spec = synth.spec_of(m.state_dict())
    del m
    sd = synth.synth_state_dict(spec, seed=0)
    B, S = 2, 256
    x, y = synth.synth_batch(B, S, S, seed=1)
    noise = synth.synth_noise(B, seed=2, channels=(64 if args.trunk == "resnet-50" else 128, 256))
    keys = orc.trainable_keys(sd)
    mom, it = {}, [0]

    def step():
        """one full train iteration as reference main.py:857-864: forward, backward, SGD(momentum, wd) + poly LR,
        BatchNorm running-statistics update (oracle.train_steps restated for one iteration)."""
        leaf = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
        work = dict(sd)
        work.update(leaf)
        new_stats = {}
        loss = orc.mrfp_forward(work, x, y, training=True, toggles=(True, True, True), noise=noise, new_stats=new_stats)
        grads = torch.autograd.grad(loss, [leaf[k] for k in keys])
        with torch.no_grad():
            orc.sgd_step({k: sd[k] for k in keys}, dict(zip(keys, grads)), mom,
                         lr=1e-6 * orc.poly_lr_factor(it[0]), first=(it[0] == 0))     # tiny lr: the timing sample stays finite
            for k, v in new_stats.items():
                sd[k].copy_(v)
        it[0] += 1
    t0 = time.time()
    for _ in range(3):                             # warm-up (allocator, oneDNN primitive caches, momentum buffers)
        step()
    print("[bench] cpu_baseline: 3 warm-up steps %.1f s" % (time.time() - t0), file=sys.stderr, flush=True)
    t0, n = time.time(), 0
    while n < 5 or (time.time() - t0 < seconds and n < 50):
        step()
        n += 1
        print("[bench] cpu_baseline: step %d at %.1f s" % (n, time.time() - t0), file=sys.stderr, flush=True)
    dt_ = time.time() - t0
    width = args.width or args.size
    scale = (args.size * width) / float(S * S)
    return {"value": round(B * n / dt_, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "per_pixel_scaled_value": round(B * n / dt_ / scale, 4),
            "sample": "%d train steps (fwd+bwd+SGD, after 3 warm-up steps) of %s MRFP+ at %dx%dx%d fp32 on the CPU oracle; "
                      "work/image is proportional to H*W, so at the bench size %dx%d (x%.0f pixels) the same host would "
                      "deliver per_pixel_scaled_value images/sec" % (n, args.trunk, B, S, S, args.size, width, scale)}


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves.  The reference's
    own launch is a single `python main.py` (reference main_script.sh:1, main.py:39-52), so nothing upstream supplies a
    launcher.  Runs BEFORE anything touches the GPU, as a CHILD process (never exec: a process that has initialised
    HIP must not be replaced), relays the child's output (rank 0 prints the one JSON line) and returns its exit code."""
    import socket
    import subprocess
    port = os.environ.get("MRFP_BENCH_PORT")
    if port is None:
        with socket.socket() as s:                       # a free port on the loopback interface
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("MRFP_FORCE_SYNC") == "1"
    # MRFP_BENCH_SHARE_GPU=1 + MRFP_DIST_BACKEND=gloo: rehearsal of the N>1 code path on a one-GPU box (every rank on
    # cuda:0, gloo transport -- RCCL refuses two ranks on one device); tests/test_ddp_gpu.py drives it
    share = os.environ.get("MRFP_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("MRFP_DIST_BACKEND", "nccl")
    if share:
        local = 0
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local if world > 1 else 0)

    from mrfp_amd import synth, deepv3
    from mrfp_amd.config import cfg
    from mrfp_amd.harness import Trainer
    cfg.MODEL.ACT_DTYPE = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model = deepv3.MRFPPlus(19, trunk=args.trunk, criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
    model.load_state_dict(synth.synth_state_dict(synth.spec_of(model.state_dict()), seed=0))
    model = model.to(dev).train()
    model.rng = deepv3.InjectedRandom((True, True, True), None, reinit=True)   # all perturbations on, HRFP re-drawn
    if args.fourier:
        from mrfp_amd.perturb import MultiResolutionFourier
        model.fourier_perturb = MultiResolutionFourier(p=1.0)
    trainer = Trainer(model)
    if args.graph:
        trainer.enable_graph()
    width = args.width or args.size
    x, y = synth.synth_batch(args.batch, args.size, width, seed=1 + rank)
    x, y = x.to(dev), y.to(dev)

    for _ in range(args.warmup):
        trainer.step(x, y)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(x, y)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    lossv = float(loss.detach())
    ranks_seen = 1
    if use_dist:                      # proof that N ranks met on the data-path communicator: a sum of ones over it
        ones = torch.ones(1, device=dev, dtype=torch.float32)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    # roofline of the dominant kernel family (the MFMA implicit-GEMM convolutions): algorithmic FLOPs of all conv
    # launches of one step / their summed device time, measured with hipEvents around each launch of ONE EXTRA step.
    # That step runs on EVERY rank (it issues the gradient all-reduces like any other step: rank 0 alone would leave the
    # other ranks' collectives unmatched); only rank 0 reports its timings.
    roof = conv_roofline(model, trainer, x, y, args)
    out = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = args.batch * world * args.steps / elapsed
        out = {"metric": "train images/sec", "value": round(value, 3), "unit": "images/sec", "n_gpus": world,
               "ranks_seen": ranks_seen,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "%s DeepLabV3+ + MRFP+ (HRFP+NP+ on, HRFP re-drawn every step%s), %dx%d, "
                                      "%d images/GPU, fwd+bwd+SGD, synthetic 19-class, random-init weights"
                                      % (args.trunk, ", + build-defined multi-resolution Fourier amplitude mix at stem/layer1/"
                                         "layer2 every step" if args.fourier else "", args.size, width, args.batch),
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "final_loss": round(lossv, 5)},
               "roofline": roof}
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))


def _eager_step(trainer, x, y):
    """one eager (non-graph) step: the per-launch timing below spies on the Python-side conv calls"""
    import mrfp_amd.conv as conv_mod
    was, trainer.graph = trainer.graph, False
    side, conv_mod.USE_WGRAD_STREAM[0] = conv_mod.USE_WGRAD_STREAM[0], False      # every conv on the timed stream
    try:
        return trainer.step(x, y)
    finally:
        trainer.graph = was
        conv_mod.USE_WGRAD_STREAM[0] = side


def conv_roofline(model, trainer, x, y, args):
    """Times every MFMA convolution launch (fwd / dgrad / wgrad) of one train step with hipEvents recorded on
    the launch stream, and divides their algorithmic FLOPs by the summed durations."""
    import mrfp_amd.conv as conv_mod
    from mrfp_amd import _lib
    timer = HipTimer()
    events, flops, shapes = [], [], []
    orig = _lib.call
    st = torch.cuda.current_stream().cuda_stream

    def spy(name, *a):
        if name in ("mrfp_conv_fwd", "mrfp_conv_fwd_gated", "mrfp_conv_wgrad"):
            if name in ("mrfp_conv_fwd", "mrfp_conv_fwd_gated"):      # (the gated form: same leading arguments)
                B, H, W, C, N, ldy, R, S, Ho, Wo = a[5:15]
                f = 2.0 * B * Ho * Wo * N * R * S * C / float(a[19] * a[19])
            else:
                B, H, W, C, Ct, N, ldn, R, S, Ho, Wo = a[5:16]
                f = 2.0 * B * Ho * Wo * N * R * S * C
            e0, e1 = timer.event(), timer.event()
            timer.record(e0, st)
            r = orig(name, *a)
            timer.record(e1, st)
            events.append((e0, e1))
            flops.append(f)
            shapes.append(("mrfp_conv_fwd" if name == "mrfp_conv_fwd_gated" else name,
                           [int(v) for v in a[5:20]]))
            return r
        return orig(name, *a)
    conv_mod.call = spy
    try:
        _eager_step(trainer, x, y)
        torch.cuda.synchronize()
    finally:
        conv_mod.call = orig
    ms = [timer.elapsed_ms(a, b) for a, b in events]
    tot_ms, tot_f = sum(ms), sum(flops)
    if args.dump_convs:
        with open(args.dump_convs, "w") as f:
            json.dump([{"name": n, "args": sh, "ms": m, "tflops": fl / (m * 1e-3) / 1e12 if m > 0 else 0, "gflop": fl / 1e9}
                       for (n, sh), m, fl in zip(shapes, ms, flops)], f)
    peak = PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS      # f16 and bf16 MFMA: same dense rate
    ach = tot_f / (tot_ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": "conv_igemm_kernel+conv_wgrad_kernel (all %d conv launches of one step)" % len(ms),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": None, "conv_ms_per_step": round(tot_ms, 3), "conv_tflop_per_step": round(tot_f / 1e12, 3)}
    roof.update(measured_traffic(args))
    return roof


def measured_traffic(args):
    """HBM bytes of the conv kernel family per step from the PMC counters (TCC_EA0_RDREQ / WRREQ passes collected by
    tools/measure_traffic.sh exactly as MI355X_MICROARCH.md prescribes, in their own rocprofv3 runs) -- read from the
    committed profiles/r02_traffic.json, which names the commit and the workload it was measured on.  Counters cannot
    be collected from inside this process, so the figure is attached only when that file describes THIS workload."""
    path = os.path.join(ROOT, "profiles", "r02_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return {}
    w = t.get("workload", {})
    if (w.get("trunk"), w.get("size"), w.get("width"), w.get("batch"), w.get("dtype")) != \
            (args.trunk, args.size, args.width or args.size, args.batch, args.dtype):
        return {}
    return {"traffic": t.get("conv_family_hbm_bytes_per_step"), "traffic_unit": "bytes/step (conv kernel family, PMC)",
            "traffic_algorithmic": t.get("conv_family_algorithmic_bytes_per_step"),
            "traffic_measured_at_commit": t.get("commit"), "traffic_source": "profiles/r02_traffic.json"}


if __name__ == "__main__":
    main()
