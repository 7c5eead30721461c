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
    ap.add_argument("--dump-launches", default=None, help="diagnostic: time EVERY C-ABI launch of one step on its own, write JSON here")
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


def _free_port():
    import socket
    with socket.socket() as s:                       # a free port on the loopback interface
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def launch_ranks(args, script=None, argv=None):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks ourselves.  The reference's
    own launch is a single `python main.py` (reference main_script.sh:1, main.py:39-52), so nothing upstream supplies a
    launcher.  Runs BEFORE anything touches the GPU, as a CHILD process (never exec: a process that has initialised
    HIP must not be replaced), relays the child's output (rank 0 prints the one JSON line) and returns its exit code.
    The rendezvous port is found by bind-and-release (torch.distributed.run has to bind it itself), which leaves a small
    window for another process to take it: if the child dies with "address already in use" before any rank printed a
    result, ONE more attempt is made, in a fresh child on a fresh port (MRFP_BENCH_PORT pins the port: no retry then)."""
    import subprocess
    import threading
    script = script or os.path.abspath(__file__)
    argv = sys.argv[1:] if argv is None else argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # N ranks share this machine's cores: each rank's host side is one Python thread enqueueing launches (+ the autograd thread)
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(4, (os.cpu_count() or 4) // max(1, args.gpus)))))
    fixed = os.environ.get("MRFP_BENCH_PORT")
    rc = 1
    for attempt in range(1 if fixed else 2):
        port = fixed or _free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, script] + list(argv)
        print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        child = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        seen = {"inuse": False, "result": False}

        def relay(src, dst, key):
            for line in src:
                if key == "err" and ("EADDRINUSE" in line or "ddress already in use" in line):
                    seen["inuse"] = True
                if key == "out" and line.startswith("{"):
                    seen["result"] = True
                dst.write(line)
                dst.flush()
        th = [threading.Thread(target=relay, args=(child.stdout, sys.stdout, "out"), daemon=True),
              threading.Thread(target=relay, args=(child.stderr, sys.stderr, "err"), daemon=True)]
        for t in th:
            t.start()
        rc = child.wait()
        for t in th:
            t.join(timeout=10)
        if rc == 0 or not seen["inuse"] or seen["result"]:
            break
        print("[bench] rendezvous port %s was taken between probing and binding: one more attempt on a fresh port" % port,
              file=sys.stderr, flush=True)
    return rc


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
        import datetime
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # The first multi-GPU run is the driver's, blind: make a failure attributable.  (a) more ranks than
        # devices is one clear line and a non-zero exit, checked HERE, in the rank (torch.cuda.device_count() does not initialise
        # the GPU on this image; the launching parent never looks at the devices); (b) the rendezvous and every collective carry an
        # explicit timeout (MRFP_DIST_TIMEOUT_S, default 120 s) so that a missing rank ends the run with a message instead of the
        # driver's limit; (c) every rank reports its first step on stderr.
        ndev = torch.cuda.device_count()
        if not share and local >= ndev:
            print("[bench] rank %d (local rank %d of %d): this node exposes only %d GPU(s) to the process -- `--gpus %d` needs %d; "
                  "nothing was run" % (rank, local, world, ndev, world, world), file=sys.stderr, flush=True)
            sys.exit(3)
        tmo = datetime.timedelta(seconds=float(os.environ.get("MRFP_DIST_TIMEOUT_S", "120")))
        torch.cuda.set_device(local)
        t_init = time.perf_counter()
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        print("[bench] rank %d/%d on cuda:%d: process group (%s) up after %.1f s" % (rank, world, local, backend,
              time.perf_counter() - t_init), file=sys.stderr, flush=True)
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

    t_first = time.perf_counter()
    for i in range(args.warmup):
        trainer.step(x, y)
        if i == 0 and use_dist:
            torch.cuda.synchronize()
            print("[bench] rank %d: first step %.0f ms" % (rank, 1e3 * (time.perf_counter() - t_first)), file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    trace = os.environ.get("MRFP_BENCH_TRACE") == "1"      # diagnostic: device time of every step of the timed region (hipEvents, no host sync)
    if trace:
        tm, evs = HipTimer(), []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if trace:
            evs.append(tm.event())
            tm.record(evs[-1], torch.cuda.current_stream().cuda_stream)
        loss = trainer.step(x, y)
    if trace:
        evs.append(tm.event())
        tm.record(evs[-1], torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    if trace and rank == 0:
        print("[bench] per-step device ms: " + " ".join("%.1f" % tm.elapsed_ms(evs[i], evs[i + 1]) for i in range(len(evs) - 1)),
              file=sys.stderr, flush=True)
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
    roof = step_roofline(model, trainer, x, y, args)
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


# ---------------------------------------------------------------------------------------------------------------------
# Roofline of one train step, measured live: every C-ABI launch of ONE EXTRA eager step is attributed to a kernel family
# (mrfp_amd._lib.HOOK sees the entry point's name and arguments right before it is called) and timed with hipEvents
# recorded on the launch stream -- one event in front of every convolution launch and at every change of family
# (consecutive launches of the other families share an interval: ~2 000 of them are a few microseconds long, and an event
# pair around each would time the event markers, not the kernels).
#   * MFMA family (the dominant one): conv forward / dgrad / wgrad; algorithmic FLOP = 2*MAC from the launch arguments.
#   * HBM family "normalisation": statistics / finalize / apply passes (+ channel copies); algorithmic bytes = elements x
#     sizeof x (tensor operands the launch reads or writes), from the launch arguments.
#   * with --fourier: the build-defined Fourier amplitude mix (3 planes per call: x, the partner, y).
# ---------------------------------------------------------------------------------------------------------------------
CONV_CALLS = ("mrfp_conv_fwd", "mrfp_conv_fwd_gated", "mrfp_conv_fwd_wstats", "mrfp_conv_wgrad", "mrfp_conv_wgrad_grouped")
NORM_CALLS = ("mrfp_maxpool_affine_fwd", "mrfp_pool_norm_bwd_stats", "mrfp_pool_norm_bwd_apply", "mrfp_stats_fwd", "mrfp_stats_bwd", "mrfp_stats_bwd_mask", "mrfp_affine_fwd", "mrfp_affine_fwd_stats", "mrfp_affine_bwd",
              "mrfp_affine_fwd_relu_mask", "mrfp_affine_bwd_mask", "mrfp_bn_finalize", "mrfp_bn_bwd_finalize", "mrfp_in_finalize",
              "mrfp_in_bwd_finalize", "mrfp_np_finalize", "mrfp_np_bwd_finalize", "mrfp_mean_finalize", "mrfp_bn_eval_coef",
              "mrfp_copy_channels")
_TENSOR_ARGS = ("x", "res", "y", "dy", "dx", "dres", "src", "dst", "a", "b")


def _family(name):
    return "conv" if name in CONV_CALLS else "normalisation" if name in NORM_CALLS else \
        "fourier" if name == "mrfp_fourier_mix" else "other"


def _launch_work(name, d, esz):
    """(algorithmic FLOP, algorithmic bytes) of one launch from its named arguments."""
    if name in ("mrfp_conv_fwd", "mrfp_conv_fwd_gated", "mrfp_conv_fwd_wstats"):
        # ALGORITHMIC work: the logical channel counts (the zero channels of a padded buffer -- network input 3 -> 8, decoder
        # concatenation 304 -> 320 -- are not work); conv.py reports them through _lib.NOTE, absent = the physical counts
        M = d["B"] * d["Ho"] * d["Wo"]
        C, N = d.get("Clog", d["C"]), d.get("Nlog", d["N"])
        flop = 2.0 * M * N * d["R"] * d["S"] * C / float(d.get("sstride", 1) ** 2)
        byts = esz * (d["B"] * d["H"] * d["W"] * C + M * N + N * d["R"] * d["S"] * C
                      + (M * N if d.get("addend") else 0)) + (M * N / 8.0 if d.get("addend_mask") else 0)
        return flop, byts
    if name in ("mrfp_conv_wgrad", "mrfp_conv_wgrad_grouped"):
        M = d["B"] * d["Ho"] * d["Wo"]
        g = d.get("count", 1)
        return (g * 2.0 * M * d["N"] * d["R"] * d["S"] * d["Ctrue"],
                g * (esz * (d["B"] * d["H"] * d["W"] * d["Ctrue"] + M * d["N"]) + 4.0 * d["N"] * d["R"] * d["S"] * d["Ctrue"]))
    if name == "mrfp_fourier_mix":
        return 0.0, 3.0 * esz * d["B"] * d["H"] * d["W"] * d["C"]
    if name in NORM_CALLS:
        if "ws" in d and "nslab" in d:                                   # finalize kernels: the partial sums
            return 0.0, 8.0 * d["B"] * d["nslab"] * d["C"]
        elems = (d["npix"] * d["C"] if "npix" in d else
                 d["B"] * d.get("Ho", d.get("H", 0)) * d.get("Wo", d.get("W", 0)) * d["C"] if "B" in d else 0)
        passes = sum(1 for k in _TENSOR_ARGS if d.get(k)) + (1.0 / (8 * esz) if d.get("mask") else 0.0)
        return 0.0, esz * elems * passes
    try:
        return 0.0, float(_other_bytes(name, d, esz))
    except (TypeError, KeyError):          # (an argument name the header spells differently: no byte count, never a failed bench)
        return 0.0, 0.0


def _other_bytes(name, d, esz):
    """Algorithmic bytes of the launches outside the convolution / normalisation families (`other_ms_per_step` was
    one number): tensors read + written once, from the launch arguments."""
    g = d.get
    if name in ("mrfp_bilinear_fwd", "mrfp_bilinear_fwd_into"):          # read the low-resolution map (+ addend), write the output
        return esz * g("B") * g("C") * (g("Hi") * g("Wi") + g("Ho") * g("Wo") * (2 if g("addend") else 1))
    if name in ("mrfp_bilinear_bwd", "mrfp_bilinear_bwd_from"):
        return esz * g("B") * g("C") * (g("Hi") * g("Wi") + g("Ho") * g("Wo"))
    if name == "mrfp_upsample_ce_fwd":                                   # low-resolution class scores + int64 labels
        return esz * g("B") * g("Hi") * g("Wi") * g("ld") + 8 * g("B") * g("H") * g("W")
    if name == "mrfp_upsample_ce_bwd":
        return 2 * esz * g("B") * g("Hi") * g("Wi") * g("ld") + 8 * g("B") * g("H") * g("W") + esz * g("B") * g("Hi") * g("Wi") * g("Cd")
    if name == "mrfp_ce_fwd":
        return g("npix") * (esz * g("C") + 8)
    if name == "mrfp_ce_bwd":
        return g("npix") * (2 * esz * g("C") + 8)
    if name == "mrfp_sgd_step":                                          # p, g, m read; p, m written
        return 20 * g("n")
    if name == "mrfp_pack_weight":                                       # fp32 master read, two packs written
        return g("N") * g("C") * g("R") * g("S") * 4 + 2 * esz * g("Npad") * g("Cpad") * g("R") * g("S")
    if name == "mrfp_nchw_to_nhwc_pad":
        return g("B") * g("H") * g("W") * (4 * g("C") + esz * g("Cpad"))
    if name in ("mrfp_maxpool_fwd", "mrfp_maxpool_bwd"):
        return g("B") * g("C") * (esz * g("H") * g("W") + (esz + 1) * ((g("H") + 1) // 2) * ((g("W") + 1) // 2))
    if name == "mrfp_argmax_hist":
        return g("npix") * (esz * g("C") + 8 + 1)
    return 0.0          # (weight packs in one batched launch, philox draws, ...: bytes not derivable from the arguments)


def step_roofline(model, trainer, x, y, args):
    from mrfp_amd import _lib
    timer = HipTimer()
    st = torch.cuda.current_stream().cuda_stream
    esz = 4 if args.dtype == "f32" else 2
    marks = []          # (event, family of the launches that follow it, [(name, named args)] for conv launches)
    cur = [None]

    def hook(name, a):
        fam = _family(name)
        # (the few dozen launches of the "other" family get an event at every change of ENTRY POINT: its breakdown below)
        key = fam if fam != "other" else "other:" + name
        if fam == "conv" or key != cur[0] or args.dump_launches:
            e = timer.event()
            timer.record(e, st)
            marks.append([e, fam, []])
            cur[0] = key
        d = dict(zip(_lib.ARG_NAMES[name], a))
        if fam == "conv" and _lib.NOTE[0] is not None and name != "mrfp_conv_wgrad_grouped" and name != "mrfp_conv_wgrad":
            d["Clog"], d["Nlog"] = _lib.NOTE[0]
        _lib.NOTE[0] = None
        marks[-1][2].append((name, d))
    _lib.lib()
    _lib.NOTE[0] = None
    _lib.HOOK[0] = hook
    try:
        _eager_step(trainer, x, y)
    finally:
        _lib.HOOK[0] = None
    end = timer.event()
    timer.record(end, st)
    torch.cuda.synchronize()
    peak_f = PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS      # f16 and bf16 MFMA: same dense rate
    fam_ms, fam_bytes, fam_n = {}, {}, {}
    convs = []
    other = {}
    for i, (e, fam, calls) in enumerate(marks):
        ms = timer.elapsed_ms(e, marks[i + 1][0] if i + 1 < len(marks) else end)
        fam_ms[fam] = fam_ms.get(fam, 0.0) + ms
        fam_n[fam] = fam_n.get(fam, 0) + len(calls)
        if fam == "other" and calls:
            o = other.setdefault(calls[0][0], {"launches": 0, "ms": 0.0, "bytes": 0.0})
            o["launches"] += len(calls)
            o["ms"] += ms
            o["bytes"] += sum(_launch_work(n_, d_, esz)[1] for n_, d_ in calls)
        for name, d in calls:
            fl, by = _launch_work(name, d, esz)
            fam_bytes[fam] = fam_bytes.get(fam, 0.0) + by
            if fam == "conv":
                wg = name in ("mrfp_conv_wgrad", "mrfp_conv_wgrad_grouped")
                first = _lib.ARG_NAMES[name].index("B")
                shape = [int(d[k]) for k in _lib.ARG_NAMES[name][first:first + 15]]
                if name == "mrfp_conv_fwd_wstats":          # (no sstride argument: 14 geometry values + sstride 1)
                    shape = shape[:14] + [1]
                convs.append({"name": "mrfp_conv_wgrad" if wg else "mrfp_conv_fwd", "ms": ms, "flop": fl, "bytes": by,
                              "group": int(d.get("count", 1)), "args": shape})
    if args.dump_launches:      # an event in front of EVERY launch (inflates the few-microsecond kernels: a diagnostic, not the bench line)
        rows = []
        for i, (e, fam, calls) in enumerate(marks):
            ms = timer.elapsed_ms(e, marks[i + 1][0] if i + 1 < len(marks) else end)
            name, d = calls[0]
            fl, by = _launch_work(name, d, esz)
            rows.append({"name": name, "family": fam, "ms": ms, "mbytes": by / 1e6, "gflop": fl / 1e9,
                         "args": {k: (v if isinstance(v, (int, float)) and abs(v) < 10 ** 7 else bool(v)) for k, v in d.items()}})
        with open(args.dump_launches, "w") as f:
            json.dump(rows, f)
    tot_ms, tot_f = sum(c["ms"] for c in convs), sum(c["flop"] for c in convs)
    if args.dump_convs:
        with open(args.dump_convs, "w") as f:
            json.dump([{"name": c["name"], "args": c["args"], "group": c["group"], "ms": c["ms"], "gflop": c["flop"] / 1e9, "mbytes": c["bytes"] / 1e6,
                        "tflops": c["flop"] / (c["ms"] * 1e-3) / 1e12 if c["ms"] > 0 else 0} for c in convs], f)
    ach = tot_f / (tot_ms * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": "conv_igemm_kernel+conv3x3_c64_kernel+conv1x1_bstat_kernel+conv1x1_longk*_kernel+conv_wgrad_kernel+conv_wg3_kernel+conv_wg1_kernel (all %d conv launches of one step, %d weight-gradient problems in them)"
            % (len(convs), sum(c["group"] for c in convs if c["name"] == "mrfp_conv_wgrad")),
            "achieved": round(ach, 2), "peak": peak_f, "unit": "TFLOP/s", "frac": round(ach / peak_f, 4),
            "traffic": None, "conv_ms_per_step": round(tot_ms, 3), "conv_tflop_per_step": round(tot_f / 1e12, 3)}
    # every conv launch against ITS OWN bound: max(FLOP / MFMA peak, algorithmic bytes / HBM peak)
    by_class = {}
    for c in convs:
        t_f, t_b = c["flop"] / (peak_f * 1e12), c["bytes"] / (PEAK_HBM_GBS * 1e9)
        k = ("mfma_bound" if t_f >= t_b else "hbm_bound") + ("_wgrad" if c["name"] == "mrfp_conv_wgrad" else "_fwd_dgrad")
        g = by_class.setdefault(k, {"launches": 0, "ms": 0.0, "tflop": 0.0, "gbytes": 0.0, "bound_ms": 0.0})
        g["launches"] += 1
        g["ms"] += c["ms"]
        g["tflop"] += c["flop"] / 1e12
        g["gbytes"] += c["bytes"] / 1e9
        g["bound_ms"] += 1e3 * max(t_f, t_b)
    for g in by_class.values():
        g["frac_of_own_bound"] = round(g["bound_ms"] / g["ms"], 4) if g["ms"] > 0 else None
        for k in ("ms", "tflop", "gbytes", "bound_ms"):
            g[k] = round(g[k], 3)
    roof["by_class"] = by_class
    att = attainable(peak_f)
    if att is not None:
        roof["attainable"] = att
        roof["frac_of_attainable"] = round(ach / att["mfma_tflops"], 4)
    pmc = measured_traffic(args)
    roof.update(pmc.get("conv", {}))

    def hbm_entry(fam, what):
        ms, by = fam_ms.get(fam, 0.0), fam_bytes.get(fam, 0.0)
        if ms <= 0:
            return None
        gbs = by / (ms * 1e-3) / 1e9
        e = {"bound": "hbm", "family": fam, "kernel": what, "launches": fam_n.get(fam, 0), "ms": round(ms, 3),
             "bytes": round(by), "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
             "traffic": None}
        if att is not None:
            e["frac_of_attainable"] = round(gbs / att["hbm_gbs"], 4)
        e.update(pmc.get(fam, {}))
        return e
    roof["hbm"] = hbm_entry("normalisation", "stats_kernel + *_finalize_kernel + affine_fwd/bwd_kernel + copy_channels_kernel")
    if args.fourier:
        roof["fourier"] = hbm_entry("fourier", "fft_rows / fft_cols_mix / dft_rows_inv kernels of mrfp_fourier_mix (3 planes per call)")
    roof["other_ms_per_step"] = round(fam_ms.get("other", 0.0), 3)
    # ... and what is in it, per entry point (each interval also holds the ATen fills / draws issued between two launches of this library)
    roof["other"] = {k: {"launches": v["launches"], "ms": round(v["ms"], 3), "bytes": round(v["bytes"]),
                         "achieved_gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 and v["bytes"] > 0 else None,
                         "bound": "hbm" if v["bytes"] > 0 else "launch"}
                     for k, v in sorted(other.items(), key=lambda kv: -kv[1]["ms"])}
    # the re-pack launch follows the optimizer step, which waits for the weight-gradient stream: its INTERVAL on the main stream is mostly
    # that wait (the kernel itself: ~0.13 ms, profiles/r05_bench_kernel_stats.md) -- say so next to the number
    if "mrfp_pack_weights_batched" in roof["other"]:
        roof["other"]["mrfp_pack_weights_batched"]["note"] = "interval on the main stream: includes waiting for the weight-gradient stream behind the SGD step; kernel time ~0.13 ms"
    return roof


def attainable(peak_f):
    """What the part can reach under this load, next to the nominal peaks: the in-kernel clock of the convolution kernels (measured
    with a diagnostic build that stamps s_memtime / s_memrealtime around every workgroup's main loop, tools/clock_stamp.py ->
    the newest profiles/r*_clock.json; MI355X_MICROARCH.md, DVFS give-back: the chip holds ~1.8 GHz, not 2.4, under an MFMA-dense
    load), the MFMA rate at that clock, and the HBM bandwidth a streaming kernel achieves (the guide's ~6.3 TB/s of the 8 TB/s
    spec).  Counters and stamps cannot be collected from inside this process: the file names what it was measured on."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_clock.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            c = json.load(f)
    except (OSError, ValueError):
        return None
    ghz = c.get("clock_ghz_median")
    if not ghz:
        return None
    return {"clock_ghz": ghz, "nominal_clock_ghz": 2.4, "mfma_tflops": round(peak_f * ghz / 2.4, 1), "hbm_gbs": 6300.0,
            "source": "profiles/" + os.path.basename(files[-1])}


def running_commit():
    c = os.environ.get("MRFP_COMMIT")
    if c:
        return c
    try:
        import subprocess
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except Exception:
        return None


def measured_traffic(args):
    """HBM bytes per step by kernel family from the PMC counters (FETCH_SIZE / WRITE_SIZE passes collected by
    tools/measure_traffic.sh exactly as MI355X_MICROARCH.md prescribes, in their own rocprofv3 runs) -- read from the newest
    committed profiles/r*_traffic.json, which names the commit and the workload it was measured on.  Counters cannot be
    collected from inside this process.  The figure is reported as `traffic` only when that file describes THIS workload AND
    the kernel sources are the ones it was measured on (`source_sha16` = mrfp_amd._lib.source_hash(): kernels, C header,
    operator layer; or the same commit); kernels may have changed since otherwise, and it goes under `traffic_reference` with
    the commit next to it."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files or args.fourier:
        return {}
    try:
        with open(files[-1]) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return {}
    w = t.get("workload", {})
    if (w.get("trunk"), w.get("size"), w.get("width"), w.get("batch"), w.get("dtype")) != \
            (args.trunk, args.size, args.width or args.size, args.batch, args.dtype):
        return {}
    from mrfp_amd import _lib
    here = running_commit()
    same = (t.get("source_sha16") == _lib.source_hash()) or \
        (bool(here) and bool(t.get("commit")) and (here.startswith(t["commit"]) or t["commit"].startswith(here)))
    key = "traffic" if same else "traffic_reference"
    out = {}
    for fam in ("conv", "normalisation"):
        hb = t.get("hbm_bytes_per_step", {}).get(fam)
        if not hb:
            continue
        e = {key: hb["read"] + hb["write"], "traffic_unit": "bytes/step (%s kernel family, PMC FETCH_SIZE x2 + WRITE_SIZE)" % fam,
             "traffic_measured_at_commit": t.get("commit"), "traffic_source": "profiles/" + os.path.basename(files[-1])}
        if fam == "conv":
            e["traffic_algorithmic"] = t.get("conv_family_algorithmic_bytes_per_step")
        out[fam] = e
    return out


if __name__ == "__main__":
    main()
