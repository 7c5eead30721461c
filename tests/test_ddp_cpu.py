"""Data-parallel gradient exchange (mrfp_amd/harness.py::GradSync) on 2 CPU processes with gloo: bucket
partition of the flat arena, hook-triggered all-reduce, 1/world scaling.  No GPU involved: the arenas are
CPU tensors and the update rule is the host formula of the fused kernel."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(7, 33)
        self.b = torch.nn.Linear(33, 65)
        self.c = torch.nn.Linear(65, 3)
        self.frozen = torch.nn.Linear(3, 3).requires_grad_(False)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x))))).sum()


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mrfp_amd.harness import FlatArena, GradSync
    torch.manual_seed(0)
    net = TinyNet()
    arena = FlatArena(net)
    assert arena.n % 4 == 0 and all(o % 4 == 0 for o in arena.offsets)
    sync = GradSync(arena, bucket_mb=2400 * 4 / (1 << 20))       # ~2400 floats per bucket -> several buckets
    assert len(sync.buckets) >= 2
    covered = sorted((lo, hi) for lo, hi, _ in sync.buckets)
    assert covered[0][0] == 0 and covered[-1][1] == arena.n
    assert all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 7)
    for _ in range(2):                                            # two steps: hooks must re-arm
        arena.zero_grad()
        sync.begin()
        net(x).backward()
        scale = sync.finish()
    torch.save((rank, arena.flat_g.clone() * scale, x), os.path.join(outdir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_mean_gradient(tmp_path):
    world, port = 2, 29611
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    out = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    sys.path.insert(0, ROOT)
    from mrfp_amd.harness import FlatArena
    torch.manual_seed(0)
    net = TinyNet()
    arena = FlatArena(net)
    ref = torch.zeros_like(arena.flat_g)
    for _, _, x in out:
        arena.zero_grad()
        net(x).backward()
        ref += arena.flat_g / world
    for _, g, _ in out:                                            # every rank holds the mean gradient
        torch.testing.assert_close(g, ref, rtol=1e-6, atol=1e-6)


class BranchNet(torch.nn.Module):
    """A net whose middle branch is skipped when `use_mid` is False: its parameters then get NO gradient on that rank
    (what a rank-local perturbation toggle would do to a branch with parameters)."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(7, 40)
        self.mid = torch.nn.Linear(40, 40)
        self.c = torch.nn.Linear(40, 3)

    def forward(self, x, use_mid=True):
        t = torch.relu(self.a(x))
        if use_mid:
            t = t + torch.relu(self.mid(t))
        return self.c(t).sum()


def _worker_hardening(rank, world, port, outdir):
    """(1) ranks construct DIFFERENT replicas -> harness.sync_replicas makes them rank 0's, and gives every rank the same
    private toggle stream; (2) ranks take different branches, so one bucket gets no gradient on rank 1: the collectives
    still pair up (fixed launch order) and the result is the mean gradient with zeros for the skipped tensors."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mrfp_amd.deepv3 import ReferenceRandom
    from mrfp_amd.harness import FlatArena, GradSync, sync_replicas
    torch.manual_seed(1000 + rank)                       # different initial weights per rank
    net = BranchNet()
    net.register_buffer("stat", torch.full((3,), float(rank)))
    net.rng = ReferenceRandom()
    arena = FlatArena(net)
    before = arena.flat_p.clone()
    sync_replicas(net, arena)
    toggles = [net.rng.toggles() for _ in range(3)]
    alpha, beta = net.rng.np_noise("np1", 4, 8, torch.device("cpu"))      # the NP+ normals of reference deepv3.py:274-275
    sync = GradSync(arena, bucket_mb=100 * 4 / (1 << 20))    # buckets: {c.*}, {mid.*}, {a.*}
    assert len(sync.buckets) >= 3
    order = []
    real_launch = sync._launch
    sync._launch = lambda b: (order.append(b), real_launch(b))[1]
    torch.manual_seed(200 + rank)
    x = torch.randn(5, 7)
    arena.zero_grad()
    sync.begin()
    net(x, use_mid=(rank == 0)).backward()
    scale = sync.finish()
    torch.save({"before": before, "after": arena.flat_p.clone(), "stat": net.stat.clone(), "toggles": toggles, "alpha": alpha, "beta": beta,
                "g": arena.flat_g.clone() * scale, "x": x, "order": order}, os.path.join(outdir, "h%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sync_fixed_bucket_order_and_missing_gradients(tmp_path):
    world, port = 2, 29617
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_hardening, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    out = [torch.load(os.path.join(str(tmp_path), "h%d.pt" % r)) for r in range(world)]
    assert not torch.equal(out[0]["before"], out[1]["before"])            # the replicas really started different
    assert torch.equal(out[0]["after"], out[1]["after"]) and torch.equal(out[0]["after"], out[0]["before"])
    assert torch.equal(out[1]["stat"], torch.zeros(3))                    # buffers follow rank 0 too
    assert out[0]["toggles"] == out[1]["toggles"]                         # same perturbation branches on every rank
    # ... but each rank draws its OWN perturbation (torch generators seeded base + rank)
    assert not torch.equal(out[0]["alpha"], out[1]["alpha"]) and not torch.equal(out[0]["beta"], out[1]["beta"])
    assert out[0]["order"] == out[1]["order"] == sorted(out[0]["order"])  # collectives issued in index order
    sys.path.insert(0, ROOT)
    from mrfp_amd.harness import FlatArena
    net = BranchNet()
    arena = FlatArena(net)
    arena.flat_p.copy_(out[0]["after"])
    ref = torch.zeros_like(arena.flat_g)
    for r in range(world):
        arena.zero_grad()
        net(out[r]["x"], use_mid=(r == 0)).backward()
        ref += arena.flat_g / world
    for r in range(world):
        torch.testing.assert_close(out[r]["g"], ref, rtol=1e-6, atol=1e-6)


def test_flat_sgd_state_dict_is_torch_sgd_layout():
    """FlatSGD.state_dict() / load_state_dict() speak torch.optim.SGD's layout (the 'optimizer' entry of reference
    main.py:867): indices over ALL model parameters (frozen ones without state), momentum_buffer per tensor, current
    and initial lr; a state dict written by torch.optim.SGD + LambdaLR loads back into the arena (host logic only)."""
    sys.path.insert(0, ROOT)
    from mrfp_amd.harness import FlatSGD, poly_lr_factor
    torch.manual_seed(3)
    net = TinyNet()
    ref_opt = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(ref_opt, lr_lambda=poly_lr_factor)
    for i in range(3):
        ref_opt.zero_grad()
        net(torch.randn(4, 7)).backward()
        ref_opt.step()
        sched.step()
    tsd = ref_opt.state_dict()
    opt = FlatSGD(net, lr=123.0)                       # host-side construction works without a GPU; step() does not
    opt.load_state_dict(tsd)
    assert opt.it == 3 and abs(opt.base_lr - 1e-2) < 1e-15 and opt.has_momentum
    assert abs(opt.lr - tsd["param_groups"][0]["lr"]) < 1e-12
    mine = opt.state_dict()
    assert mine["param_groups"][0]["params"] == tsd["param_groups"][0]["params"] == list(range(8))
    assert sorted(mine["state"].keys()) == sorted(tsd["state"].keys()) == list(range(6))   # the frozen Linear has none
    for k, st in tsd["state"].items():
        assert torch.equal(mine["state"][k]["momentum_buffer"], st["momentum_buffer"])
    for key in ("momentum", "dampening", "weight_decay", "nesterov", "initial_lr"):
        assert mine["param_groups"][0][key] == tsd["param_groups"][0][key], key
    ref_opt2 = torch.optim.SGD(net.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
    ref_opt2.load_state_dict({k: v for k, v in mine.items() if k != "mrfp_iteration"})   # and torch accepts ours
    # torch keeps no state for a parameter that never got a gradient: such a checkpoint loads with zero momentum there
    part = {"state": {k: v for k, v in tsd["state"].items() if k >= 2}, "param_groups": tsd["param_groups"]}
    opt3 = FlatSGD(net)
    opt3.load_state_dict(part)
    assert opt3.has_momentum and opt3.missing_state == 2
    o0, n0 = opt3.offsets[0], opt3.params[0].numel()
    assert float(opt3.flat_m[o0:o0 + n0].abs().max()) == 0.0 and float(opt3.flat_m.abs().max()) > 0.0
    bad = {"state": {0: {"momentum_buffer": torch.zeros(5)}}, "param_groups": tsd["param_groups"]}
    with pytest.raises(Exception):
        FlatSGD(net).load_state_dict(bad)               # a shape mismatch is still an error
    fresh = FlatSGD(TinyNet())
    assert fresh.state_dict()["state"] == {} and not fresh.has_momentum
    with pytest.raises(Exception):
        fresh.step()                                    # no CPU fallback for the update itself


def test_poly_lr_and_sgd_rule_host_formula():
    from mrfp_amd.harness import poly_lr_factor
    assert poly_lr_factor(0) == 1.0
    assert abs(poly_lr_factor(20000) - 0.5 ** 0.9) < 1e-12         # reference main.py:832-839
    assert poly_lr_factor(40000) == 0.0


class _FakeChild:
    def __init__(self, out, err, rc):
        self.stdout, self.stderr, self._rc = iter(out), iter(err), rc

    def wait(self):
        return self._rc


def test_bench_gpus_flag_builds_a_child_launch(monkeypatch, capsys):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE) must start N ranks through torch.distributed.run as a CHILD
    process on the loopback address, relay its output and hand back its exit code (the GPU end-to-end form is
    tests/test_ddp_gpu.py)."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_popen(cmd, env=None, cwd=None, **kw):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return _FakeChild(['{"metric": "x"}\n'], ["some warning\n"], 7)
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MRFP_BENCH_PORT", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[-7]) == "bench.py" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert int(seen["env"]["OMP_NUM_THREADS"]) >= 1
    assert '{"metric": "x"}' in capsys.readouterr().out                  # the child's JSON line is relayed


def test_bench_launcher_retries_once_when_the_probed_port_was_taken(monkeypatch):
    """The rendezvous port is probed by bind-and-release; if another process takes it before torch.distributed.run binds it the
    child dies with EADDRINUSE: ONE more attempt in a FRESH child on a fresh port (never a re-exec), none after a result line or
    any other failure."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    launches = []
    script = [(["\n"], ["RuntimeError: The server socket has failed to listen on any local network address. "
                        "port: 1, useIpv6: 0, code: -98, name: EADDRINUSE, message: address already in use\n"], 1),
              (['{"metric": "x"}\n'], [], 0)]

    def fake_popen(cmd, env=None, cwd=None, **kw):
        launches.append(cmd[cmd.index("--master-port") + 1])
        out, err, rc = script[len(launches) - 1]
        return _FakeChild(out, err, rc)
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.delenv("MRFP_BENCH_PORT", raising=False)
    ports = iter(["41001", "41002", "41003"])
    monkeypatch.setattr(bench, "_free_port", lambda: next(ports))

    class A:
        gpus = 8
    assert bench.launch_ranks(A(), argv=["--gpus", "8"]) == 0
    assert launches == ["41001", "41002"]
    # any other failure: no second attempt
    del launches[:]
    script[0] = (["\n"], ["Traceback ...\n"], 3)
    assert bench.launch_ranks(A(), argv=["--gpus", "8"]) == 3 and launches == ["41003"]


_STUB = '''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
from mrfp_amd.harness import FlatArena, GradSync
torch.manual_seed(0)
net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.ReLU(), torch.nn.Linear(33, 5))
arena = FlatArena(net)
sync = GradSync(arena, bucket_mb=64 * 4 / (1 << 20))
arena.zero_grad(); sync.begin()
torch.manual_seed(100 + dist.get_rank())
net(torch.randn(4, 7)).sum().backward()
scale = sync.finish()
ones = torch.ones(1); dist.all_reduce(ones)
g = arena.flat_g * scale
ref = [torch.zeros_like(g) for _ in range(dist.get_world_size())]
dist.all_gather(ref, g)
ok = all(torch.equal(ref[0], r) for r in ref)
dist.barrier()
if dist.get_rank() == 0:
    print('{"ranks_seen": %%d, "equal": %%s, "omp": "%%s"}' %% (int(ones.item()), "true" if ok else "false", os.environ.get("OMP_NUM_THREADS")))
dist.destroy_process_group()
'''


def test_eight_rank_launch_rehearsal_on_cpu(tmp_path, capfd):
    """The driver's 8-GPU run is the first time eight ranks meet (this pool gives the builder one GPU and at most six processes
    on it): rehearse what does not need the card -- bench.launch_ranks itself starting EIGHT ranks through torch.distributed.run
    on the loopback address, the rendezvous, the bucketed gradient exchange over the data-path group (gloo, CPU arenas) and the
    `ranks_seen` all-reduce -- with a stub rank program."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB % ROOT)

    class A:
        gpus = 8
    env_before = os.environ.get("MRFP_BENCH_PORT")
    assert env_before is None
    rc = bench.launch_ranks(A(), script=str(stub), argv=[])
    assert rc == 0
    lines = [ln for ln in capfd.readouterr().out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["ranks_seen"] == 8 and out["equal"] is True and int(out["omp"]) >= 1


def _worker_deferred(outdir):
    """One rank, MRFP_FORCE_SYNC=1 (the bucket machinery runs at world size 1): a backward node that only QUEUES its weight
    gradient (what conv._queue_wgrad does for the grouped launches) must hold its bucket back until ops.notify_grad reports
    the launch -- autograd's post-accumulate hook fires for the parameter as soon as the node has run."""
    sys.path.insert(0, ROOT)
    os.environ["MRFP_FORCE_SYNC"] = "1"
    dist.init_process_group("gloo", init_method="file://" + os.path.join(outdir, "store"), rank=0, world_size=1)
    from mrfp_amd import ops
    from mrfp_amd.harness import FlatArena, GradSync
    queue = []

    class Deferred(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            ctx.w = w
            return x @ w.t()

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            ops.GRAD_DEFERRED.add(id(ctx.w))
            queue.append((ctx.w, dy.t() @ x))
            return dy @ w, None

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(7, 33)
            self.w = torch.nn.Parameter(torch.randn(40, 33))
            self.c = torch.nn.Linear(40, 3)

        def forward(self, x):
            return self.c(Deferred.apply(torch.relu(self.a(x)), self.w)).sum()

    torch.manual_seed(0)
    net = Net()
    arena = FlatArena(net)
    sync = GradSync(arena, bucket_mb=100 * 4 / (1 << 20))       # buckets (from the end): {c.*}, {w}, {a.*}
    launched = []
    real = sync._launch
    sync._launch = lambda b: (launched.append(b), real(b))[1]
    w_index = [i for i, p in enumerate(arena.params) if p is net.w][0]
    wb = sync.bucket_of[w_index]
    arena.zero_grad()
    sync.begin()
    net(torch.randn(5, 7)).backward()
    held = wb not in launched and all(b < wb for b in launched)        # w's bucket and every later one wait for the queued launch
    for w, g in queue:                                                  # the "flush": write the gradient, then report it
        w.grad.copy_(g)
        ops.GRAD_DEFERRED.discard(id(w))
        ops.notify_grad(w)
    after = list(launched)
    sync.finish()
    torch.save({"held": held, "after": after, "n": len(sync.buckets), "wb": wb, "gw": net.w.grad.clone(), "ref": queue[0][1]},
               os.path.join(outdir, "d.pt"))
    dist.destroy_process_group()


def test_deferred_weight_gradient_holds_its_bucket_back(tmp_path):
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_worker_deferred, args=(str(tmp_path),))
    p.start()
    p.join(timeout=120)
    assert p.exitcode == 0
    out = torch.load(os.path.join(str(tmp_path), "d.pt"))
    assert out["n"] >= 3 and out["held"]
    assert out["after"] == list(range(out["n"]))                         # the notification released every bucket, in index order
    torch.testing.assert_close(out["gw"], out["ref"])


def test_more_ranks_than_devices_is_one_clear_line_and_a_nonzero_exit():
    """a rank whose LOCAL_RANK has no device says so and exits 3 before anything touches a GPU or a
    rendezvous (the check runs in the RANK, not in the launching parent).  This container has no GPU: every local rank is one
    too many."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    env.pop("MRFP_BENCH_SHARE_GPU", None)
    import torch
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a host with fewer than two GPUs")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-800:])
    assert "exposes only" in r.stderr and "--gpus 2" in r.stderr and not r.stdout.strip()
