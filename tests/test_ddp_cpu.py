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


def test_bench_gpus_flag_builds_a_child_launch(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE) must start N ranks through torch.distributed.run as a CHILD
    process on the loopback address and hand back its exit code (the GPU end-to-end form is tests/test_ddp_gpu.py)."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[-7]) == "bench.py" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
