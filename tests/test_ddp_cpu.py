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


def test_poly_lr_and_sgd_rule_host_formula():
    from mrfp_amd.harness import poly_lr_factor
    assert poly_lr_factor(0) == 1.0
    assert abs(poly_lr_factor(20000) - 0.5 ** 0.9) < 1e-12         # reference main.py:832-839
    assert poly_lr_factor(40000) == 0.0
