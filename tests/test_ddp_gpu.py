"""Data-parallel training step on the real GPU path with 2 processes (both on cuda:0, gloo transport -- RCCL
refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  Exercises what changes at world size > 1:
bucket hooks fired by autograd AND by the direct gradient sinks of the HIP backward kernels, the side stream,
the 1/world scale folded into the fused SGD kernel.  Checks: both ranks end with identical parameters, and the
synchronised gradient equals the mean of the two local gradients."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(seed=0):
    sys.path.insert(0, ROOT)
    from mrfp_amd.config import cfg
    from mrfp_amd.network import Resnet
    cfg.MODEL.CONV_BACKEND, cfg.MODEL.ACT_DTYPE = "hip", torch.float32
    torch.manual_seed(seed)
    net = Resnet.resnet18(pretrained=False, wt_layer=[0, 0, 4, 4, 0, 0, 0])
    del net.fc, net.avgpool
    return net.to("cuda:0").train()


class _Wrap(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, x, y, training=True):
        return self.net(x).float().pow(2).mean()


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from mrfp_amd.harness import Trainer
    model = _Wrap(_build(0))
    tr = Trainer(model, lr=1e-3, bucket_mb=4.0)
    assert tr.sync.enabled and len(tr.sync.buckets) >= 3
    g = torch.Generator().manual_seed(100 + rank)
    x = (torch.rand(2, 3, 64, 64, generator=g) * 255).cuda()
    # step 1 by hand to look at the synchronised gradient
    tr.opt.zero_grad()
    tr.sync.begin()
    model(x, None).backward()
    scale = tr.sync.finish()
    torch.cuda.synchronize()
    torch.save((tr.opt.flat_g.cpu() * scale, x.cpu()), os.path.join(outdir, "g%d.pt" % rank))
    tr.opt.step(scale)
    loss = tr.step(x, None)            # a full second step through Trainer
    torch.cuda.synchronize()
    torch.save((tr.opt.flat_p.cpu(), float(loss)), os.path.join(outdir, "p%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_sync_on_gpu(tmp_path):
    world, port = 2, 29733
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    g = [torch.load(os.path.join(str(tmp_path), "g%d.pt" % r)) for r in range(world)]
    p = [torch.load(os.path.join(str(tmp_path), "p%d.pt" % r)) for r in range(world)]
    assert torch.equal(g[0][0], g[1][0])                       # same averaged gradient on both ranks
    assert torch.equal(p[0][0], p[1][0])                       # replicas stay bit-identical
    # the averaged gradient == mean of the two local gradients (recomputed here, single process)
    sys.path.insert(0, ROOT)
    from mrfp_amd.harness import FlatArena
    model = _Wrap(_build(0))
    arena = FlatArena(model)
    ref = torch.zeros_like(arena.flat_g)
    for r in range(world):
        arena.zero_grad()
        model(g[r][1].cuda(), None).backward()
        ref += arena.flat_g / world
    torch.cuda.synchronize()
    torch.testing.assert_close(g[0][0], ref.cpu(), rtol=1e-4, atol=1e-6)
