import os, sys, torch, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
def worker(rank, world, port):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_ddp_gpu import _build, _Wrap
    from mrfp_amd.harness import Trainer
    model = _Wrap(_build(0))
    tr = Trainer(model, lr=1e-3, bucket_mb=4.0)
    order = []
    orig = tr.sync._launch
    def spy(b):
        lo, hi, mem = tr.sync.buckets[b]
        torch.cuda.synchronize()
        order.append((b, float(tr.opt.flat_g[lo:hi].double().norm())))
        orig(b)
    tr.sync._launch = spy
    g = torch.Generator().manual_seed(100 + rank)
    x = (torch.rand(2, 3, 64, 64, generator=g) * 255).cuda()
    tr.opt.zero_grad(); tr.sync.begin()
    model(x, None).backward()
    pend = list(tr.sync.pending)
    scale = tr.sync.finish()
    torch.cuda.synchronize()
    after = [float(tr.opt.flat_g[lo:hi].double().norm()) for lo, hi, _ in tr.sync.buckets]
    print("rank", rank, "buckets", [(lo, hi, len(m)) for lo, hi, m in tr.sync.buckets], "\n   pending after bwd", pend, "\n   launch order (bucket, local norm)", order, "\n   after sync norms", after, flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, 29741)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]
