#!/usr/bin/env python3
"""Golden vectors for the CROSS-RANK path of switchable whitening -- runs ONLY in the build container.

The reference's `SyncSwitchWhiten2d` (network/sync_switchwhiten.py:59-223) with its `SyncMeanCov` Function (:9-56: two
forward and two backward `dist.all_reduce`s) is run here by TWO processes under a local gloo group, each rank on its own
seeded input; forward, backward, running statistics and the eval pass of each rank are stored in
tests/golden/syncsw.npz.  tests/test_ddp_gpu.py runs the HIP implementation on two ranks (gloo transport, both on the
one GPU of the box) against these numbers.
"""
import os
import sys
import tempfile
import warnings

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
WORLD, C, N, H, W = 2, 32, 2, 10, 12


def case(rank):
    """Inputs of rank `rank` (shared with the test through tests/golden_common.py-style seeds)."""
    g = torch.Generator().manual_seed(70 + rank)
    x = torch.randn(N, C, H, W, generator=g) * (1.5 + rank) + 0.3 * rank
    gy = torch.randn(N, C, H, W, generator=g)
    return x, gy


def params():
    g = torch.Generator().manual_seed(7)
    return {"weight": torch.rand(C, generator=g) + 0.5, "bias": torch.randn(C, generator=g) * 0.1,
            "sw_mean_weight": torch.tensor([0.3, -0.2]), "sw_var_weight": torch.tensor([-0.1, 0.4])}


def worker(rank, store, outdir):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, REF)
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=WORLD)
    from network import sync_switchwhiten as ref
    sw = ref.SyncSwitchWhiten2d(C, num_pergroup=16, sw_type=2, T=5, tie_weight=False, eps=1e-5, momentum=0.99, affine=True)
    with torch.no_grad():
        for k, v in params().items():
            getattr(sw, k).copy_(v)
    sw.train()
    x, gy = case(rank)
    x.requires_grad_(True)
    y = sw(x)
    y.backward(gy)
    out = {"y": y.detach().numpy(), "gx": x.grad.numpy(), "g_weight": sw.weight.grad.numpy(), "g_bias": sw.bias.grad.numpy(),
           "g_mean_w": sw.sw_mean_weight.grad.numpy(), "g_var_w": sw.sw_var_weight.grad.numpy(),
           "running_mean": sw.running_mean.numpy().copy(), "running_cov": sw.running_cov.numpy().copy()}
    sw.eval()
    with torch.no_grad():
        out["y_eval"] = sw(x.detach()).numpy()
    np.savez(os.path.join(outdir, "r%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    with tempfile.TemporaryDirectory() as d:
        store = os.path.join(d, "store")
        mp.spawn(worker, args=(store, d), nprocs=WORLD, join=True)
        out = {}
        for r in range(WORLD):
            z = np.load(os.path.join(d, "r%d.npz" % r))
            for k in z.files:
                out["r%d_%s" % (r, k)] = z[k]
    # the two ranks must agree on the synchronised statistics
    assert np.array_equal(out["r0_running_mean"], out["r1_running_mean"])
    assert np.array_equal(out["r0_running_cov"], out["r1_running_cov"])
    np.savez_compressed(os.path.join(HERE, "syncsw.npz"), **out)
    print("wrote syncsw.npz:", {k: v.shape for k, v in out.items() if k.startswith("r0_")})


if __name__ == "__main__":
    main()
