"""HIP kernels (through the C ABI / mrfp_amd.ops) against the plain PyTorch fp32 restatement of the
same reference arithmetic, computed on the CPU (oracle/mrfp_oracle.py helpers + torch.nn.functional).

fp32 activations: tolerance 2e-5 relative to the tensor's max (fp32 summation-order noise);
bf16 activations: 2e-2 (bf16 has 8 bits of mantissa; statistics are still accumulated in fp32).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import mrfp_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def ops():
    from mrfp_amd import ops as o
    return o


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2.5e-2


def rnd(*shape, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale + shift


def dev(x, dtype):
    return x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)


DTYPES = [torch.float32, torch.bfloat16]
SHAPES = [(2, 64, 17, 23), (3, 48, 9, 31), (2, 256, 12, 12), (2, 2048, 5, 7), (1, 19, 13, 11)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_batch_norm_act(dtype, shape, relu, res):
    o = ops()
    B, C, H, W = shape
    x = rnd(*shape, seed=1, scale=3.0, shift=1.5)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    r = rnd(*shape, seed=2) if res else None
    if r is not None and dtype == torch.bfloat16:
        r = r.bfloat16().float()
    w, b = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    rm, rv = torch.randn(C) * 0.1, torch.rand(C) + 0.5
    gy = rnd(*shape, seed=3)
    # CPU fp32 restatement
    xc = x.clone().requires_grad_(True)
    wc, bc = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rc = r.clone().requires_grad_(True) if res else None
    rm_c, rv_c = rm.clone(), rv.clone()
    yc = F.batch_norm(xc, rm_c, rv_c, wc, bc, True, 0.1, 1e-5)
    if res:
        yc = yc + rc
    if relu:
        yc = F.relu(yc)
    yc.backward(gy)
    # HIP
    xd = dev(x, dtype)
    rd = dev(r, dtype) if res else None
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    yd = o.batch_norm_act(xd, wd, bd, rm_d, rv_d, training=True, relu=relu, res=rd)
    yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    t = tol(dtype)
    assert relerr(yd, yc) < t
    assert relerr(rm_d, rm_c) < 1e-5 and relerr(rv_d, rv_c) < 1e-5 + (0 if dtype == torch.float32 else 1e-2)
    assert relerr(xd.grad, xc.grad) < 10 * t
    assert relerr(wd.grad, wc.grad) < 10 * t and relerr(bd.grad, bc.grad) < 10 * t
    if res:
        assert relerr(rd.grad, rc.grad) < t
    # eval mode coefficients
    ye = o.batch_norm_act(xd.detach(), wd.detach(), bd.detach(), rm_d, rv_d, training=False, relu=relu)
    yec = F.batch_norm(x, rm_c, rv_c, w, b, False, 0.1, 1e-5)
    assert relerr(ye, F.relu(yec) if relu else yec) < t
    # eval mode backward (module in eval(), training=True argument: reference deepv3.py keys the two separately)
    xe = dev(x, dtype)
    we, be = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    ye = o.batch_norm_act(xe, we, be, rm_d, rv_d, training=False, relu=relu)
    ye.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    xc2 = x.clone().requires_grad_(True)
    wc2, bc2 = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yc2 = F.batch_norm(xc2, rm_c, rv_c, wc2, bc2, False, 0.1, 1e-5)
    (F.relu(yc2) if relu else yc2).backward(gy)
    assert relerr(xe.grad, xc2.grad) < 10 * t
    assert relerr(we.grad, wc2.grad) < 10 * t and relerr(be.grad, bc2.grad) < 10 * t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", SHAPES[:4])
@pytest.mark.parametrize("relu,affine", [(False, True), (True, True), (False, False)])
def test_instance_norm_act(dtype, shape, relu, affine):
    o = ops()
    B, C, H, W = shape
    x = rnd(*shape, seed=4, scale=50.0, shift=120.0)       # stem-like magnitudes (inputs are 0..255)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    w = (torch.rand(C) + 0.5) if affine else None
    b = (torch.randn(C) * 0.1) if affine else None
    gy = rnd(*shape, seed=5)
    xc = x.clone().requires_grad_(True)
    wc = w.clone().requires_grad_(True) if affine else None
    bc = b.clone().requires_grad_(True) if affine else None
    yc = F.instance_norm(xc, None, None, wc, bc, True, 0.1, 1e-5)
    if relu:
        yc = F.relu(yc)
    yc.backward(gy)
    xd = dev(x, dtype)
    wd = w.to(DEV).requires_grad_(True) if affine else None
    bd = b.to(DEV).requires_grad_(True) if affine else None
    yd = o.instance_norm_act(xd, wd, bd, relu=relu)
    yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    t = tol(dtype)
    assert relerr(yd, yc) < t * (1 if dtype == torch.float32 else 2)
    assert relerr(xd.grad, xc.grad) < 20 * t
    if affine:
        assert relerr(wd.grad, wc.grad) < 10 * t and relerr(bd.grad, bc.grad) < 10 * t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (4, 256, 9, 13), (16, 64, 6, 5)])
def test_np_plus(dtype, shape):
    o = ops()
    B, C, H, W = shape
    x = rnd(*shape, seed=6, scale=2.0) + rnd(B, C, 1, 1, seed=7, scale=3.0)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    alpha = 1 + 0.75 * rnd(B, C, 1, 1, seed=8)
    beta = 0.75 * rnd(B, C, 1, 1, seed=9)
    gy = rnd(*shape, seed=10)
    xc = x.clone().requires_grad_(True)
    yc = orc.np_plus(xc, alpha, beta)
    yc.backward(gy)
    xd = dev(x, dtype)
    yd = o.np_plus(xd, alpha.to(DEV), beta.to(DEV))
    yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    t = tol(dtype)
    assert relerr(yd, yc) < t
    assert relerr(xd.grad, xc.grad) < 10 * t
    # with the residual added in the same pass (the HRFP output, reference deepv3.py:333-334): fp32 bit-identical with the two
    # passes; the 16-bit types round once instead of twice
    r = rnd(*shape, seed=11)
    if dtype == torch.bfloat16:
        r = r.bfloat16().float()
    xd2, rd = dev(x, dtype), dev(r, dtype)
    y2 = o.np_plus(xd2, alpha.to(DEV), beta.to(DEV), res=rd)
    y2.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    two = o.add(rd.detach(), o.np_plus(xd2.detach(), alpha.to(DEV), beta.to(DEV)))
    assert torch.equal(y2, two) if dtype == torch.float32 else relerr(y2, two) < t
    assert torch.equal(xd2.grad, xd.grad) and relerr(rd.grad, gy) < (1e-7 if dtype == torch.float32 else t)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [((2, 64, 12, 12), dict(scale=1.205)), ((2, 64, 14, 19), dict(scale=1.2)),
                                  ((2, 128, 21, 21), dict(size=(24, 24))), ((2, 64, 24, 24), dict(scale=0.838)),
                                  ((2, 64, 20, 20), dict(scale=0.798)), ((1, 256, 83, 83), dict(size=(96, 96)))])
def test_hrfp_stage_resize_bn_relu(dtype, case):
    """nearest resize -> BN(train) -> ReLU fused, against F.interpolate + F.batch_norm + relu."""
    o = ops()
    shape, rs = case
    B, C, H, W = shape
    x = rnd(*shape, seed=11, scale=2.0, shift=0.3)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    w, b = torch.randn(C) * 0.5, torch.zeros(C)
    xc = x.clone().requires_grad_(True)
    if "scale" in rs:
        up = F.interpolate(xc, scale_factor=(rs["scale"], rs["scale"]))
    else:
        up = F.interpolate(xc, size=rs["size"])
    yc = F.relu(F.batch_norm(up, None, None, w, b, True, 0.1, 1e-5))
    gy = rnd(*yc.shape, seed=12)
    yc.backward(gy)
    xd = dev(x, dtype)
    plan = o.nearest_plan(H, W, device=DEV, **rs)
    assert (plan.Ho, plan.Wo) == tuple(yc.shape[2:])
    yd = o.batch_norm_act(xd, w.to(DEV), b.to(DEV), None, None, training=True, relu=True, plan=plan)
    yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    t = tol(dtype)
    assert relerr(yd, yc) < t
    assert relerr(xd.grad, xc.grad) < 10 * t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [((2, 256, 1, 1), (7, 9), False), ((2, 256, 6, 6), (24, 24), False),
                                  ((2, 19, 24, 20), (96, 80), False), ((1, 256, 12, 12), (24, 24), True),
                                  ((2, 64, 5, 7), (5, 7), False), ((1, 8, 9, 9), (4, 5), False)])
def test_bilinear(dtype, case):
    o = ops()
    shape, size, with_add = case
    x = rnd(*shape, seed=13)
    add = rnd(shape[0], shape[1], *size, seed=14) if with_add else None
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
        add = add.bfloat16().float() if with_add else None
    xc = x.clone().requires_grad_(True)
    ac = add.clone().requires_grad_(True) if with_add else None
    yc = orc.upsample_bilinear_ac(xc, size)
    if with_add:
        yc = ac + yc
    gy = rnd(*yc.shape, seed=15)
    yc.backward(gy)
    xd = dev(x, dtype)
    ad = dev(add, dtype) if with_add else None
    yd = o.upsample_bilinear(xd, size, addend=ad)
    yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    t = tol(dtype)
    assert relerr(yd, yc) < t
    assert relerr(xd.grad, xc.grad) < 4 * t
    if with_add:
        assert relerr(ad.grad, ac.grad) < t


def test_bilinear_backward_over_the_window_of_nonzero_columns_equals_the_full_loop(tmp_path):
    """bilinear_bwd_kernel<T, VEC, KW, KN> (csrc/resize_pool.hip, round 5): at a 2x / 4x upsample only the 5 of 8 / 9 of 12 candidate destination
    columns that can carry a weight are evaluated.  Against the full loop (MRFP_BILINEAR_WINDOW=0, child process: the switch is read once): the same
    non-zero products in the same order -- bit-identical -- and against the CPU restatement."""
    import os
    import subprocess
    import sys
    o = ops()
    cases = [((1, 32, 48, 48), (96, 96)), ((1, 64, 12, 12), (48, 48)), ((2, 24, 96, 96), (192, 192)), ((1, 256, 12, 12), (24, 24)), ((1, 8, 31, 57), (61, 113))]
    code = ("import sys, torch\nsys.path.insert(0, %r)\nfrom mrfp_amd import ops\nout = []\n"
            "for (shape, size) in %r:\n"
            "    for dtype in (torch.bfloat16, torch.float32):\n"
            "        g = torch.Generator().manual_seed(7)\n"
            "        x = torch.randn(*shape, generator=g).to('cuda:0', dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)\n"
            "        gy = torch.randn(shape[0], shape[1], *size, generator=g).to('cuda:0', dtype).contiguous(memory_format=torch.channels_last)\n"
            "        ops.upsample_bilinear(x, size).backward(gy)\n"
            "        out.append(x.grad.cpu())\n"
            "torch.save(out, sys.argv[1])\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), cases)
    files = []
    for tag, env in (("window", {}), ("full", {"MRFP_BILINEAR_WINDOW": "0"})):
        f = str(tmp_path / (tag + ".pt"))
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        files.append(torch.load(f))
    for a, b in zip(*files):
        assert torch.equal(a, b)
    i = 0
    for shape, size in cases:
        for dtype in (torch.bfloat16, torch.float32):
            g = torch.Generator().manual_seed(7)
            x = torch.randn(*shape, generator=g).to(dtype).float().requires_grad_(True)
            gy = torch.randn(shape[0], shape[1], *size, generator=g).to(dtype).float()
            orc.upsample_bilinear_ac(x, size).backward(gy)
            assert relerr(files[0][i], x.grad) < 4 * tol(dtype)
            i += 1


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (2, 64, 17, 13), (1, 8, 1, 1), (2, 128, 2, 5)])
def test_maxpool(dtype, shape):
    o = ops()
    x = rnd(*shape, seed=16)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xc = x.clone().requires_grad_(True)
    yc = F.max_pool2d(xc, 3, 2, 1)
    gy = rnd(*yc.shape, seed=17)
    yc.backward(gy)
    xd = dev(x, dtype)
    yd = o.max_pool_3x3_s2(xd)
    yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    assert relerr(yd, yc) == 0.0
    assert relerr(xd.grad, xc.grad) < tol(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 64, 16, 16), (2, 64, 17, 13), (3, 128, 24, 40), (1, 8, 1, 1), (2, 19, 7, 6), (2, 2048, 4, 5)])
@pytest.mark.parametrize("affine", [True, False])
def test_instance_norm_relu_pool_is_the_two_operator_sequence(dtype, shape, affine):
    """The stem's norm -> ReLU -> maxpool as one operator (reference Resnet.py:549-551): same output bits as
    instance_norm_act(relu=True) + max_pool_3x3_s2 (each window value rounded before the comparison), same gradient routing;
    the fp32 run is also held against torch."""
    o = ops()
    B, C, H, W = shape
    x = rnd(*shape, seed=41, scale=50.0, shift=120.0)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    w, b = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    outs = []
    hits = o.POOL_FUSED_HITS[0]
    for fused in (True, False):
        o.POOL_FUSED[0] = fused
        try:
            xd = dev(x, dtype)
            wd = w.to(DEV).requires_grad_(True) if affine else None
            bd = b.to(DEV).requires_grad_(True) if affine else None
            yd = o.instance_norm_relu_pool(xd, wd, bd)
            gy = rnd(*yd.shape, seed=42)
            yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
            outs.append((yd.detach(), xd.grad, wd.grad if affine else None, bd.grad if affine else None))
        finally:
            o.POOL_FUSED[0] = True
    assert o.POOL_FUSED_HITS[0] == hits + 1
    (yf, gxf, gwf, gbf), (yu, gxu, gwu, gbu) = outs
    assert torch.equal(yf, yu)
    t = tol(dtype)
    # the un-pooled gradient is rebuilt with the rounding maxpool_bwd stores it with: the two paths differ by the summation order
    # of the backward statistics only
    assert relerr(gxf, gxu) < (1e-5 if dtype == torch.float32 else 1e-2)
    if affine:
        assert relerr(gwf, gwu) < 1e-4 and relerr(gbf, gbu) < 1e-4
    xc = x.clone().requires_grad_(True)
    wc = w.clone().requires_grad_(True) if affine else None
    bc = b.clone().requires_grad_(True) if affine else None
    yc = F.max_pool2d(F.relu(F.instance_norm(xc, None, None, wc, bc, True, 0.1, 1e-5)), 3, 2, 1) if H * W > 1 else None
    if yc is not None:
        yc.backward(rnd(*yc.shape, seed=42))
        assert relerr(yf, yc) < t * (1 if dtype == torch.float32 else 2)
        if dtype == torch.float32:
            assert relerr(gxf, xc.grad) < 20 * t
            if affine:
                assert relerr(gwf, wc.grad) < 10 * t and relerr(gbf, bc.grad) < 10 * t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 64, 64, 24, 40, 1, dict(scale=1.205)), (2, 64, 128, 33, 47, 2, dict(scale=1.2)),
                                  (3, 128, 64, 20, 28, 1, dict(scale=0.838)), (2, 128, 256, 16, 48, 2, dict(size=(16, 48))),
                                  (2, 256, 128, 24, 24, 1, dict(size=(7, 9))), (1, 64, 96, 96, 96, 2, dict(scale=0.798))])
def test_batch_norm_behind_a_resize_takes_its_statistics_from_the_convolution_epilogue(dtype, case):
    """HRFP stage conv -> F.interpolate(nearest) -> BatchNorm(train) -> ReLU (reference deepv3.py:320-327): the convolution epilogue
    sums its output with the resize's pixel multiplicities (mrfp_conv_fwd_wstats), the BatchNorm runs no statistics pass.  Against
    the two-pass form (same kernels otherwise) and against torch."""
    o = ops()
    from mrfp_amd import conv as cv
    B, C, N, H, W, dil, rs = case
    x = rnd(B, C, H, W, seed=51)
    w = rnd(N, C, 3, 3, seed=52, scale=(2.0 / (C * 9)) ** 0.5)
    b = rnd(N, seed=53, scale=0.1)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    gam, bet = torch.rand(N) + 0.5, torch.randn(N) * 0.1
    outs = []
    for fused in (True, False):
        cv.WSTATS[0] = fused
        try:
            xd = dev(x, dtype)
            wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
            g, be = gam.to(DEV).requires_grad_(True), bet.to(DEV).requires_grad_(True)
            rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
            plan = o.nearest_plan(H, W, device=DEV, **rs)
            hits = cv.WSTATS_HITS[0]
            cv.STAT_RESIZE[0] = plan
            c = o.conv2d(xd, wd, bd, 1, dil, dil)
            assert cv.STAT_RESIZE[0] is None
            assert cv.WSTATS_HITS[0] == hits + (1 if fused else 0)
            assert (getattr(c, "_mrfp_colstats", None) is not None) == fused
            y = o.batch_norm_act(c, g, be, rm, rv, training=True, momentum=0.1, eps=1e-5, relu=True, plan=plan)
            gy = rnd(*y.shape, seed=54)
            y.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
            outs.append((y.detach().float(), xd.grad.float(), g.grad, be.grad, rm, rv))
        finally:
            cv.WSTATS[0] = True
    t = tol(dtype)
    for a, bb in zip(outs[0], outs[1]):       # fused vs two-pass: only the summation order of the statistics differs
        assert relerr(a, bb) < (2e-5 if dtype == torch.float32 else 2e-2)
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    gc, bec = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    cc = F.conv2d(xc, wc, bc, 1, dil, dil)
    kw = dict(scale_factor=rs["scale"]) if "scale" in rs else dict(size=rs["size"])
    rmc, rvc = torch.zeros(N), torch.ones(N)
    yc = F.relu(F.batch_norm(F.interpolate(cc, mode="nearest", **kw), rmc, rvc, gc, bec, True, 0.1, 1e-5))
    yc.backward(rnd(*yc.shape, seed=54))
    y, gx, gg, gb, rm, rv = outs[0]
    assert tuple(y.shape) == tuple(yc.shape)
    assert relerr(y, yc) < (20 * t if dtype == torch.float32 else 2 * t)
    assert relerr(rm.cpu(), rmc) < 20 * t and relerr(rv.cpu(), rvc) < 20 * t
    if dtype == torch.float32:
        # (relative L2: an output within rounding of 0 may sit on the other side of the ReLU than torch's, and that one gate moves
        #  a handful of input gradients by percents of the maximum)
        l2 = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
        assert l2(gx, xc.grad) < 50 * t and l2(gg, gc.grad) < 50 * t and l2(gb, bec.grad) < 50 * t


@pytest.mark.parametrize("dtype", DTYPES)
def test_global_avg_pool_add_relu(dtype):
    o = ops()
    x = rnd(2, 256, 7, 9, seed=18)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xc = x.clone().requires_grad_(True)
    yc = F.adaptive_avg_pool2d(xc, 1)
    gy = rnd(2, 256, 1, 1, seed=19)
    yc.backward(gy)
    xd = dev(x, dtype)
    yd = o.global_avg_pool(xd)
    yd.backward(gy.to(DEV, dtype))
    assert relerr(yd, yc) < tol(dtype) and relerr(xd.grad, xc.grad) < tol(dtype)
    a, b = dev(rnd(2, 48, 5, 5, seed=20), dtype), dev(rnd(2, 48, 5, 5, seed=21), dtype)
    s = o.relu(o.add(a, b))
    s.backward(torch.ones_like(s))
    ref = F.relu(a.detach().float() + b.detach().float())
    assert relerr(s, ref) < tol(dtype)
    assert relerr(a.grad, (ref > 0).float()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 19, 24, 20), (1, 19, 7, 5)])
def test_cross_entropy_and_hist(dtype, shape):
    o = ops()
    B, C, H, W = shape
    x = rnd(*shape, seed=22, scale=3.0)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    g = torch.Generator().manual_seed(23)
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[torch.rand(B, H, W, generator=g) < 0.1] = 255
    xc = x.clone().requires_grad_(True)
    lc = orc.cross_entropy_255(xc, y)
    (lc * 1.7).backward()
    xd = dev(x, dtype)
    ld = o.cross_entropy(xd, y.to(DEV), 255)
    (ld * 1.7).backward()
    assert abs(ld.item() - lc.item()) / abs(lc.item()) < 1e-5
    assert relerr(xd.grad, xc.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    hist, pred = o.argmax_hist(xd, y.to(DEV), want_pred=True)
    ref_pred = x.numpy().argmax(1)
    np.testing.assert_array_equal(pred.cpu().numpy(), ref_pred)
    np.testing.assert_array_equal(hist.cpu().numpy(), orc.fast_hist(ref_pred.flatten(), y.numpy().flatten(), C))
    # all-ignored batch -> NaN like torch
    l2 = o.cross_entropy(xd.detach(), torch.full((B, H, W), 255, dtype=torch.long, device=DEV), 255)
    assert math.isnan(l2.item())


def test_cpu_input_fails_loudly():
    from mrfp_amd import _lib
    o = ops()
    with pytest.raises(_lib.MrfpHipError):
        o.relu(torch.zeros(1, 8, 2, 2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [((2, 19, 12, 10), (48, 40)), ((1, 19, 7, 9), (25, 33)), ((2, 19, 16, 16), (16, 16)),
                                  # rows longer than one 256-pixel segment of the row kernel; scale 1 (a segment then touches 257 source pixels)
                                  ((1, 19, 20, 150), (40, 300)), ((1, 19, 5, 600), (5, 600)), ((2, 19, 3, 257), (7, 771))])
def test_fused_upsample_cross_entropy(dtype, case):
    """upsample + CE fused (no full-resolution logits) == Upsample() then CrossEntropyLoss(255) on the CPU."""
    o = ops()
    (B, C, Hi, Wi), (H, W) = case
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, C, Hi, Wi, generator=g) * 2
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    y = torch.randint(0, C, (B, H, W), generator=g)
    y[torch.rand(B, H, W, generator=g) < 0.1] = 255
    xc = x.clone().requires_grad_(True)
    lc = orc.cross_entropy_255(orc.upsample_bilinear_ac(xc, (H, W)), y)
    (lc * 0.7).backward()
    P = torch.zeros(B, 32, Hi, Wi)
    P[:, :C] = x
    Pd = P.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ld = o.upsample_cross_entropy(Pd, y.to(DEV), (H, W), C, 255)
    (ld * 0.7).backward()
    assert abs(ld.item() - lc.item()) / abs(lc.item()) < (1e-5 if dtype == torch.float32 else 2e-3)
    assert relerr(Pd.grad[:, :C], xc.grad) < (2e-5 if dtype == torch.float32 else 1.5e-2)
    assert float(Pd.grad[:, C:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("chans", [(48, 256), (256, 256, 256, 256, 256), (19, 8), (64,)])
def test_concat_channels(dtype, chans):
    """torch.cat(dim=1) of NHWC activations and its backward: strided channel-block copies (bit-exact data movement)."""
    o = ops()
    xs = [rnd(2, c, 9, 7, seed=30 + i) for i, c in enumerate(chans)]
    xd = [x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True) for x in xs]
    y = o.concat_channels(xd)
    ref = torch.cat([x.to(DEV, dtype) for x in xs], 1)
    assert torch.equal(y, ref) and y.is_contiguous(memory_format=torch.channels_last)
    gy = rnd(*ref.shape, seed=40).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    c0 = 0
    for x, c in zip(xd, chans):
        assert torch.equal(x.grad, gy[:, c0:c0 + c])
        c0 += c


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("case", [
    # (B, Cin, H, W, Cmid, k, stride, dil, residual)   conv_a -> BN(+res) -> ReLU -> conv_b(k, stride, dil)
    (2, 64, 20, 18, 64, 3, 1, 1, False),        # bn -> relu -> 3x3: mask recomputed from x*A+S
    (3, 128, 12, 12, 256, 1, 1, 1, True),       # bn + residual -> relu -> 1x1: mask from the stored output
    (2, 64, 16, 16, 128, 3, 2, 1, False),       # strided consumer: the dgrad launch runs with a source stride
    (2, 64, 9, 11, 64, 3, 1, 2, False),         # dilated consumer, ragged M (not a multiple of any tile)
    (2, 64, 48, 48, 64, 1, 1, 1, True),         # N = 64 -> the 256x64 tile (no wide epilogue)
    (16, 64, 24, 24, 256, 3, 1, 1, False),      # M = 9216 rows: several row blocks per launch
])
def test_conv_bn_relu_conv_chain_backward_vs_torch(dtype, tol, case):
    """conv_a -> BatchNorm(+residual) -> ReLU -> conv_b (reference Resnet.py:202-216 and its autograd): every gradient of the
    chain against torch autograd on the CPU in fp32 -- the BatchNorm-backward statistics pass, the recomputed / sign-mask ReLU
    gates and the strided / dilated dgrad launches that feed it.  (Round 2 also produced those statistics in the dgrad epilogue,
    mrfp_conv_dgrad_bnstats; it measured a wash -- profiles/r02_experiments.md -- and was removed in round 3.)"""
    import torch.nn.functional as F
    from mrfp_amd import conv, ops
    B, Cin, H, W, Cm, k, st, dil, with_res = case
    g = torch.Generator().manual_seed(sum(case[:8]))
    x0 = torch.randn(B, Cin, H, W, generator=g)
    wa = torch.randn(Cm, Cin, 1, 1, generator=g) * (1.0 / Cin) ** 0.5
    wb = torch.randn(64, Cm, k, k, generator=g) * (1.0 / (Cm * k * k)) ** 0.5
    gam, bet = torch.rand(Cm, generator=g) + 0.5, torch.randn(Cm, generator=g) * 0.2
    res = torch.randn(B, Cm, H, W, generator=g) if with_res else None
    if dtype != torch.float32:
        x0, wa, wb = x0.to(dtype).float(), wa.to(dtype).float(), wb.to(dtype).float()
        res = res.to(dtype).float() if with_res else None
    pad = dil * (k - 1) // 2

    def cpu():
        x, a, b2, ga, be = (t.clone().requires_grad_(True) for t in (x0, wa, wb, gam, bet))
        h = F.batch_norm(F.conv2d(x, a), None, None, ga, be, True, 0.1, 1e-5)
        h = F.relu(h + res) if with_res else F.relu(h)
        y = F.conv2d(h, b2, None, st, pad, dil)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7))
        y.backward(gy)
        return gy, (x.grad, a.grad, ga.grad, be.grad)

    def hip(gy):
        x = x0.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        a, b2 = wa.to(DEV).requires_grad_(True), wb.to(DEV).requires_grad_(True)
        ga, be = gam.to(DEV).requires_grad_(True), bet.to(DEV).requires_grad_(True)
        r = res.to(DEV, dtype).contiguous(memory_format=torch.channels_last) if with_res else None
        h = ops.batch_norm_act(conv.conv2d(x, a, None, 1, 0, 1), ga, be, None, None, training=True, relu=True, res=r)
        y = conv.conv2d(h, b2, None, st, pad, dil)
        y.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
        return x.grad.float().cpu(), a.grad.cpu(), ga.grad.cpu(), be.grad.cpu()

    gy, ref = cpu()
    if dtype != torch.float32:
        gy = gy.to(dtype).float()
    got = hip(gy)
    for name, f, r_ in zip(("dx", "dw", "dgamma", "dbeta"), got, ref):
        scale = r_.abs().max().item()
        if dtype == torch.float32:
            assert (f - r_).abs().max().item() <= tol * scale, name
        else:       # bf16 activations flip the ReLU mask of a few near-zero pre-activations: compare in the L2 norm
            assert ((f - r_).norm() / r_.norm()).item() <= 2.5 * tol, name


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 64, 9, 7), (3, 256, 12, 10), (1, 1024, 5, 6)])
def test_residual_bn_relu_sign_mask_equals_reading_y(dtype, shape):
    """The residual BatchNorm -> add -> ReLU tail (reference Resnet.py:202-225) keeps ONE BIT per output element for its
    backward (mrfp_affine_fwd_relu_mask / mrfp_stats_bwd_mask / mrfp_affine_bwd_mask) instead of re-reading y in both backward
    passes: outputs and every gradient must be bit-identical with the path that reads y (ops.SIGN_MASK off), including
    elements that are exactly zero after rounding."""
    o = ops()
    B, C, H, W = shape
    x = rnd(*shape, seed=11, scale=2.0)
    r = rnd(*shape, seed=12)
    r[0, :, 0, 0] = -1e30                         # a clamped pixel
    r[-1, :, -1, -1] = 0.0
    w, b = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    gy = rnd(*shape, seed=13)
    outs = []
    for use_mask in (True, False):
        o.SIGN_MASK[0] = use_mask
        try:
            xd, rd = dev(x, dtype), dev(r, dtype)
            wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
            rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            yd = o.batch_norm_act(xd, wd, bd, rm, rv, training=True, relu=True, res=rd)
            yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
            outs.append((yd.detach().clone(), xd.grad.clone(), rd.grad.clone(), wd.grad.clone(), bd.grad.clone()))
        finally:
            o.SIGN_MASK[0] = True
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)
    assert (outs[0][0] == 0).any() and (outs[0][0] > 0).any()


def test_gated_skip_gradient_equals_the_materialised_one():
    """Residual blocks (reference Resnet.py:202-225): the gradient of the skip connection is dy * [out > 0].  With the sign mask
    of the block's tail and a skip that is the alias of the block's first convolution, backward hands that convolution the
    UNMASKED dy + the mask and its dgrad epilogue gates the addend (mrfp_conv_fwd_gated) -- every parameter gradient of a
    ResNet-50 trunk must be bit-identical with the path that writes the masked gradient, and the gated launches must be the
    blocks without a downsample branch (12 of 16)."""
    from mrfp_amd import conv
    from mrfp_amd.config import cfg
    from mrfp_amd.network import Resnet
    o = ops()
    cfg.MODEL.ACT_DTYPE = torch.bfloat16
    try:
        torch.manual_seed(3)
        net = Resnet.resnet50(pretrained=False, wt_layer=[0] * 7).to(DEV).train()
        x = (torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(4)) * 255).to(DEV)
        grads = []
        for gated in (True, False):
            o.GATED_SKIP[0] = gated
            conv.GATED_SKIP_HITS[0] = 0
            o.GATED_BN_HITS[0] = 0
            net.zero_grad(set_to_none=True)
            y = net(x)
            y = y[0] if isinstance(y, (tuple, list)) else y
            y.float().pow(2).mean().backward()
            torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
            assert conv.GATED_SKIP_HITS[0] == (12 if gated else 0)
            # ... and the four blocks WITH a downsample branch hand the unmasked gradient + mask to that branch's BatchNorm
            assert o.GATED_BN_HITS[0] == (4 if gated else 0)
        assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 100
        for k in grads[0]:
            assert torch.equal(grads[0][k], grads[1][k]), k
        # the stand-alone form of the gate (what a consumer that cannot fuse it gets)
        t = torch.randn(2, 64, 5, 7, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
        bits = torch.randint(0, 256, (t.numel() // 8,), dtype=torch.uint8, device=DEV)
        t._mrfp_gate = (bits, t._version)
        ref = t.permute(0, 2, 3, 1).reshape(-1, 8).float() * torch.stack([(bits >> i) & 1 for i in range(8)], 1).float()
        out = conv.ungate(t).permute(0, 2, 3, 1).reshape(-1, 8).float()
        assert torch.equal(out, ref)
        # a skip alias with a SECOND consumer: autograd sums the two gradients into an untagged tensor, so the tail must fall back to
        # the materialised masked gradient (the use count of the alias is read in backward) -- same gradients as with the gate off
        from mrfp_amd.network.mynn import HipConv2d, HipBatchNorm2d
        torch.manual_seed(5)
        cv, bn = HipConv2d(64, 64, kernel_size=1, bias=False).to(DEV), HipBatchNorm2d(64).to(DEV)
        xin = torch.randn(2, 64, 12, 12, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
        res = []
        for gated in (True, False):
            o.GATED_SKIP[0] = gated
            conv.GATED_SKIP_HITS[0] = 0
            xi = xin.clone().requires_grad_(True)
            c, alias = cv.forward_skip(xi)
            out = bn.fused(c, relu=True, res=alias) + o.add(alias, alias)       # the alias feeds the tail AND another operator
            out.float().pow(2).mean().backward()
            res.append(xi.grad.clone())
            assert conv.GATED_SKIP_HITS[0] == 0
        assert torch.equal(res[0], res[1])
        # ... and a consumer the use count cannot see (a raw torch op on the alias): the convolution that owns the alias gets an
        # untagged sum where the tagged tensor was promised and must refuse it instead of adding it as if it were masked
        from mrfp_amd._lib import MrfpHipError
        o.GATED_SKIP[0] = True
        xi = xin.clone().requires_grad_(True)
        c, alias = cv.forward_skip(xi)
        out = bn.fused(c, relu=True, res=alias).float() + alias.float() * 2.0
        with pytest.raises(MrfpHipError, match="without its gate"):
            out.pow(2).mean().backward()
    finally:
        o.GATED_SKIP[0] = True
        cfg.MODEL.ACT_DTYPE = torch.float32


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_concat_upsample_equals_the_composition(dtype):
    """cat([a, Upsample(b)], 1) with the bilinear kernel writing into / reading from its channel block of the concatenation
    (mrfp_bilinear_fwd_into / mrfp_bilinear_bwd_from): bit-identical with concat_channels([a, upsample_bilinear(b)])."""
    o = ops()
    a = rnd(2, 48, 24, 20, seed=21)
    b = rnd(2, 256, 6, 5, seed=22)
    gy = rnd(2, 304, 24, 20, seed=23).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    outs = []
    for fused in (True, False):
        ad, bd = dev(a, dtype), dev(b, dtype)
        y = o.concat_upsample(ad, bd, (24, 20)) if fused else o.concat_channels([ad, o.upsample_bilinear(bd, (24, 20))])
        y.backward(gy)
        outs.append((y.detach().clone(), ad.grad.clone(), bd.grad.clone()))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    yc = torch.cat([a, F.interpolate(b, size=(24, 20), mode="bilinear", align_corners=True)], 1)
    assert relerr(outs[0][0], yc) < tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_padded_decoder_concat_feeds_a_channel_padded_convolution(dtype):
    """cfg.MODEL.DECODER_PAD: cat([a 48, Upsample(b) 256]) carried as 320 channels (zeros behind the 304) into the 304-channel 3x3
    convolution of the decoder (reference deepv3.py:200-208, 349-353): forward, the gradients of a and b (dgrad over a
    channel-padded input: its pad channels come out exactly zero) and the weight gradient against torch on the unpadded
    tensors."""
    from mrfp_amd import conv
    o = ops()
    a, b = rnd(2, 48, 24, 16, seed=31), rnd(2, 256, 6, 4, seed=32)
    w = rnd(64, 304, 3, 3, seed=33, scale=0.05)
    if dtype != torch.float32:
        a, b, w = a.to(dtype).float(), b.to(dtype).float(), w.to(dtype).float()
    ac, bc, wc = a.clone().requires_grad_(True), b.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yc = F.conv2d(torch.cat([ac, F.interpolate(bc, size=(24, 16), mode="bilinear", align_corners=True)], 1), wc, None, 1, 1)
    gy = rnd(*yc.shape, seed=34)
    if dtype != torch.float32:
        gy = gy.to(dtype).float()
    yc.backward(gy)
    ad, bd = dev(a, dtype), dev(b, dtype)
    wd = w.to(DEV).requires_grad_(True)
    cat = o.concat_upsample(ad, bd, (24, 16), 64)
    assert cat.shape[1] == 320 and float(cat.detach()[:, 304:].abs().max()) == 0.0
    cat.retain_grad()
    y = conv.conv2d(cat, wd, None, 1, 1, 1)
    y.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    t = 1e-5 if dtype == torch.float32 else 1.5e-2
    assert relerr(y, yc) < t and relerr(ad.grad, ac.grad) < t and relerr(bd.grad, bc.grad) < t and relerr(wd.grad, wc.grad) < 2 * t
    assert tuple(wd.grad.shape) == (64, 304, 3, 3) and float(cat.grad[:, 304:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(4, 8, 64, 96, 96, 3, 2), (2, 64, 64, 64, 48, 3, 1), (2, 64, 128, 48, 48, 3, 1), (3, 64, 256, 24, 24, 1, 1)])
def test_instance_norm_takes_its_statistics_from_the_convolution_epilogue(dtype, case):
    """reference Resnet.py:534-536 / 176-178: the stem convolutions feed nn.InstanceNorm2d.  When no statistics row block of the
    producing convolution straddles an image (H*W a multiple of the block height) its epilogue sums ARE the per-image partials:
    the statistics pass over the convolution output is skipped.  Against the separate pass (same stored values, another summation
    order): output, input gradient and weight gradients equal to fp32 rounding; a geometry whose blocks straddle images falls back."""
    from mrfp_amd import conv
    o = ops()
    B, C, N, H, W, k, stride = case
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(N, C, k, k, generator=g) * 0.1
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.1
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    gy = torch.randn(B, N, Ho, Wo, generator=g)
    res = []
    for fused in (True, False):
        o.IN_FUSED_STATS[0] = fused
        o.IN_FUSED_HITS[0] = 0
        try:
            xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wd, gd, bd = w.to(DEV).requires_grad_(True), gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
            y = o.instance_norm_act(conv.conv2d(xd, wd, None, stride, k // 2, 1), gd, bd, relu=True)
            y.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
            torch.cuda.synchronize()
            res.append((y.detach().float().cpu(), xd.grad.float().cpu(), wd.grad.cpu(), gd.grad.cpu(), bd.grad.cpu(), o.IN_FUSED_HITS[0]))
        finally:
            o.IN_FUSED_STATS[0] = True
    assert res[1][5] == 0
    if dtype == torch.float32:
        assert res[0][5] == 0                        # fp32 (parity mode) keeps the dedicated statistics pass
    elif (Ho * Wo) % 192 == 0 and k == 3:
        assert res[0][5] == 1                        # whatever tile the launch runs on (64- or 96-row blocks), no block straddles an image
    # (the pointwise case runs on the persistent kernel, whose row blocks are workgroup tile ranges: they straddle images here -> fallback)
    t = 2e-5 if dtype == torch.float32 else 2e-2
    for a, b in zip(res[0][:5], res[1][:5]):
        assert relerr(a, b) < t
    if dtype == torch.float32:          # and against torch
        yc = torch.relu(F.instance_norm(F.conv2d(x, w, None, stride, k // 2), weight=gamma, bias=beta))
        assert relerr(res[0][0], yc) < 1e-4
