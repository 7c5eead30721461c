"""MFMA implicit-GEMM convolution (fwd / dgrad / wgrad, through the C ABI) against torch's fp32
convolution on the CPU, for every (channels, kernel, stride, dilation) family of the network.

fp32: v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain -> 1e-5 relative to the tensor max.
bf16: inputs are rounded to bf16 first (so only accumulation-order and output rounding differ) -> 1e-2.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (B, Cin, H, W, Cout, k, stride, pad, dil, bias)
CASES = [
    (2, 3, 40, 36, 64, 7, 2, 3, 1, False),       # stem (channel-padded input, no dgrad)
    (2, 64, 20, 18, 64, 1, 1, 0, 1, False),      # bottleneck 1x1
    (2, 64, 20, 18, 64, 3, 1, 1, 1, True),       # HRFP 3x3 + bias, Cout 64 -> 256x64 tile
    (2, 64, 19, 23, 128, 3, 1, 2, 2, True),      # HRFP dilated
    (1, 128, 17, 17, 256, 3, 1, 2, 2, True),
    (2, 128, 16, 16, 128, 3, 2, 1, 1, False),    # strided 3x3 (layer2.0.conv2)
    (2, 256, 16, 16, 512, 1, 2, 0, 1, False),    # strided 1x1 downsample
    (2, 256, 9, 9, 256, 3, 1, 6, 6, False),      # ASPP dilation 6 (> feature size: mostly padding)
    (1, 512, 8, 8, 256, 3, 1, 12, 12, False),
    (2, 304, 12, 10, 256, 3, 1, 1, 1, False),    # decoder concat 48+256
    (2, 256, 12, 10, 48, 1, 1, 0, 1, False),     # bot_fine
    (2, 256, 12, 10, 19, 1, 1, 0, 1, True),      # final2 (N padded to a chunk)
    (3, 64, 33, 31, 64, 3, 1, 1, 1, False),      # M not a multiple of the tile
    (2, 128, 240, 240, 256, 3, 1, 1, 1, True),   # enough rows and K tiles for the 192x128 tile / row-reuse kernel
    (2, 128, 240, 236, 256, 3, 1, 2, 2, False),  # 192x128 tile, dilated, ragged M
    (2, 1152, 244, 240, 256, 1, 1, 0, 1, False), # long-K 1x1 conv with 18 K tiles
    (2, 304, 240, 240, 256, 3, 1, 1, 1, False),  # taps straddling K tiles (304 channels)
    # the B-stationary kernel (16-bit types, pointwise, C in {64, 128, 256}, N >= 128): every K depth, ragged M (not a
    # multiple of the 64-row tile), N that ends inside a 128-column panel / inside a 16-column block, one-tile problems
    (2, 64, 20, 18, 256, 1, 1, 0, 1, False),
    (3, 128, 33, 31, 512, 1, 1, 0, 1, False),
    (2, 256, 24, 24, 1024, 1, 1, 0, 1, False),
    (1, 256, 7, 9, 136, 1, 1, 0, 1, False),
    (2, 128, 96, 96, 200, 1, 1, 0, 1, False),
    (4, 256, 48, 48, 1024, 1, 1, 0, 1, False),   # several tiles per workgroup: the ring wraps
    # K = 32 (64-byte rows: the HALF variant, round 5): the dgrad shape of the 19-class head (32 padded classes -> 256), ragged M / N
    (2, 32, 20, 18, 256, 1, 1, 0, 1, False),
    (3, 32, 33, 31, 136, 1, 1, 0, 1, False),
    (2, 32, 96, 96, 256, 1, 1, 0, 1, False),
    # dgrad of a stride-2 convolution with parity-class-major rows (even sizes, classes of whole tiles): 3x3 (1, 2, 2, 4 taps per
    # class), 1x1 (one class with a tap, three that only store zeros), the 256x64 tile (N <= 64 in dgrad form)
    (4, 128, 32, 32, 128, 3, 2, 1, 1, False),
    (4, 256, 32, 32, 512, 1, 2, 0, 1, False),
    (4, 64, 64, 48, 128, 3, 2, 1, 1, False),
]


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(dtype, case):
    from mrfp_amd import conv, ops
    B, Cin, H, W, Cout, k, st, pad, dil, has_bias = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1 if has_bias else None
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bc = b.clone().requires_grad_(True) if has_bias else None
    yc = F.conv2d(xc, wc, bc, st, pad, dil)
    gy = torch.randn(yc.shape, generator=g)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    yc.backward(gy)

    stem = Cin == 3
    if stem:
        xd = conv.pad_input_channels(x.to(DEV), dtype)
    else:
        xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV).requires_grad_(True) if has_bias else None
    yd = conv.conv2d(xd, wd, bd, st, pad, dil)
    assert tuple(yd.shape) == tuple(yc.shape)
    yd.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert relerr(yd, yc) < tol
    if not stem:
        assert relerr(xd.grad, xc.grad) < tol
    assert relerr(wd.grad, wc.grad) < tol * (1 if dtype == torch.float32 else 2)
    if has_bias:
        assert relerr(bd.grad, bc.grad) < tol
    # weight pack cache follows in-place updates of the master weight
    with torch.no_grad():
        wd.mul_(0.5)
    y2 = conv.conv2d(xd.detach(), wd, bd, st, pad, dil)
    y2c = F.conv2d(x, w * 0.5, b, st, pad, dil)
    assert relerr(y2, y2c) < tol


def test_padded_logits_path():
    """final2 with the 32-channel padded low-resolution buffer + bilinear on the first 19 channels."""
    from mrfp_amd import conv, ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 256, 12, 10, generator=g)
    w = torch.randn(19, 256, 1, 1, generator=g) * 0.1
    b = torch.randn(19, generator=g) * 0.1
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yc = F.interpolate(F.conv2d(xc, wc, bc), size=(48, 40), mode="bilinear", align_corners=True)
    gy = torch.randn(yc.shape, generator=g)
    yc.backward(gy)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    p = conv.conv2d(xd, wd, bd, 1, 0, 1, phys_out=32)
    assert tuple(p.shape) == (2, 32, 12, 10) and float(p.detach()[:, 19:].abs().max()) == 0.0
    yd = ops.upsample_bilinear(p, (48, 40), channels=19)
    yd.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
    assert relerr(yd, yc) < 1e-5 and relerr(xd.grad, xc.grad) < 1e-5
    assert relerr(wd.grad, wc.grad) < 1e-5 and relerr(bd.grad, bc.grad) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 64, 20, 18, 128, 3), (3, 128, 33, 31, 64, 1), (2, 128, 240, 240, 256, 3), (2, 128, 48, 40, 256, 1),
                                  (4, 256, 48, 48, 1024, 1)])
def test_conv_epilogue_statistics_feed_batchnorm(dtype, case):
    """BatchNorm statistics fused into the producing conv's epilogue == the separate statistics pass.  (The 16-bit pointwise
    kernels of conv_pw.hip sum their fp32 accumulators, the separate pass the rounded stored values: zero-mean rounding errors of
    2^-9 relative, i.e. ~1e-5 of a standard deviation in the batch mean -- the last two cases, with their own tolerance.)"""
    from mrfp_amd import conv, ops
    B, Cin, H, W, Cout, k = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g).to(DEV, dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).to(DEV)
    gamma, beta = torch.rand(Cout, device=DEV) + 0.5, torch.randn(Cout, device=DEV) * 0.1
    conv.FUSE_STATS[0] = True
    y1 = conv.conv2d(x, w, None, 1, k // 2, 1)
    assert getattr(y1, "_mrfp_colstats", None) is not None
    rm1, rv1 = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    z1 = ops.batch_norm_act(y1, gamma, beta, rm1, rv1, training=True, relu=True)
    conv.FUSE_STATS[0] = False
    y2 = conv.conv2d(x, w, None, 1, k // 2, 1)
    assert getattr(y2, "_mrfp_colstats", None) is None
    rm2, rv2 = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    z2 = ops.batch_norm_act(y2, gamma, beta, rm2, rv2, training=True, relu=True)
    conv.FUSE_STATS[0] = True
    assert torch.equal(y1, y2)
    assert relerr(z1, z2) < (1e-5 if dtype == torch.float32 else 1e-2)
    pw16 = dtype != torch.float32 and k == 1 and Cout >= 128
    assert relerr(rm1, rm2) < (5e-3 if pw16 else 1e-5) and relerr(rv1, rv2) < (1e-3 if pw16 else 1e-4)
    if pw16:          # the mean itself, against the spread of the channel: ~1e-5 standard deviations
        assert float(((rm1 - rm2).abs() / (0.1 * ((rv1 - 0.9) / 0.1).clamp_min(1e-12).sqrt())).max()) < 2e-4


def test_alternative_tile_variants_in_subprocess():
    """The generic kernels on the shapes the pointwise / dense-wgrad kernels normally take (MRFP_CONV_PW=0, MRFP_WGRAD_DENSE=0),
    and every tile shape forced onto shapes it is not the default for (MRFP_CONV_T96 / T192 = 2).  The switches are read once
    per process, so they are exercised in child processes.  (The measured-slower kernel variants of rounds 1-2 -- 8-wave tile,
    asynchronous rings on the 4-wave tiles, register staging -- left the library in round 3: profiles/r02_experiments.md.)"""
    import os
    import subprocess
    import sys
    code = (
        "import torch, torch.nn.functional as F\n"
        "from mrfp_amd import conv\n"
        # (the last two: dgrads of stride-2 convolutions whose parity classes -- 1024 rows -- are whole 128- / 256-row tiles but NOT whole
        #  96- / 192-row tiles: with those tiles forced the launch must fall back to the per-pixel row order)
        "for (B,Cin,H,W,Cout,k,pad,dil,st) in [(2,128,240,240,256,3,1,1,1),(2,304,120,120,256,3,1,1,1),(3,64,33,31,64,3,1,1,1),(2,256,48,40,512,1,0,1,1),\n"
        "                                     (4,128,32,32,128,3,1,1,2),(4,256,32,32,512,1,0,1,2),\n"
        # K = 32 pointwise (the HALF variant of the B-stationary kernel with MRFP_CONV_PW32=1, the generic kernel otherwise)
        "                                     (2,32,20,18,256,1,0,1,1),(3,32,33,31,136,1,0,1,1),(2,32,96,96,256,1,0,1,1)]:\n"
        "    g = torch.Generator().manual_seed(1)\n"
        "    x = torch.randn(B,Cin,H,W,generator=g).bfloat16().float(); w = (torch.randn(Cout,Cin,k,k,generator=g)*0.05).bfloat16().float()\n"
        "    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)\n"
        "    yc = F.conv2d(xc, wc, None, st, pad, dil); gy = torch.randn(yc.shape, generator=g).bfloat16().float(); yc.backward(gy)\n"
        "    xd = x.cuda().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True); wd = w.cuda().requires_grad_(True)\n"
        "    yd = conv.conv2d(xd, wd, None, st, pad, dil); yd.backward(gy.cuda().bfloat16().contiguous(memory_format=torch.channels_last))\n"
        "    rel = lambda a, b: ((a.double().cpu()-b.double()).abs().max()/b.double().abs().max()).item()\n"
        "    assert rel(yd, yc) < 1e-2 and rel(xd.grad, xc.grad) < 1e-2 and rel(wd.grad, wc.grad) < 2e-2, (Cin, rel(yd,yc), rel(xd.grad,xc.grad), rel(wd.grad,wc.grad))\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (MRFP_CONV_C64=0: the 64 / 128-channel 3x3 shapes stay on the implicit-GEMM tiles this test is about)
    for extra in ({"MRFP_CONV_PW": "0", "MRFP_WGRAD_DENSE": "0"}, {"MRFP_CONV_T96": "2", "MRFP_CONV_C64": "0"},
                  {"MRFP_CONV_T192": "2", "MRFP_CONV_RR": "0", "MRFP_CONV_C64": "0"}, {"MRFP_CONV_PW32": "1"}):
        env = dict(os.environ, PYTHONPATH=root, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stdout[-500:], r.stderr[-1500:])


def test_activation_above_the_buffer_descriptor_range():
    """One launch reads its input through 32-bit buffer-descriptor offsets (< 3.75 GB).  A larger activation --
    BASELINE.json configs[4] at 16 images per GPU has a 4.3 GB one -- is walked in batch ranges by mrfp_conv_fwd /
    mrfp_conv_wgrad themselves.  6 x 64 x 2432 x 2432 bf16 = 4.54 GB through a 1x1 convolution, forward + dgrad + wgrad,
    against the same convolution run on the two halves of the batch (each below the limit; that path is checked against
    torch above): forward and dgrad must be bit-identical, the weight gradient equal up to the split-K summation order."""
    from mrfp_amd import _lib, conv
    B, C, H, W, N = 6, 64, 2432, 2432, 64
    assert B * C * H * W * 2 > 0xF0000000 and (B // 2) * C * H * W * 2 < 0xF0000000
    assert _lib.lib().mrfp_conv_single_launch(B, C * H * W * 2) == 0 and _lib.lib().mrfp_conv_single_launch(B // 2, C * H * W * 2) == 1
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.empty(B, C, H, W, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x.normal_(generator=g)
    gy = torch.empty(B, N, H, W, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy.normal_(generator=g)
    w = (torch.randn(N, C, 1, 1, device=DEV, generator=g) * 0.1)

    def run(xs, gys):
        xs = xs.detach().requires_grad_(True)
        ws = w.detach().clone().requires_grad_(True)
        y = conv.conv2d(xs, ws, None, 1, 0, 1)
        y.backward(gys)
        return y.detach(), xs.grad, ws.grad

    y, dx, dw = run(x, gy)
    h = B // 2
    dws = []
    for sl in (slice(0, h), slice(h, B)):
        yh, dxh, dwh = run(x[sl], gy[sl])
        assert torch.equal(y[sl], yh) and torch.equal(dx[sl], dxh)
        dws.append(dwh)
        del yh, dxh
    ref = dws[0].double() + dws[1].double()
    assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    assert torch.isfinite(dw).all() and dw.abs().max() > 0


@pytest.mark.parametrize("shape", [(4, 256, 192, 192, 128, 1, 0), (4, 256, 48, 48, 1024, 1, 0), (3, 128, 96, 96, 512, 1, 0),
                                   (4, 256, 48, 48, 256, 3, 1), (2, 512, 48, 48, 512, 3, 2)])
def test_repeated_launches_are_bit_identical(shape):
    """The asynchronous transfers of the persistent pointwise kernel and the early-issue K loop are ordered by counted
    `s_waitcnt vmcnt(N)`: a miscount shows up as run-to-run differences (and NaNs) on ranges a few tiles long, not as a
    wrong result on every run -- the first shape here is the one that exposed one (a range of 5 tiles whose second tile
    was read before its last pieces had landed).  Six launches on the same operands: identical outputs and fused
    statistics, and equal to torch's convolution of the same bf16 operands."""
    from mrfp_amd import conv
    B, C, H, W, N, k, pad = shape
    dil = pad if k == 3 else 1
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.empty(B, C, H, W, device=DEV, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).normal_(generator=g)
    w = torch.empty(N, C, k, k, device=DEV).normal_(generator=g) * 0.05
    with torch.no_grad():
        ys, sts = [], []
        for _ in range(6):
            y = conv.conv2d(x, w, None, 1, pad, dil)
            sts.append(y._mrfp_colstats[0].clone())
            ys.append(y.clone())
        ref = F.conv2d(x.float(), w.bfloat16().float(), None, 1, pad, dil)
    assert torch.isfinite(ys[0].float()).all()
    for y, s in zip(ys[1:], sts[1:]):
        assert torch.equal(ys[0], y) and torch.equal(sts[0], s)
    assert ((ys[0].float() - ref).abs().max() / ref.abs().max()).item() < 1.5e-2


def _wgrad_single(x, dy, N, C, k, stride, pad, dil):
    from mrfp_amd import _lib
    from mrfp_amd._lib import call, dt, ptr, stream
    B, Cp, H, W = x.shape
    _, Np, Ho, Wo = dy.shape
    ws = torch.empty(int(_lib.lib().mrfp_conv_wgrad_ws_bytes(B * Ho * Wo, N, k * k * Cp)), dtype=torch.uint8, device=DEV)
    dw = torch.full((N, C, k, k), float("nan"), device=DEV)
    call("mrfp_conv_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), dt(x), B, H, W, Cp, C, N, Np, k, k, Ho, Wo, stride, pad, pad, dil, stream())
    return dw


def _wgrad_grouped(xs, dys, N, C, k, stride, pad, dil):
    import ctypes
    from mrfp_amd import _lib
    from mrfp_amd._lib import call, dt, ptr, stream
    B, Cp, H, W = xs[0].shape
    _, Np, Ho, Wo = dys[0].shape
    n = len(xs)
    ws = torch.empty(int(_lib.lib().mrfp_conv_wgrad_grouped_ws_bytes(B * Ho * Wo, N, k * k * Cp, n)), dtype=torch.uint8, device=DEV)
    dws = [torch.full((N, C, k, k), float("nan"), device=DEV) for _ in range(n)]
    arr = ctypes.c_void_p * n
    call("mrfp_conv_wgrad_grouped", arr(*[ptr(t) for t in xs]), arr(*[ptr(t) for t in dys]), arr(*[ptr(t) for t in dws]), n, ptr(ws),
         dt(xs[0]), B, H, W, Cp, C, N, Np, k, k, Ho, Wo, stride, pad, pad, dil, stream())
    return dws


# (B, Cin, H, W, Cout, k, stride, pad, dil, problems)
GROUP_CASES = [
    (4, 256, 48, 48, 1024, 1, 1, 0, 1, 22),      # layer3 conv3 (dense pointwise kernel)
    (4, 1024, 48, 48, 256, 1, 1, 0, 1, 22),      # layer3 conv1
    (4, 256, 48, 48, 256, 3, 1, 1, 1, 22),       # layer3 conv2 (gather kernel)
    (2, 64, 40, 36, 64, 3, 1, 1, 1, 2),          # N <= 64: the 64x256 tile, ragged M
    (3, 128, 33, 31, 128, 3, 2, 1, 1, 3),        # strided, M not a multiple of the K' tile
    (2, 512, 24, 24, 512, 3, 1, 2, 2, 2),        # dilated (layer4)
    (2, 304, 12, 10, 256, 3, 1, 1, 1, 5),        # channel-padded operand (Ctrue < C)
    (2, 256, 12, 10, 19, 1, 1, 0, 1, 32),        # the group limit; N padded to a chunk in dy
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", GROUP_CASES)
def test_grouped_wgrad_equals_the_single_launches(dtype, case):
    """mrfp_conv_wgrad_grouped: `problems` weight gradients of one geometry in one launch (reference network/Resnet.py:579-585: the
    repeated Bottlenecks of a stage).  Every problem's result against torch's convolution gradient of the same rounded operands
    and against the single launch (same products, another association of the K' splits: fp32 rounding apart), and a second
    grouped launch bit-identical to the first (fixed summation order)."""
    from mrfp_amd import conv
    B, C, H, W, N, k, stride, pad, dil, n = case
    g = torch.Generator(device=DEV).manual_seed(11)
    epc = 4 if dtype == torch.float32 else 8
    Cp, Np = conv._round_up(C, 64 if C == 304 else epc), conv._round_up(N, epc)
    Ho, Wo = conv._out_size(H, k, stride, pad, dil), conv._out_size(W, k, stride, pad, dil)
    xs, dys = [], []
    for _ in range(n):
        x = torch.zeros(B, H, W, Cp, device=DEV, dtype=dtype)
        x[..., :C].normal_(generator=g)
        dy = torch.zeros(B, Ho, Wo, Np, device=DEV, dtype=dtype)
        dy[..., :N].normal_(generator=g)
        xs.append(x.permute(0, 3, 1, 2))
        dys.append(dy.permute(0, 3, 1, 2))
    a = _wgrad_grouped(xs, dys, N, C, k, stride, pad, dil)
    b = _wgrad_grouped(xs, dys, N, C, k, stride, pad, dil)
    tol = 1e-5 if dtype == torch.float32 else 2e-5          # (bf16 operands are exact inputs; the accumulation is fp32 in both)
    for i in range(n):
        assert torch.equal(a[i], b[i])
        assert torch.isfinite(a[i]).all()
        single = _wgrad_single(xs[i], dys[i], N, C, k, stride, pad, dil)
        assert relerr(a[i], single) < tol
        if i in (0, n - 1):
            ref = torch.nn.grad.conv2d_weight(xs[i][:, :C].float().cpu(), (N, C, k, k), dys[i][:, :N].float().cpu(), stride, pad, dil)
            assert relerr(a[i], ref) < (2e-5 if dtype == torch.float32 else 1e-4)


def test_strided_dgrad_class_major_rows_equal_the_per_pixel_form():
    """dgrad of a stride-2 convolution (reference Resnet.py:202-216 conv2 of the first block of a stage, :579-585 its downsample
    branch): with the output rows enumerated parity class by parity class a tile walks only the taps its class has.  The same
    nonzero products in the same order as the per-pixel form (MRFP_DGRAD_CLASSED=0, child process; it adds exact zeros for the taps
    that do not exist): bit-identical dx, also with a skip-gradient addend in the epilogue, and equal to torch's gradient."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch, torch.nn.functional as F\n"
        "from mrfp_amd import conv, ops\n"
        "out = {}\n"
        "for (B,C,H,W,N,k,pad) in [(4,128,32,32,128,3,1),(4,256,32,32,512,1,0),(4,64,64,48,128,3,1),(16,256,96,96,256,3,1)]:\n"
        "    g = torch.Generator(device='cuda:0').manual_seed(5)\n"
        "    x = torch.randn(B,C,H,W,device='cuda:0',generator=g).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)\n"
        "    w = (torch.randn(N,C,k,k,device='cuda:0',generator=g)*0.05).requires_grad_(True)\n"
        "    y, skip = conv.conv2d(x, w, None, 2, pad, 1, want_skip=True)\n"
        "    gy = torch.randn(y.shape,device='cuda:0',generator=g).bfloat16().contiguous(memory_format=torch.channels_last)\n"
        "    z = ops.add(skip, skip)\n"                       # a second consumer of the block input: its gradient is the dgrad's addend
        "    gz = torch.randn(z.shape,device='cuda:0',generator=g).bfloat16().contiguous(memory_format=torch.channels_last)\n"
        "    torch.autograd.backward([y, z], [gy, gz])\n"
        "    ref = torch.nn.grad.conv2d_input(x.shape, w.detach().bfloat16().float().cpu(), gy.float().cpu(), 2, pad, 1) + 2 * gz.float().cpu()\n"
        "    err = ((x.grad.float().cpu() - ref).abs().max() / ref.abs().max()).item()\n"
        "    assert err < 1.5e-2, (C, N, k, err)\n"
        "    out[(B,C,H,W,N,k)] = x.grad.cpu()\n"
        "torch.save(out, sys.argv[1])\nprint('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for tag, extra in (("classed", {}), ("pixel", {"MRFP_DGRAD_CLASSED": "0"})):
        f = os.path.join("/tmp", "mrfp_dgrad_%s_%d.pt" % (tag, os.getpid()))
        env = dict(os.environ, PYTHONPATH=root, **extra)
        r = subprocess.run([sys.executable, "-c", code, f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stdout[-500:], r.stderr[-1500:])
        files.append(f)
    a, b = torch.load(files[0]), torch.load(files[1])
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for f in files:
        os.remove(f)


def test_accumulator_stationary_3x3_wgrad_in_subprocess():
    """conv_wg3_kernel (csrc/conv_wg3.hip): the weight gradient of the 3x3 / stride 1 / dilation 1-2 layers with dW[64 n][9 taps][64 c] held
    in the accumulators of a workgroup and the pixels streamed past once (rolling window of image rows + the dY strip), forced wherever
    it is legal (MRFP_WGRAD3=2; read once per process): all three strip forms (64 x 1 row, 96 x 1, 48 x 2 rows), both dilations, several
    channel blocks, single and grouped launches, main chunks only and main + remainder workgroups (7 and 22 problems), a channel-padded
    operand, bf16 and f16 -- against torch's convolution gradient of the same rounded operands, bit-identical between two launches."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import test_conv_gpu as t\n"
        "for (B,C,H,W,N,dil,n,Ct,dt) in [(2,64,16,64,64,1,1,64,'b'),(1,64,24,128,128,1,1,64,'b'),(2,128,12,96,128,1,3,128,'b'),(2,64,8,48,64,1,1,64,'b'),"
        "(2,128,16,48,256,2,2,128,'b'),(2,64,20,64,128,2,1,64,'h'),(2,320,12,64,256,1,1,304,'b'),(3,64,40,192,64,1,1,64,'b'),(5,128,40,64,128,1,7,128,'b'),"
        "(4,256,48,48,256,1,22,256,'b'),(2,128,24,96,64,2,2,128,'h'),(1,64,4,144,64,1,1,64,'b'),(2,64,8,144,128,2,1,64,'b'),(1,128,2,64,64,1,32,128,'b'),"
        "(4,128,48,192,128,1,5,128,'b'),(16,64,96,192,64,1,3,64,'b')]:\n"
        "    g = torch.Generator(device='cuda:0').manual_seed(3)\n"
        "    dtype = torch.bfloat16 if dt == 'b' else torch.float16\n"
        "    xs, dys = [], []\n"
        "    for _ in range(n):\n"
        "        x = torch.zeros(B,H,W,C,device='cuda:0',dtype=dtype)\n"
        "        x[..., :Ct] = torch.randn(B,H,W,Ct,device='cuda:0',generator=g).to(dtype)\n"
        "        xs.append(x.permute(0,3,1,2))\n"
        "        dys.append((torch.randn(B,H,W,N,device='cuda:0',generator=g) * (0.25 if dt == 'h' else 1.0)).to(dtype).permute(0,3,1,2))\n"
        "    a = t._wgrad_grouped(xs,dys,N,Ct,3,1,dil,dil) if n > 1 else [t._wgrad_single(xs[0],dys[0],N,Ct,3,1,dil,dil)]\n"
        "    b = t._wgrad_grouped(xs,dys,N,Ct,3,1,dil,dil) if n > 1 else [t._wgrad_single(xs[0],dys[0],N,Ct,3,1,dil,dil)]\n"
        "    for i in range(n):\n"
        "        assert torch.equal(a[i], b[i]) and torch.isfinite(a[i]).all(), (C, N, H, W, i)\n"
        "        ref = torch.nn.grad.conv2d_weight(xs[i][:, :Ct].float().cpu(), (N,Ct,3,3), dys[i].float().cpu(), 1, dil, dil)\n"
        "        assert t.relerr(a[i], ref) < 1e-4, (C, N, H, W, dil, i, t.relerr(a[i], ref))\n"
        "print('ok')\n") % os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"MRFP_WGRAD3": "2"}, {"MRFP_WGRAD3": "0"}):
        env = dict(os.environ, PYTHONPATH=root, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stdout[-500:], r.stderr[-1500:])


def test_accumulator_stationary_pointwise_wgrad_in_subprocess():
    """conv_wg1_kernel (csrc/conv_wg1.hip): the weight gradient of the pointwise layers with C % 256 == N % 256 == 0 with dW[256 n][256 c] held in
    the accumulators of a 512-thread workgroup and the pixels streamed through a ring of four stages, forced wherever it is legal
    (MRFP_WGRAD1=2; read once per process): one and several channel blocks on either side, single and grouped launches, main chunks only and
    main + remainder workgroups, rings shorter than their depth (1 and 2 units), bf16 and f16 -- against torch's convolution gradient of the
    same rounded operands, bit-identical between two launches."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import test_conv_gpu as t\n"
        "for (B,C,H,W,N,n,dt) in [(2,256,8,16,256,1,'b'),(1,512,16,16,256,3,'b'),(2,256,12,16,1024,5,'b'),(4,1024,48,48,256,22,'b'),(2,256,10,16,256,1,'h'),"
        "(1,2048,8,8,512,2,'b'),(1,256,4,8,256,1,'b'),(4,256,48,48,1024,23,'b'),(2,512,24,24,2048,3,'h')]:\n"
        "    g = torch.Generator(device='cuda:0').manual_seed(3)\n"
        "    dtype = torch.bfloat16 if dt == 'b' else torch.float16\n"
        "    xs = [torch.randn(B,H,W,C,device='cuda:0',generator=g).to(dtype).permute(0,3,1,2) for _ in range(n)]\n"
        "    dys = [(torch.randn(B,H,W,N,device='cuda:0',generator=g) * (0.25 if dt == 'h' else 1.0)).to(dtype).permute(0,3,1,2) for _ in range(n)]\n"
        "    a = t._wgrad_grouped(xs,dys,N,C,1,1,0,1) if n > 1 else [t._wgrad_single(xs[0],dys[0],N,C,1,1,0,1)]\n"
        "    b = t._wgrad_grouped(xs,dys,N,C,1,1,0,1) if n > 1 else [t._wgrad_single(xs[0],dys[0],N,C,1,1,0,1)]\n"
        "    for i in range(n):\n"
        "        assert torch.equal(a[i], b[i]) and torch.isfinite(a[i]).all(), (C, N, H, W, i)\n"
        "        if i in (0, n - 1):\n"
        "            ref = torch.nn.grad.conv2d_weight(xs[i].float().cpu(), (N,C,1,1), dys[i].float().cpu(), 1, 0, 1)\n"
        "            assert t.relerr(a[i], ref) < 1e-4, (C, N, H, W, i, t.relerr(a[i], ref))\n"
        "print('ok')\n") % os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"MRFP_WGRAD1": "2"}, {"MRFP_WGRAD1": "0"}):
        env = dict(os.environ, PYTHONPATH=root, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stdout[-500:], r.stderr[-1500:])


def test_wgrad_256x128_tile_in_subprocess():
    """conv_wgrad_kernel<..., TMB = 4>: the 256 x 128 weight-gradient tile (16-bit LDS-DMA kernels, N % 256 == 0) forced wherever
    it is legal (MRFP_WGRAD_BIG=2; the switch is read once per process) -- single and grouped launches, dense (pointwise) and
    gather (3x3, dilated, strided) forms, ragged M -- against torch's convolution gradient of the same rounded operands and
    bit-identical between two launches; and the same shapes with the tile switched off (MRFP_WGRAD_BIG=0)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, %r)\n"
        "import test_conv_gpu as t\n"
        "from mrfp_amd import conv\n"
        "for (B,C,H,W,N,k,st,pad,dil,n) in [(4,256,48,48,1024,1,1,0,1,5),(4,1024,48,48,256,1,1,0,1,3),(4,256,48,48,256,3,1,1,1,4),(3,128,33,31,256,3,2,1,1,2),"
        "(2,512,24,24,512,3,1,2,2,1),(2,256,40,36,256,1,1,0,1,1),(2,64,96,96,256,1,1,0,1,1)]:\n"
        "    g = torch.Generator(device='cuda:0').manual_seed(3)\n"
        "    Ho, Wo = conv._out_size(H,k,st,pad,dil), conv._out_size(W,k,st,pad,dil)\n"
        "    xs = [torch.randn(B,H,W,C,device='cuda:0',generator=g).bfloat16().permute(0,3,1,2) for _ in range(n)]\n"
        "    dys = [torch.randn(B,Ho,Wo,N,device='cuda:0',generator=g).bfloat16().permute(0,3,1,2) for _ in range(n)]\n"
        "    a = t._wgrad_grouped(xs,dys,N,C,k,st,pad,dil) if n > 1 else [t._wgrad_single(xs[0],dys[0],N,C,k,st,pad,dil)]\n"
        "    b = t._wgrad_grouped(xs,dys,N,C,k,st,pad,dil) if n > 1 else [t._wgrad_single(xs[0],dys[0],N,C,k,st,pad,dil)]\n"
        "    for i in range(n):\n"
        "        assert torch.equal(a[i], b[i]) and torch.isfinite(a[i]).all()\n"
        "        ref = torch.nn.grad.conv2d_weight(xs[i].float().cpu(), (N,C,k,k), dys[i].float().cpu(), st, pad, dil)\n"
        "        assert t.relerr(a[i], ref) < 1e-4, (C, N, k, i, t.relerr(a[i], ref))\n"
        "print('ok')\n") % os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({"MRFP_WGRAD_BIG": "2"}, {"MRFP_WGRAD_BIG": "0"}):
        env = dict(os.environ, PYTHONPATH=root, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (extra, r.stdout[-500:], r.stderr[-1500:])


def test_row_reuse_kernels_in_subprocess():
    """conv_igemm_kernel<..., RR = true> (3x3, stride 1, pad = dilation: one fill of a haloed pixel patch serves the three taps of
    a filter row) on every tile geometry it supports -- a tile inside one image row (W = 384, 192), whole rows per tile (W = 96,
    48, 16), dilation 2, batch > 1 -- forward + dgrad (the same kernel on the flipped pack) against torch, and repeated
    launches bit-identical.  MRFP_CONV_RR=3 routes every shape that fits to these kernels (read once per process; the default
    mode keeps them to the long-K layers where they pay)."""
    import os
    import subprocess
    import sys
    code = (
        "import torch, torch.nn.functional as F\n"
        "from mrfp_amd import conv\n"
        "for (B,Cin,H,W,Cout,pad) in [(2,128,192,192,256,1),(2,64,96,96,128,1),(3,256,48,48,256,1),(2,128,48,48,128,2),(1,64,384,384,128,1),"
        "(2,64,192,384,192,1),(4,64,48,16,128,1),(2,192,96,192,320,1),(1,64,24,32,128,2),"
        # N <= 64: the 384 x 64 tile of four stacked waves (round 5) -- one image row per tile, 2 / 4 / 8 rows per tile, dilation 2,
        # N < 64; their dgrads (C <= 64 -> N = Cin) run the 192 x 128 kernels again
        "(1,128,384,384,64,1),(2,64,192,192,64,1),(3,64,96,96,64,2),(2,64,48,48,64,1),(2,128,192,192,48,1),(1,64,384,384,64,2)]:\n"
        "    g = torch.Generator().manual_seed(1)\n"
        "    x = torch.randn(B,Cin,H,W,generator=g).bfloat16().float(); w = (torch.randn(Cout,Cin,3,3,generator=g)*0.05).bfloat16().float()\n"
        "    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)\n"
        "    yc = F.conv2d(xc, wc, None, 1, pad, pad); gy = torch.randn(yc.shape, generator=g).bfloat16().float(); yc.backward(gy)\n"
        "    xd = x.cuda().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True); wd = w.cuda().requires_grad_(True)\n"
        "    yd = conv.conv2d(xd, wd, None, 1, pad, pad); yd.backward(gy.cuda().bfloat16().contiguous(memory_format=torch.channels_last))\n"
        "    rel = lambda a, b: ((a.double().cpu()-b.double()).abs().max()/b.double().abs().max()).item()\n"
        "    assert rel(yd, yc) < 1e-2 and rel(xd.grad, xc.grad) < 1e-2 and rel(wd.grad, wc.grad) < 2e-2, ((B,Cin,H,W,Cout,pad), rel(yd,yc), rel(xd.grad,xc.grad), rel(wd.grad,wc.grad))\n"
        "    st, cnt, npix = yd._mrfp_colstats[:3]; S = st.view(cnt, 2, -1).double().sum(0); yf = yd.detach().double()\n"
        "    assert npix == B*H*W and rel(S[0], yf.sum((0,2,3)).cpu()) < 1e-4 and rel(S[1], (yf*yf).sum((0,2,3)).cpu()) < 1e-4, ('stats', (B,Cin,H,W,Cout,pad))\n"
        "    with torch.no_grad():\n"
        "        ys = [conv.conv2d(xd.detach(), wd.detach(), None, 1, pad, pad).clone() for _ in range(4)]\n"
        "    assert all(torch.equal(ys[0], y) for y in ys[1:]), (B,Cin,H,W,Cout,pad)\n"
        "print('ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, MRFP_CONV_RR="3", MRFP_CONV_C64="0")      # (C = 64 would otherwise go to conv_c64.hip)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


C64_CASES = [
    # (B, C, H, W, N, dil, bias): one strip / several strips / a ragged last strip, odd sizes of the HRFP chain (231, 277, 321), both
    # dilation classes with odd H, spans that cross images and strips (B * strips * H not a multiple of the grid), every (C, N) pair
    (2, 64, 20, 18, 64, 1, True), (2, 64, 19, 23, 128, 2, True), (3, 64, 33, 31, 64, 1, False), (1, 64, 96, 384, 64, 1, False),
    (2, 64, 77, 231, 64, 1, True), (1, 64, 61, 277, 128, 2, True), (2, 64, 45, 321, 64, 2, False), (4, 64, 192, 192, 64, 1, False),
    (2, 64, 130, 200, 128, 1, False), (16, 64, 48, 48, 128, 2, True),
    # 128 input channels, a launch large enough for the default rule (>= 32 row strips per workgroup)
    (16, 128, 192, 192, 128, 1, False),
]
# 128 input channels (one wave per SIMD, two 64-channel halves per window pixel), N = 64 / 128 / 256 (two column groups): small shapes,
# which the default rule leaves to the implicit-GEMM tiles -- run in a child process with MRFP_CONV_C128=2 (the switch is read once)
C128_CASES = [
    (2, 128, 20, 18, 64, 1, True), (2, 128, 19, 23, 128, 2, False), (1, 128, 61, 277, 256, 2, True), (2, 128, 45, 200, 64, 2, False),
    (3, 128, 96, 96, 128, 1, False), (2, 128, 70, 130, 256, 1, False), (16, 128, 24, 24, 256, 1, True),
]


@pytest.mark.parametrize("case", C64_CASES)
def test_weight_stationary_c64_kernel(case):
    """csrc/conv_c64.hip (3x3, 64 / 128 input channels, dilation 1 / 2: weights in registers, rolling window of image rows) against torch:
    forward with bias, fused statistics (per image: the InstanceNorm form, and their total), the dgrad form, the addend form (a dgrad
    with a skip gradient), repeated launches bit-identical."""
    _c64_check(case)


def test_weight_stationary_kernel_128_channels_small_shapes_in_subprocess():
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import test_conv_gpu as t\nfor c in t.C128_CASES:\n    t._c64_check(c)\nprint('ok')\n"
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.path.join(root, "tests"), MRFP_CONV_C128="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def _c64_check(case):
    from mrfp_amd import conv, _lib
    from mrfp_amd._lib import call, ptr, stream
    B, C, H, W, N, dil, has_bias = case
    g = torch.Generator().manual_seed(H * 7 + W + C)
    x = torch.randn(B, C, H, W, generator=g).bfloat16().float()
    w = (torch.randn(N, C, 3, 3, generator=g) * (0.06 if C == 64 else 0.04)).bfloat16().float()
    b = torch.randn(N, generator=g) * 0.1 if has_bias else None
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yc = F.conv2d(xc, wc, b, 1, dil, dil)
    gy = torch.randn(yc.shape, generator=g).bfloat16().float()
    yc.backward(gy)
    xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    bd = b.to(DEV) if has_bias else None
    assert int(_lib.lib().mrfp_conv_stats_block_rows(_lib.BF16, B, H, W, C, N, 3, 3, H, W, 1, dil, dil, dil, 1)) < 0      # this kernel
    yd = conv.conv2d(xd, wd, bd, 1, dil, dil)
    assert relerr(yd, yc) < 1e-2
    if not has_bias:        # fused statistics: rows per image, and their total
        st = yd._mrfp_colstats
        rows, nblk, rb = st[3], st[4], st[5]
        assert rb < 0 and B * (-rb) == nblk
        per = rows[:nblk * 2 * N].view(B, -rb, 2, N).double().sum(1).cpu()
        yf = yd.detach().double().cpu()
        assert relerr(per[:, 0], yf.sum((2, 3))) < 1e-4 and relerr(per[:, 1], (yf * yf).sum((2, 3))) < 1e-4
        tot = st[0].view(st[1], 2, N).double().sum(0).cpu()
        assert relerr(tot[0], yf.sum((0, 2, 3))) < 1e-4
    yd.backward(gy.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert relerr(xd.grad, xc.grad) < 1e-2 and relerr(wd.grad, wc.grad) < 2e-2
    with torch.no_grad():
        ys = [conv.conv2d(xd.detach(), wd.detach(), bd, 1, dil, dil).clone() for _ in range(3)]
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    # the addend form (a dgrad with a skip gradient), through the C ABI: y = conv(x) + addend
    pk = conv.get_pack(wd.detach(), None, torch.bfloat16, C, N)
    add = torch.randn(B, N, H, W, generator=g).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out = torch.empty_like(add)
    call("mrfp_conv_fwd", ptr(xd.detach()), ptr(pk.wf), None, ptr(out), _lib.BF16, B, H, W, C, N, N, 3, 3, H, W, 1, dil, dil, dil, 1,
         ptr(add), None, stream())
    ref = F.conv2d(x, w, None, 1, dil, dil) + add.float().cpu()
    assert relerr(out, ref) < 1e-2


# ---------------------------------------------------------------------------------------------------------------------
# Counted waits against the conservative build.  The asynchronous LDS-DMA rings (conv_pw.hip, and the
# weight-gradient kernels where they use one) order their transfers with counted `s_waitcnt vmcnt(N)`; an under-wait shows
# only when the race happens to fire.  libmrfp_hip_vm0.so is the same library with every counted wait replaced by vmcnt(0)
# (-DMRFP_VMCNT0=1, built by __graft_entry__.build()): every pointwise launch shape of the bench workload -- forward with
# fused statistics, dgrad, dgrad with a skip-gradient addend, weight gradient -- must give bit-identical results in both.
# ---------------------------------------------------------------------------------------------------------------------
_VM0_CODE = r"""
import hashlib, json, sys, torch
from mrfp_amd import conv
shapes = json.load(open(sys.argv[1]))
out = {}
def h(t):
    return hashlib.sha256(t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes()).hexdigest()
for (B, H, W, C, N) in shapes:
    g = torch.Generator(device='cuda').manual_seed(B + H + C + N)
    x = torch.empty(B, C, H, W, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).normal_(generator=g).requires_grad_(True)
    w = (torch.empty(N, C, 1, 1, device='cuda').normal_(generator=g) * 0.05).requires_grad_(True)
    gy = torch.empty(B, N, H, W, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).normal_(generator=g)
    gs = torch.empty(B, C, H, W, device='cuda', dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).normal_(generator=g)
    for rep in range(2):
        x.grad = None; w.grad = None
        y, xs = conv.conv2d(x, w, None, 1, 0, 1, want_skip=True)
        st = y._mrfp_colstats[0].clone()
        torch.autograd.backward([y, xs], [gy, gs])          # dgrad with the skip gradient as its epilogue addend + wgrad
        key = '%dx%dx%dx%d->%d' % (B, H, W, C, N)
        rec = [h(y), h(st), h(x.grad), h(w.grad)]
        x2 = x.detach().clone().requires_grad_(True)
        y2 = conv.conv2d(x2, w.detach(), None, 1, 0, 1)
        y2.backward(gy)                                       # plain dgrad
        rec.append(h(x2.grad))
        assert out.setdefault(key, rec) == rec, ('run-to-run difference', key)
torch.cuda.synchronize()
json.dump(out, open(sys.argv[2], 'w'))
"""


def test_counted_waits_equal_the_vmcnt0_build(tmp_path):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    vm0 = os.path.join(root, "mrfp_amd", "csrc", "libmrfp_hip_vm0.so")
    assert os.path.exists(vm0), "libmrfp_hip_vm0.so is not built (__graft_entry__.build() / mrfp_amd.build.build_vm0())"
    shapes = set()
    for name, a in json.load(open(os.path.join(root, "tools", "bench_conv_shapes.json"))):
        B, H, W, C, N, ldy, R, S, Ho, Wo, stride = a[:11]
        if name == "mrfp_conv_fwd" and R == 1 and S == 1 and stride == 1 and a[14] == 1 and H > 1 and C % 32 == 0 and N % 8 == 0:
            shapes.add((min(B, 4) if H * W >= 96 * 96 else B, H, W, C, N))      # (the big maps: 4 images are enough tiles)
    shapes = sorted(shapes)
    assert len(shapes) >= 8, shapes
    sf = tmp_path / "shapes.json"
    sf.write_text(json.dumps(shapes))
    res = []
    for tag, extra in (("default", {}), ("vm0", {"MRFP_HIP_LIB": vm0})):
        out = tmp_path / (tag + ".json")
        env = dict(os.environ, PYTHONPATH=root, **extra)
        env.pop("MRFP_HIP_LIB", None) if not extra else None
        r = subprocess.run([sys.executable, "-c", _VM0_CODE, str(sf), str(out)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (tag, r.stdout[-500:], r.stderr[-2000:])
        res.append(json.loads(out.read_text()))
    assert res[0].keys() == res[1].keys() and len(res[0]) == len(shapes)
    bad = [k for k in res[0] if res[0][k] != res[1][k]]
    assert not bad, bad


# (B, C, H, W, N): the long-K pointwise kernel (csrc/conv_pwk.hip: K = 512 / 1024 / 1280, weights in registers, X through an LDS ring)
PWK_CASES = [
    (16, 1024, 48, 48, 256),      # the layer-3 reduce convolution / the dgrad of the expand one: 6 tiles per workgroup, two panels
    (16, 1024, 49, 47, 256),      # ragged M (the last 48-row tile is partial)
    (11, 512, 70, 70, 136),       # K = 512 (two workgroups per CU), N ends inside a panel
    (16, 1280, 48, 48, 256),      # bot_aspp: K = 1280 (five K quarters, 320 weight registers)
    (16, 512, 48, 48, 2048),      # layer-4 expand: sixteen panels
]


@pytest.mark.parametrize("case", PWK_CASES)
def test_long_k_pointwise_kernel(case):
    """forward + fused statistics, plain dgrad-form launch, the addend form and the GATED addend form (through the C ABI), repeated
    launches bit-identical -- against torch."""
    from mrfp_amd import conv, _lib
    from mrfp_amd._lib import call, ptr, stream
    B, C, H, W, N = case
    g = torch.Generator().manual_seed(C + N + H)
    x = torch.randn(B, C, H, W, generator=g).bfloat16().float()
    w = (torch.randn(N, C, 1, 1, generator=g) * (2.0 / C) ** 0.5).bfloat16().float()
    xd = x.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(DEV).requires_grad_(True)
    yd = conv.conv2d(xd, wd, None, 1, 0, 1)
    yc = F.conv2d(x, w)
    assert relerr(yd, yc) < 1e-2
    st = yd._mrfp_colstats
    tot = st[0].view(st[1], 2, N).double().sum(0).cpu()
    yf = yd.detach().double().cpu()
    # (statistics of the fp32 accumulators against sums of the bf16-rounded stored values: zero-mean rounding errors)
    assert relerr(tot[0], yf.sum((0, 2, 3))) < 5e-3 and relerr(tot[1], (yf * yf).sum((0, 2, 3))) < 1e-3
    with torch.no_grad():
        ys = [conv.conv2d(xd.detach(), wd.detach(), None, 1, 0, 1).clone() for _ in range(3)]
    assert all(torch.equal(ys[0], y) for y in ys[1:])
    # addend and gated addend (the dgrad forms), through the C ABI
    pk = conv.get_pack(wd.detach(), None, torch.bfloat16, C, N)
    add = torch.randn(B, N, H, W, generator=g).to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out = torch.empty_like(add)
    call("mrfp_conv_fwd", ptr(xd.detach()), ptr(pk.wf), None, ptr(out), _lib.BF16, B, H, W, C, N, N, 1, 1, H, W, 1, 0, 0, 1, 1,
         ptr(add), None, stream())
    addf = add.float().cpu()
    assert relerr(out, yc + addf) < 1e-2
    bits = torch.randint(0, 256, (B * H * W * N // 8,), generator=g, dtype=torch.uint8)
    keep = ((bits.view(-1, 1) >> torch.arange(8, dtype=torch.uint8)) & 1).bool().view(B, H, W, N).permute(0, 3, 1, 2)
    out2 = torch.empty_like(add)
    maskd = bits.to(DEV)
    call("mrfp_conv_fwd_gated", ptr(xd.detach()), ptr(pk.wf), None, ptr(out2), _lib.BF16, B, H, W, C, N, N, 1, 1, H, W, 1, 0, 0, 1, 1,
         ptr(add), ptr(maskd), stream())
    assert relerr(out2, yc + addf * keep) < 1e-2
    # backward through the operator layer (dgrad + wgrad of this convolution)
    gy = torch.randn(yc.shape, generator=g).bfloat16().float()
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xc, wc).backward(gy)
    yd.backward(gy.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert relerr(xd.grad, xc.grad) < 1e-2 and relerr(wd.grad, wc.grad) < 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_batched_weight_pack_equals_the_per_layer_pack(dtype):
    """mrfp_pack_weights_batched (ONE launch after the optimizer step for every pack of the model; round 6: 16-byte stores for the
    16-bit packs) against mrfp_pack_weight (the per-layer kernel) -- forward pack [N][r][s][C] and flipped dgrad pack [C][r][s][N] bit
    for bit, pad rows / pad channels included: 3x3 and 1x1 filters, N = 19 -> 32 (final2), C = 3 -> 8 (network input), 304 -> 320
    (decoder concatenation), a 2x2 filter, dimensions that are not multiples of the 64 x 8 brick."""
    from mrfp_amd import conv
    epc = 4 if dtype == torch.float32 else 8
    g = torch.Generator().manual_seed(3)
    cases = [(256, 256, 3, 3, 256, 256), (1024, 256, 1, 1, 1024, 256), (19, 256, 1, 1, 32, 256), (64, 3, 3, 3, 64, 8),
             (256, 304, 3, 3, 256, 320), (72, 40, 2, 2, 72, 40), (100, 24, 3, 3, 104, 24), (48, 1280, 1, 1, 48, 1280)]
    todo, ref = [], []
    for (N, C, R, S, Nphys, Cphys) in cases:
        Nphys, Cphys = (Nphys + epc - 1) // epc * epc, (Cphys + epc - 1) // epc * epc
        w = torch.randn(N, C, R, S, generator=g).to(DEV).requires_grad_(True)
        pk = conv.get_pack(w, None, dtype, Cphys, Nphys)          # per-layer kernel
        ref.append((pk.wf.clone(), pk.wd.clone()))
        pk.wf.fill_(7.0)
        if Cphys == C:
            pk.wd.fill_(7.0)                                      # (pad input-channel rows of wd are zero by allocation and never written)
        else:
            pk.wd.view(Cphys, -1)[:C].fill_(7.0)
        todo.append(((dtype, Cphys, Nphys, w.data_ptr(), 0), pk, w))
    conv._batched_repack(todo, "test_pack_%s" % dtype)
    torch.cuda.synchronize()
    for (case, (_, pk, _), (wf, wd)) in zip(cases, todo, ref):
        assert torch.equal(pk.wf, wf), ("wf", case)
        assert torch.equal(pk.wd, wd), ("wd", case)
