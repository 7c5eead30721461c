"""Whitening options of the trunk (`wt_layer` codes 1, 2, 5) against vectors produced by the
reference's own classes (tests/golden/whitening.npz: network/switchwhiten.py SwitchWhiten2d train fwd+bwd / eval,
network/instance_whitening.py InstanceWhitening, get_covariance_matrix, instance_whitening_loss)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "whitening.npz"))


def relerr(a, b):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _x():
    return torch.from_numpy(G["x"]).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)


def test_switch_whiten_train_eval_matches_reference():
    from mrfp_amd.network.sync_switchwhiten import SyncSwitchWhiten2d
    sw = SyncSwitchWhiten2d(32, num_pergroup=16, sw_type=2, T=5, tie_weight=False, eps=1e-5, momentum=0.99, affine=True).to(DEV)
    with torch.no_grad():
        sw.weight.copy_(torch.from_numpy(G["weight"]))
        sw.bias.copy_(torch.from_numpy(G["bias"]))
        sw.sw_mean_weight.copy_(torch.tensor([0.3, -0.2]))
        sw.sw_var_weight.copy_(torch.tensor([-0.1, 0.4]))
    sw.train()
    x = _x()
    y = sw(x)
    y.backward(torch.from_numpy(G["gy"]).to(DEV).contiguous(memory_format=torch.channels_last))
    assert relerr(y, G["y_train"]) < 2e-4
    assert relerr(x.grad, G["gx"]) < 2e-3
    assert relerr(sw.weight.grad, G["g_weight"]) < 2e-3 and relerr(sw.bias.grad, G["g_bias"]) < 2e-4
    assert relerr(sw.sw_mean_weight.grad, G["g_mean_w"]) < 5e-3 and relerr(sw.sw_var_weight.grad, G["g_var_w"]) < 5e-3
    assert relerr(sw.running_mean, G["running_mean"]) < 1e-4 and relerr(sw.running_cov, G["running_cov"]) < 1e-4
    sw.eval()
    with torch.no_grad():
        ye = sw(x.detach())
    assert relerr(ye, G["y_eval"]) < 2e-4
    assert sorted(sw.state_dict().keys()) == sorted(["sw_mean_weight", "sw_var_weight", "weight", "bias", "running_mean", "running_cov"])


def test_instance_whitening_and_covariance_loss():
    from mrfp_amd.network import instance_whitening as iw
    x = _x()
    mod = iw.InstanceWhitening(32).to(DEV)
    y, w = mod(x)
    assert relerr(y, G["iw_y"]) < 2e-5 and w is y
    cov, B = iw.get_covariance_matrix(x.detach(), eye=torch.eye(32, device=DEV))
    assert B == 3 and relerr(cov, G["iw_cov"]) < 2e-5
    mask = torch.triu(torch.ones(32, 32, device=DEV), diagonal=1)
    loss = iw.instance_whitening_loss(x, torch.eye(32, device=DEV), mask, margin=0.0, num_remove_cov=mask.sum())
    assert abs(loss.item() - float(G["iw_loss"])) / float(G["iw_loss"]) < 2e-5
    # gradient of the covariance loss against torch autograd on the CPU restatement of the same expression
    xc = torch.from_numpy(G["x"]).requires_grad_(True)
    f = xc.view(3, 32, -1)
    cc = torch.bmm(f, f.transpose(1, 2)).div(12 * 10 - 1) + 1e-5 * torch.eye(32)
    lc = torch.clamp((cc * mask.cpu()).abs().sum((1, 2), keepdim=True) / mask.sum().cpu(), min=0).sum() / 3
    lc.backward()
    loss.backward()
    assert relerr(x.grad, xc.grad) < 2e-5


def test_resnet_with_wt_layer_5_and_1_runs():
    """The trunk accepts the whitening codes of the reference's wt_layer argument (Resnet.py:525-549, 166-190)."""
    from mrfp_amd.network import Resnet
    from mrfp_amd.config import cfg
    cfg.MODEL.ACT_DTYPE = torch.float32
    for wt in ([0, 0, 5, 5, 0, 0, 0], [0, 0, 1, 2, 0, 0, 0], [0, 0, 3, 0, 0, 0, 0]):
        net = Resnet.resnet18(pretrained=False, wt_layer=wt).to(DEV).train()
        out = net(torch.rand(2, 3, 64, 64, device=DEV) * 255)
        assert tuple(out.shape) == (2, 512, 2, 2) and torch.isfinite(out).all()
        out.sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in net.named_parameters() if not n.startswith("fc"))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2), (torch.float16, 2e-3)])
@pytest.mark.parametrize("shape", [(2, 32, 5, 7), (3, 64, 17, 9), (2, 48, 12, 20), (2, 256, 24, 24), (1, 1024, 6, 5)])
def test_group_moments_and_apply_ops(dtype, tol, shape):
    """csrc/whiten.hip against plain torch: sums + 16x16 second moments per group, the group matrix application, and
    both backward passes."""
    from mrfp_amd import ops
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(*shape, generator=g) + 0.2).to(dtype).float()
    Wm = torch.randn(B, C // 16, 16, 16, generator=g) * 0.3
    sh = torch.randn(B, C, generator=g)
    gy, gs, gM = torch.randn(*shape, generator=g), torch.randn(B, C, generator=g), torch.randn(B, C // 16, 16, 16, generator=g)
    # torch reference
    xr = x.clone().requires_grad_(True)
    Wr, sr = Wm.clone().requires_grad_(True), sh.clone().requires_grad_(True)
    xg = xr.view(B, C // 16, 16, H * W)
    s_ref, M_ref = xr.sum((2, 3)), xg @ xg.transpose(-1, -2)
    y_ref = (Wr @ xg).view(B, C, H, W) + sr.view(B, C, 1, 1)
    ((s_ref * gs).sum() + (M_ref * gM).sum() + (y_ref * gy.to(dtype).float()).sum()).backward()
    # HIP
    xd = x.to(DEV, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    Wd, sd = Wm.to(DEV).requires_grad_(True), sh.to(DEV).requires_grad_(True)
    s, M = ops.group_moments(xd)
    y = ops.group_apply(xd, Wd, sd)
    assert relerr(s, s_ref) < tol and relerr(M, M_ref) < tol and relerr(y, y_ref) < tol
    ((s * gs.to(DEV)).sum() + (M * gM.to(DEV)).sum()).backward()
    g1 = xd.grad.clone()
    xd.grad = None
    y.backward(gy.to(DEV, dtype).contiguous(memory_format=torch.channels_last))
    xr2 = x.clone().requires_grad_(True)
    xg2 = xr2.view(B, C // 16, 16, H * W)
    ((xr2.sum((2, 3)) * gs).sum() + ((xg2 @ xg2.transpose(-1, -2)) * gM).sum()).backward()
    assert relerr(g1, xr2.grad) < tol                                     # through the moments
    assert relerr(xd.grad + g1.float(), xr.grad) < 2 * tol                # apply + moments = the joint reference gradient
    assert relerr(Wd.grad, Wr.grad) < tol and relerr(sd.grad, sr.grad) < tol


@pytest.mark.parametrize("sw_type,tie", [(2, False), (3, True), (5, False)])
def test_switch_whiten_group_kernels_match_generic_passes(sw_type, tie):
    """the fused group path (one node: moments -> 16x16 algebra -> apply) against the generic Gram / per-image GEMM path."""
    from mrfp_amd.network.sync_switchwhiten import SwitchWhiten2d
    torch.manual_seed(3)
    x0 = torch.randn(3, 64, 14, 10) * 1.5 + 0.3
    gy = torch.randn(3, 64, 14, 10)
    outs = []
    for fast in (True, False):
        torch.manual_seed(5)
        sw = SwitchWhiten2d(64, num_pergroup=16, sw_type=sw_type, T=5, tie_weight=tie).to(DEV)
        with torch.no_grad():
            sw.weight.uniform_(0.5, 1.5)
            sw.bias.uniform_(-0.5, 0.5)
            sw.sw_mean_weight.uniform_(-0.5, 0.5)
            if not tie:
                sw.sw_var_weight.uniform_(-0.5, 0.5)
        sw.use_group_kernels = fast
        sw.train()
        x = x0.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = sw(x)
        y.backward(gy.to(DEV).contiguous(memory_format=torch.channels_last))
        outs.append((y, x.grad, sw.weight.grad, sw.bias.grad, sw.sw_mean_weight.grad, sw.running_mean.clone(), sw.running_cov.clone()))
    for a, b in zip(*outs):
        assert relerr(a, b) < 5e-4


@pytest.mark.parametrize("sw_type,tie", [(2, False), (5, True)])
def test_switch_whiten_algebra_as_hipgraphs_is_bit_identical_to_the_eager_algebra(sw_type, tie, monkeypatch):
    """ops._AlgebraGraph: the 16x16 algebra of a training SwitchWhiten2d replayed as two captured hipGraphs (forward, autograd.grad) --
    the same kernels in the same order, so output, every gradient and the running statistics equal the eager path bit for bit over
    three steps with changing input and parameters (the warm-up runs of the capture must not have touched the running statistics);
    a second forward before the first backward falls back to the eager algebra and is still correct."""
    from mrfp_amd.network.sync_switchwhiten import SwitchWhiten2d
    res = []
    for graph in ("1", "0"):
        monkeypatch.setenv("MRFP_WHITEN_GRAPH", graph)
        torch.manual_seed(5)
        sw = SwitchWhiten2d(64, num_pergroup=16, sw_type=sw_type, T=5, tie_weight=tie).to(DEV).train()
        with torch.no_grad():
            sw.weight.uniform_(0.5, 1.5)
            sw.bias.uniform_(-0.5, 0.5)
            sw.sw_mean_weight.uniform_(-0.5, 0.5)
        out = []
        g = torch.Generator().manual_seed(9)
        for step in range(3):
            x = (torch.randn(4, 64, 12, 10, generator=g) * (1.0 + step) + 0.2).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            gy = torch.randn(4, 64, 12, 10, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
            for p in sw.parameters():
                p.grad = None
            y = sw(x)
            y.backward(gy)
            out += [y.detach().clone(), x.grad.clone(), sw.weight.grad.clone(), sw.bias.grad.clone(), sw.sw_mean_weight.grad.clone(),
                    sw.running_mean.clone(), sw.running_cov.clone()]
            with torch.no_grad():
                for p in sw.parameters():
                    p.sub_(0.05 * p.grad)
        # two forwards of the layer before any backward (the second one must not disturb the first one's backward)
        xa = torch.randn(4, 64, 12, 10, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        xb = torch.randn(4, 64, 12, 10, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ya, yb = sw(xa), sw(xb)
        (ya.float().square().sum() + yb.float().sum()).backward()
        out += [ya.detach().clone(), yb.detach().clone(), xa.grad.clone(), xb.grad.clone(), sw.running_cov.clone()]
        res.append(out)
        from mrfp_amd import ops
        assert (len(ops._ALG_GRAPHS.get(sw, {})) == 1) == (graph == "1"), "one algebra graph per (layer, shape) when the switch is on, none when off"
    for a, b in zip(*res):
        assert torch.equal(a, b)


def _sw_step(sw, x, gy):
    for p in sw.parameters():
        p.grad = None
    xx = x.clone().requires_grad_(True)
    y = sw(xx)
    y.backward(gy)
    return [y.detach().clone(), xx.grad.clone()] + [None if p.grad is None else p.grad.clone() for p in sw.parameters()]


def _sw_case(seed=5):
    from mrfp_amd.network.sync_switchwhiten import SwitchWhiten2d
    torch.manual_seed(seed)
    sw = SwitchWhiten2d(64, num_pergroup=16, sw_type=2, T=5).to(DEV).train()
    with torch.no_grad():
        sw.weight.uniform_(0.5, 1.5)
        sw.bias.uniform_(-0.5, 0.5)
        sw.sw_mean_weight.uniform_(-0.5, 0.5)
        sw.sw_var_weight.uniform_(-0.5, 0.5)
    g = torch.Generator().manual_seed(9)
    xs = [(torch.randn(4, 64, 12, 10, generator=g) * (1.0 + i) + 0.2).to(DEV).contiguous(memory_format=torch.channels_last) for i in range(4)]
    gys = [torch.randn(4, 64, 12, 10, generator=g).to(DEV).contiguous(memory_format=torch.channels_last) for _ in range(4)]
    return sw, xs, gys


def test_algebra_graph_follows_a_parameter_frozen_after_the_first_capture(monkeypatch):
    """the captured backward graph differentiates with respect to the parameters that required a gradient at capture
    time.  Freezing `weight` after the first step (fine-tuning) must not hand `bias` the gradient of `weight` (equal shapes), nor
    raise; unfreezing again must work too.  Compared against the eager algebra, bit for bit."""
    res = []
    for graph in ("1", "0"):
        monkeypatch.setenv("MRFP_WHITEN_GRAPH", graph)
        sw, xs, gys = _sw_case()
        out = _sw_step(sw, xs[0], gys[0])
        sw.weight.requires_grad_(False)
        out += [t for t in _sw_step(sw, xs[1], gys[1]) if t is not None]
        assert sw.weight.grad is None and sw.bias.grad is not None
        sw.sw_var_weight.requires_grad_(False)
        sw.weight.requires_grad_(True)
        out += [t for t in _sw_step(sw, xs[2], gys[2]) if t is not None]
        assert sw.sw_var_weight.grad is None and sw.weight.grad is not None
        res.append(out)
    assert len(res[0]) == len(res[1])
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_algebra_graph_is_released_by_a_forward_that_never_gets_its_backward(monkeypatch):
    """a training forward whose backward never comes (a dropped loss) must not leave the layer on the eager algebra:
    the graph is busy only while that forward's autograd node is alive.  The per-module cache is bounded."""
    from mrfp_amd import ops
    monkeypatch.setenv("MRFP_WHITEN_GRAPH", "1")
    sw, xs, gys = _sw_case()
    _sw_step(sw, xs[0], gys[0])
    (gr,) = ops._ALG_GRAPHS[sw].values()
    assert not gr.busy
    y = sw(xs[1].clone().requires_grad_(True))          # forward only
    assert gr.busy
    y2 = sw(xs[2].clone().requires_grad_(True))         # second forward while the first is pending: eager algebra, graph untouched
    assert gr.busy
    del y, y2                                           # the losses are dropped: nodes collected
    assert not gr.busy, "a forward without backward must release the graph when its autograd node dies"
    replays = []
    orig = gr.fwd.replay
    gr.fwd.replay = lambda: (replays.append(1), orig())[1]
    _sw_step(sw, xs[3], gys[3])
    assert replays == [1], "the next training step runs on the captured graph again"
    # bounded cache: more shapes than the cap
    for h in range(6, 6 + ops._ALG_GRAPH_CAP + 2):
        x = torch.randn(2, 64, h, 8, device=DEV).contiguous(memory_format=torch.channels_last)
        _sw_step(sw, x, torch.ones_like(x))
    assert len(ops._ALG_GRAPHS[sw]) <= ops._ALG_GRAPH_CAP


def test_algebra_graph_is_not_built_from_inside_a_backward_pass(monkeypatch):
    """a first forward issued from INSIDE the autograd engine (activation checkpointing recomputes the layer during
    backward) must not run the graph build there -- a capture begun inside the autograd machinery aborts the process on this
    stack.  The recomputation runs the eager algebra; results equal the plain eager run."""
    from torch.utils.checkpoint import checkpoint
    from mrfp_amd import ops
    res = []
    for graph in ("1", "0"):
        monkeypatch.setenv("MRFP_WHITEN_GRAPH", graph)
        sw, xs, gys = _sw_case()
        xx = xs[0].clone().requires_grad_(True)
        y = checkpoint(sw, xx, use_reentrant=False)     # saved-tensor hooks in the forward; recomputed in backward
        y.backward(gys[0])
        res.append([y.detach().clone(), xx.grad.clone(), sw.weight.grad.clone(), sw.sw_mean_weight.grad.clone()])
        # under saved-tensor hooks the layer runs the eager algebra in BOTH passes (the checkpoint compares what they save)
        assert len(ops._ALG_GRAPHS.get(sw, {})) == 0
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # ... and a layer whose FIRST contact is the recomputation itself (reentrant checkpoint: the first forward runs under no_grad)
    monkeypatch.setenv("MRFP_WHITEN_GRAPH", "1")
    sw, xs, gys = _sw_case()
    xx = xs[0].clone().requires_grad_(True)
    y = checkpoint(sw, xx, use_reentrant=True)
    assert len(ops._ALG_GRAPHS.get(sw, {})) == 0
    y.backward(gys[0])
    assert len(ops._ALG_GRAPHS.get(sw, {})) == 0, "no graph build on the autograd engine's thread"
    assert torch.equal(y.detach(), res[1][0]) and torch.equal(xx.grad, res[1][1])


def test_group_isqrt_matches_newton_schulz_autograd():
    """mrfp_group_isqrt_{fwd,bwd} against the reference's Newton-Schulz loop differentiated by torch autograd."""
    from mrfp_amd import ops
    g = torch.Generator().manual_seed(11)
    a = torch.randn(37, 16, 48, generator=g)
    cov0 = (a @ a.transpose(-1, -2) / 48 + 1e-3 * torch.eye(16)).to(DEV)
    gw = torch.randn(37, 16, 16, generator=g).to(DEV)
    for T in (1, 5, 8):
        cr = cov0.clone().requires_grad_(True)
        rTr = 1.0 / torch.diagonal(cr, dim1=-2, dim2=-1).sum(-1).view(-1, 1, 1)
        cn = cr * rTr
        P = torch.eye(16, device=DEV).expand(37, 16, 16)
        for _ in range(T):
            P = 1.5 * P - 0.5 * (P @ P @ P) @ cn
        ref = P * rTr.sqrt()
        (ref * gw).sum().backward()
        cd = cov0.clone().requires_grad_(True)
        out = ops.group_isqrt(cd, T)
        (out * gw).sum().backward()
        assert relerr(out, ref) < 2e-5 and relerr(cd.grad, cr.grad) < 2e-4


def test_isw_irw_workflow_on_a_trunk_against_the_restatement():
    """reference network/deepv3.py:534-545 (cal_covstat), 561-567 (wt_loss) with network/cov_settings.py on the whitened
    feature maps (w_arr) of a ResNet trunk built with wt_layer codes 2 (ISW) and 1 (IRW): covariance statistics, k-means
    mask and the loss from the HIP path against oracle/mrfp_oracle.py::isw_* on the same feature maps; the loss's gradient
    reaches the trunk's first convolution."""
    from mrfp_amd.config import cfg
    from mrfp_amd.network import Resnet, cov_settings as cs
    from oracle import mrfp_oracle as orc
    cfg.MODEL.ACT_DTYPE = torch.float32
    torch.manual_seed(5)
    net = Resnet.resnet18(pretrained=False, wt_layer=[0, 0, 2, 1, 0, 0, 0]).to(DEV).train()
    layers, kinds = cs.build_cov_matrix_layers([0, 0, 2, 1, 0, 0, 0], [0, 0, 64, 64, 128, 256, 512], relax_denom=0, clusters=3)
    assert kinds == [2, 1]

    def w_arr_of(x):
        from mrfp_amd import ops
        w_arr = []
        t = net.conv1(ops.as_activation(x))
        t = Resnet._norm_relu(net.bn1, net.wt_layer[2], t, w_arr)
        t = net.layer1([ops.max_pool_3x3_s2(t), w_arr])
        return t[1]
    g = torch.Generator().manual_seed(11)
    acc = None
    for _ in range(3):                                    # calibration: batches of (image, transformed image)
        img = torch.rand(1, 3, 64, 64, generator=g)
        x = torch.cat([img, (img * 1.3 - 0.1).clamp(0, 1)], 0).to(DEV) * 255
        with torch.no_grad():
            w_arr = w_arr_of(x)
        assert len(w_arr) == 2 and w_arr[0].shape[1] == 64
        cs.covariance_statistics(w_arr[:1], layers[:1])   # only ISW layers keep statistics (CovMatrix_IRW has no such method)
        v = orc.isw_variance_of_covariance(w_arr[0].float().cpu().numpy())
        acc = v if acc is None else acc + v
    assert relerr(layers[0].var_matrix, acc) < 1e-4
    var_dev = (layers[0].var_matrix / 3).cpu().numpy().astype(np.float64)
    eye, mask, margin, n_sens = layers[0].get_mask_matrix()
    np.testing.assert_array_equal(mask.cpu().numpy(), orc.isw_mask(var_dev, 3))
    assert 0 < float(n_sens) < 64 * 63 / 2

    x = (torch.rand(2, 3, 64, 64, generator=g) * 255).to(DEV)
    w_arr = w_arr_of(x)
    loss = cs.whitening_loss(w_arr, layers)
    ref = 0.0
    for f, layer in zip(w_arr, layers):
        e, m, mg, nr = layer.get_mask_matrix()
        ref += orc.isw_loss(f.detach().float().cpu().numpy(), m.cpu().numpy().astype(np.float64), float(mg), float(nr))
    assert abs(loss.item() - ref / 2) / (ref / 2) < 1e-4
    loss.backward()
    gw = net.conv1.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and gw.abs().max() > 0
