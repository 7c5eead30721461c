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
    cfg.MODEL.CONV_BACKEND, cfg.MODEL.ACT_DTYPE = "hip", torch.float32
    for wt in ([0, 0, 5, 5, 0, 0, 0], [0, 0, 1, 2, 0, 0, 0], [0, 0, 3, 0, 0, 0, 0]):
        net = Resnet.resnet18(pretrained=False, wt_layer=wt).to(DEV).train()
        out = net(torch.rand(2, 3, 64, 64, device=DEV) * 255)
        assert tuple(out.shape) == (2, 512, 2, 2) and torch.isfinite(out).all()
        out.sum().backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in net.named_parameters() if not n.startswith("fc"))
