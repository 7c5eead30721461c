"""reference network/instance_whitening.py:5-39 on the HIP kernels."""
import torch
import torch.nn as nn

from .mynn import HipInstanceNorm2d


class InstanceWhitening(nn.Module):
    """InstanceNorm2d(affine=False) returning (x, w) -- reference instance_whitening.py:5-16."""

    def __init__(self, dim):
        super().__init__()
        self.instance_standardization = HipInstanceNorm2d(dim, affine=False)

    def forward(self, x):
        x = self.instance_standardization(x)
        return x, x


def get_covariance_matrix(f_map, eye=None):
    """reference instance_whitening.py:30-39: bmm(f, f^T)/(HW-1) + eps*I  -> ([B,C,C], B)."""
    from .. import ops
    eps = 1e-5
    B, C, H, W = f_map.shape
    if eye is None:
        eye = torch.eye(C, device=f_map.device)
    f_cor = ops.channel_gram(f_map).div(H * W - 1) + eps * eye
    return f_cor, B


def instance_whitening_loss(f_map, eye, mask_matrix, margin, num_remove_cov):
    """reference instance_whitening.py:19-27."""
    f_cor, B = get_covariance_matrix(f_map, eye=eye)
    f_cor_masked = f_cor * mask_matrix
    off_diag_sum = torch.sum(torch.abs(f_cor_masked), dim=(1, 2), keepdim=True) - margin
    loss = torch.clamp(torch.div(off_diag_sum, num_remove_cov), min=0)
    return torch.sum(loss) / B
