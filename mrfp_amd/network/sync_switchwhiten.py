"""Switchable whitening (the `wt_layer` code 5 option): reference network/sync_switchwhiten.py:9-223 and its
single-process twin network/switchwhiten.py:7-183.

Split of the work: everything that touches the [N,C,H,W] activation runs on the HIP kernels -- for the reference's
group size 16 the dedicated group passes of csrc/whiten.hip (sums + 16x16 second moments in one read; folded whitening
matrix and offset in one read / one write; ops.group_whiten), otherwise plane means (statistics kernel), the full
Gram (MFMA "reduce over pixels" GEMM, ops.channel_gram) and a 1x1 implicit GEMM per image.  The 16x16 algebra in between
(softmax blend of batch / instance statistics, trace normalisation, T Newton-Schulz steps, affine folding) is a
few KB of data and is expressed with torch ops so that autograd provides exactly the reference's backward.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.nn.parameter import Parameter

from .. import ops


def _world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


class _AllReduceMean(torch.autograd.Function):
    """all_reduce(sum)/world in forward and in backward -- SyncMeanCov of reference sync_switchwhiten.py:20-26, 44-45."""

    @staticmethod
    def forward(ctx, t):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return t
        t = t.clone()
        dist.all_reduce(t)
        return t / dist.get_world_size()

    @staticmethod
    def backward(ctx, g):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return g
        g = g.clone()
        dist.all_reduce(g)
        return g / dist.get_world_size()


class SyncSwitchWhiten2d(nn.Module):
    """Same constructor, parameters, buffers and state_dict keys as the reference class."""

    def __init__(self, num_features, num_pergroup=16, sw_type=2, T=5, tie_weight=False, eps=1e-5, momentum=0.99,
                 affine=True):
        super().__init__()
        if sw_type not in [2, 3, 4, 5]:
            raise ValueError("sw_type should be in [2, 3, 4, 5], but got {}".format(sw_type))
        assert num_features % num_pergroup == 0
        self.num_features, self.num_pergroup = num_features, num_pergroup
        self.num_groups = num_features // num_pergroup
        self.sw_type, self.T, self.tie_weight, self.eps, self.momentum, self.affine = sw_type, T, tie_weight, eps, momentum, affine
        self.sw_mean_weight = Parameter(torch.ones(sw_type))
        if not tie_weight:
            self.sw_var_weight = Parameter(torch.ones(sw_type))
        else:
            self.register_parameter("sw_var_weight", None)
        if affine:
            self.weight = Parameter(torch.ones(num_features))
            self.bias = Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        self.register_buffer("running_mean", torch.zeros(self.num_groups, num_pergroup, 1))
        self.register_buffer("running_cov", torch.eye(num_pergroup).unsqueeze(0).repeat(self.num_groups, 1, 1))
        self.use_group_kernels = True        # False: the generic (any group size) passes, kept for cross-checks
        self.reset_parameters()

    def reset_parameters(self):
        self.running_mean.zero_()
        self.running_cov.zero_()          # the reference zeroes it too (sync_switchwhiten.py:124)
        nn.init.ones_(self.sw_mean_weight)
        if not self.tie_weight:
            nn.init.ones_(self.sw_var_weight)
        if self.affine:
            nn.init.ones_(self.weight)
            nn.init.zeros_(self.bias)

    def __repr__(self):
        return ("{name}({num_features}, num_pergroup={num_pergroup}, sw_type={sw_type}, T={T}, tie_weight={tie_weight}, "
                "eps={eps}, momentum={momentum}, affine={affine})".format(name=self.__class__.__name__, **self.__dict__))

    def forward(self, x):
        N, C, H, W = x.shape
        c, g = self.num_pergroup, self.num_groups
        hw = float(H * W)
        if c == 16 and C <= 1024 and self.use_group_kernels:
            # dedicated group passes (csrc/whiten.hip): sums and 16x16 second moments in ONE read of x, the folded
            # whitening matrix + offset in one read / one write; the backward pass reads (dy, x) twice and writes dx once
            def algebra(s, M):
                sq = torch.diagonal(M, dim1=-2, dim2=-1).sum((1, 2))
                wm, shift = self._transform(s / hw, M, sq, N, C, hw)
                return wm, shift
            params = [p for p in (self.sw_mean_weight, self.sw_var_weight, self.weight, self.bias) if p is not None]
            # training, one process: the ~200 launches of the algebra replay as two captured hipGraphs (ops._AlgebraGraph); with several
            # ranks the algebra contains the two all-reduces of the batch statistics and stays eager
            graph = None
            if self.training and x.is_cuda and torch.is_grad_enabled() and ops.whiten_graph_enabled() and _world_size() == 1:
                graph = (self, (N, C, H, W, self.sw_type, self.T, self.tie_weight, self.affine, float(self.eps), float(self.momentum)),
                         (self.running_mean, self.running_cov))
            return ops.group_whiten(x, algebra, params, graph)
        # other group sizes: plane means + the full C x C Gram on the MFMA "reduce over pixels" GEMM, application as a
        # per-image 1x1 implicit GEMM
        mu = ops.plane_mean(x)                                   # [N, C]
        gram = ops.channel_gram(x)                               # [N, C, C] = sum_p x x^T
        idx = torch.arange(C, device=x.device).view(g, c)
        blocks = gram[:, idx.unsqueeze(-1), idx.unsqueeze(-2)]   # [N, g, c, c] diagonal blocks
        sq = torch.diagonal(gram, dim1=1, dim2=2).sum(1)
        wm, shift = self._transform(mu, blocks, sq, N, C, hw)
        full = torch.zeros(N, C, C, device=x.device, dtype=torch.float32)
        full[:, idx.unsqueeze(-1), idx.unsqueeze(-2)] = wm
        return ops.per_image_matmul(x, full, shift)

    def _transform(self, mu, blocks, sq, N, C, hw):
        """(plane means [N,C], sum_p x_g x_g^T [N,g,c,c], sum of squares per image [N]) -> folded whitening matrix
        [N,g,c,c] and offset [N,C]: the 16x16 algebra of reference sync_switchwhiten.py:161-223, a few KB, in torch ops
        so that autograd yields the reference's backward."""
        c, g = self.num_pergroup, self.num_groups
        dev = mu.device
        mean_in = mu.view(N, g, c, 1)
        m2_in = blocks / hw                                      # E[x x^T] per image
        cov_in = m2_in - mean_in @ mean_in.transpose(-1, -2)     # centred instance covariance (/HW, biased)

        if self.training:
            mean_bn = _AllReduceMean.apply(mean_in.mean(0))      # [g, c, 1]
            # batch covariance around the batch mean: E_b[E[x x^T]] - mean_bn mean_bn^T, then averaged over ranks
            # exactly as the reference (each rank centres with the GLOBAL mean before its all_reduce)
            cov_bn = m2_in.mean(0) - mean_in.mean(0) @ mean_bn.transpose(-1, -2) - mean_bn @ mean_in.mean(0).transpose(-1, -2) \
                + mean_bn @ mean_bn.transpose(-1, -2)
            cov_bn = _AllReduceMean.apply(cov_bn)
            with torch.no_grad():
                self.running_mean.mul_(self.momentum).add_((1 - self.momentum) * mean_bn)
                self.running_cov.mul_(self.momentum).add_((1 - self.momentum) * cov_bn)
        else:
            mean_bn, cov_bn = self.running_mean, self.running_cov
        mean_bn = mean_bn.unsqueeze(0).expand(N, g, c, 1)
        cov_bn = cov_bn.unsqueeze(0).expand(N, g, c, c)
        eye = torch.eye(c, device=dev, dtype=torch.float32).view(1, 1, c, c)

        mean_weight = torch.softmax(self.sw_mean_weight, 0)
        var_weight = mean_weight if self.tie_weight else torch.softmax(self.sw_var_weight, 0)
        if self.sw_type in (3, 5):
            # layer statistics over (C, H, W) per sample: mean and UNBIASED variance (x.var(-1))
            n_el = float(C) * hw
            mean_ln = mu.mean(1).view(N, 1, 1, 1)
            var_ln = (sq.view(N, 1, 1, 1) - n_el * mean_ln * mean_ln) / (n_el - 1.0)
            var_ln = var_ln * eye
        if self.sw_type == 2:
            mean = mean_weight[0] * mean_bn + mean_weight[1] * mean_in
            cov = var_weight[0] * cov_bn + var_weight[1] * cov_in + self.eps * eye
        elif self.sw_type == 3:
            mean = mean_weight[0] * mean_bn + mean_weight[1] * mean_in + mean_weight[2] * mean_ln
            cov = var_weight[0] * cov_bn + var_weight[1] * cov_in + var_weight[2] * var_ln + self.eps * eye
        elif self.sw_type == 5:
            var_bn = torch.diag_embed(torch.diagonal(cov_bn, dim1=-2, dim2=-1))
            var_in = torch.diag_embed(torch.diagonal(cov_in, dim1=-2, dim2=-1))
            mean = (mean_weight[0] + mean_weight[2]) * mean_bn + (mean_weight[1] + mean_weight[3]) * mean_in + \
                mean_weight[4] * mean_ln
            cov = var_weight[0] * cov_bn + var_weight[1] * cov_in + var_weight[0] * var_bn + var_weight[1] * var_in + \
                var_weight[4] * var_ln + self.eps * eye
        else:
            raise NotImplementedError("sw_type 4 is accepted by the reference constructor but has no forward branch")

        # Newton-Schulz inverse square root (reference sync_switchwhiten.py:206-215)
        if c == 16 and self.use_group_kernels and self.T <= 8 and cov.is_cuda:
            wm = ops.group_isqrt(cov, self.T)                     # one launch forward, one backward (csrc/whiten.hip)
        else:
            rTr = 1.0 / torch.diagonal(cov, dim1=-2, dim2=-1).sum(-1).view(N, g, 1, 1)
            cov_n = cov * rTr
            P = eye.expand(N, g, c, c)
            for _ in range(self.T):
                P = 1.5 * P - 0.5 * (P @ P @ P) @ cov_n
            wm = P * rTr.sqrt()                                   # cov^{-1/2}, [N, g, c, c]

        # fold mean and affine into one per-image block-diagonal matrix + offset
        if self.affine:
            wm = wm * self.weight.view(1, g, c, 1)
        shift = -(wm @ mean).view(N, C)
        if self.affine:
            shift = shift + self.bias.view(1, C)
        return wm, shift


class SwitchWhiten2d(SyncSwitchWhiten2d):
    """reference network/switchwhiten.py: the same maths without the cross-rank exchange (world size 1)."""
