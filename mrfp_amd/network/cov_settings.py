"""Covariance bookkeeping of the instance-whitening losses (ISW / IRW) -- reference network/cov_settings.py:8-107 and
its call sites network/deepv3.py:455-475 (construction), 534-545 (variance of the covariances over a batch of
augmented pairs), 561-567 (the loss).

The arithmetic on activations runs on the HIP kernels: the per-image channel Gram matrix is the MFMA "reduce over pixels"
GEMM (ops.channel_gram -> mrfp_conv_wgrad) and its backward a per-image 1x1 implicit GEMM (instance_whitening.py).  What
is left here is C x C bookkeeping on the host side of the boundary: masks, the running variance matrix, the top-k
selection, and the 1-D k-means that splits "insensitive" from "sensitive" covariances (mrfp_kmeans1d in
csrc/hostmath.hip; the reference calls the un-vendored `kmeans1d` package there).

PARITY: get_covariance_matrix / instance_whitening_loss are pinned to the reference (tests/golden/whitening.npz).  The
classes below are NOT pinned: reference cov_settings.py imports `kmeans1d` (absent from this image) at module level and
builds its matrices with `.cuda()`, so it can neither be imported nor run in the build container; they are checked against
the numpy restatement in oracle/mrfp_oracle.py (isw_* functions) only."""
import ctypes

import numpy as np
import torch

from .. import _lib
from .instance_whitening import get_covariance_matrix, instance_whitening_loss


def _device(device):
    if device is not None:
        return torch.device(device)
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def kmeans1d_cluster(values, k):
    """(labels, centroids) of the optimal 1-D k-means, clusters numbered by ascending centroid (`kmeans1d.cluster`)."""
    x = np.ascontiguousarray(np.asarray(values, dtype=np.float64).reshape(-1))
    k = int(min(k, x.size))
    labels = np.empty(x.size, dtype=np.int32)
    cent = np.empty(k, dtype=np.float64)
    _lib.call("mrfp_kmeans1d", x.ctypes.data_as(ctypes.c_void_p), x.size, k, labels.ctypes.data_as(ctypes.c_void_p),
              cent.ctypes.data_as(ctypes.c_void_p))
    return labels, cent


def make_cov_index_matrix(dim):
    """Symmetric matrix numbering the off-diagonal pairs 1 .. dim(dim-1)/2, zero diagonal (reference cov_settings.py:8-14)."""
    i = torch.arange(dim).unsqueeze(1)
    j = torch.arange(dim).unsqueeze(0)
    lo, hi = torch.minimum(i, j), torch.maximum(i, j)
    # pair (lo, hi), lo < hi, row-major over the strict upper triangle, counted from 1: lo*dim - lo(lo+1)/2 + (hi - lo)
    idx = lo * dim - (lo * (lo + 1)) // 2 + (hi - lo)
    return torch.where(i == j, torch.zeros_like(idx), idx)


class CovMatrix_ISW:
    """Instance SELECTIVE whitening: the mask keeps the covariance entries whose variance across photometric
    augmentations is high (reference cov_settings.py:17-89)."""

    def __init__(self, dim, relax_denom=0, clusters=50, device=None):
        self.dim = dim
        self.device = _device(device)
        self.i = torch.eye(dim, dim, device=self.device)
        self.reversal_i = torch.ones(dim, dim, device=self.device).triu(diagonal=1)
        self.num_off_diagonal = torch.sum(self.reversal_i)
        self.num_sensitive = 0
        self.var_matrix = None
        self.count_var_cov = 0
        self.mask_matrix = None
        self.clusters = clusters
        self.margin = 0 if relax_denom == 0 else self.num_off_diagonal // relax_denom

    def get_eye_matrix(self):
        return self.i, self.reversal_i

    def get_mask_matrix(self, mask=True):
        if self.mask_matrix is None:
            self.set_mask_matrix()
        return self.i, self.mask_matrix, 0, self.num_sensitive

    def reset_mask_matrix(self):
        self.mask_matrix = None

    def set_mask_matrix(self):
        self.var_matrix = self.var_matrix / self.count_var_cov
        var_flatten = torch.flatten(self.var_matrix)
        if self.margin == 0:           # k-means over the variances: cluster 0 = insensitive, 1 .. k-1 = sensitive
            labels, _ = kmeans1d_cluster(var_flatten.detach().double().cpu().numpy(), self.clusters)
            num_sensitive = int(var_flatten.numel() - int((labels == 0).sum()))
        else:
            num_sensitive = int(self.num_off_diagonal - self.margin)
        _, indices = torch.topk(var_flatten, k=num_sensitive)
        mask_matrix = torch.zeros(self.dim * self.dim, device=self.device)
        mask_matrix[indices] = 1
        if self.mask_matrix is not None:
            self.mask_matrix = (self.mask_matrix.int() & mask_matrix.view(self.dim, self.dim).int()).float()
        else:
            self.mask_matrix = mask_matrix.view(self.dim, self.dim)
        self.num_sensitive = torch.sum(self.mask_matrix)
        self.var_matrix = None
        self.count_var_cov = 0

    def set_variance_of_covariance(self, var_cov):
        self.var_matrix = var_cov if self.var_matrix is None else self.var_matrix + var_cov
        self.count_var_cov += 1


class CovMatrix_IRW:
    """Instance RELAXED whitening: every off-diagonal entry, with a margin (reference cov_settings.py:91-107)."""

    def __init__(self, dim, relax_denom=0, device=None):
        self.dim = dim
        self.device = _device(device)
        self.i = torch.eye(dim, dim, device=self.device)
        self.reversal_i = torch.ones(dim, dim, device=self.device).triu(diagonal=1)
        self.num_off_diagonal = torch.sum(self.reversal_i)
        self.margin = 0 if relax_denom == 0 else self.num_off_diagonal // relax_denom

    def get_mask_matrix(self):
        return self.i, self.reversal_i, self.margin, self.num_off_diagonal


def build_cov_matrix_layers(wt_layer, in_channel_list, relax_denom=0, clusters=50, device=None):
    """One CovMatrix per whitened stage: wt_layer[i] == 1 -> IRW, == 2 -> ISW (reference network/deepv3.py:455-465)."""
    layers, kinds = [], []
    for i, w in enumerate(wt_layer):
        if w == 1:
            layers.append(CovMatrix_IRW(dim=in_channel_list[i], relax_denom=relax_denom, device=device))
            kinds.append(w)
        elif w == 2:
            layers.append(CovMatrix_ISW(dim=in_channel_list[i], relax_denom=relax_denom, clusters=clusters, device=device))
            kinds.append(w)
    return layers, kinds


def covariance_statistics(w_arr, cov_matrix_layer):
    """`cal_covstat` pass (reference network/deepv3.py:534-545): for every whitened feature map of a batch made of an image
    and its photometric transform, the variance over the batch of each off-diagonal covariance, accumulated in the layer."""
    for index, f_map in enumerate(w_arr):
        eye, reverse_eye = cov_matrix_layer[index].get_eye_matrix()
        f_cor, _ = get_covariance_matrix(f_map.detach(), eye=eye)
        off_diag_elements = f_cor * reverse_eye
        cov_matrix_layer[index].set_variance_of_covariance(torch.var(off_diag_elements, dim=0))
    return 0


def whitening_loss(w_arr, cov_matrix_layer):
    """`wt_loss` of reference network/deepv3.py:561-567: mean over the whitened stages of instance_whitening_loss."""
    wt_loss = torch.zeros(1, device=w_arr[0].device)
    for index, f_map in enumerate(w_arr):
        eye, mask_matrix, margin, num_remove_cov = cov_matrix_layer[index].get_mask_matrix()
        wt_loss = wt_loss + instance_whitening_loss(f_map, eye, mask_matrix, margin, num_remove_cov)
    return wt_loss / len(w_arr)
