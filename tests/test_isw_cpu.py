"""Host side of the ISW / IRW whitening losses (mrfp_amd/network/cov_settings.py, csrc/hostmath.hip::mrfp_kmeans1d) against
the numpy restatement in oracle/mrfp_oracle.py.  No GPU: the CovMatrix classes are C x C bookkeeping (built on the CPU
device here), the k-means is host code of the C-ABI library."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrfp_amd.network import cov_settings as cs   # noqa: E402
from oracle import mrfp_oracle as orc              # noqa: E402


def _sse(x, labels, k):
    return sum(float(((x[labels == q] - x[labels == q].mean()) ** 2).sum()) for q in range(k) if (labels == q).any())


def test_kmeans1d_is_optimal_on_small_inputs():
    rng = np.random.default_rng(0)
    for n, k in [(1, 1), (5, 2), (12, 3), (40, 5), (64, 7), (30, 30), (9, 20)]:
        for trial in range(3):
            x = rng.standard_normal(n) ** 3 if trial else np.round(rng.standard_normal(n), 1)   # trial 0: many ties
            labels, cent = cs.kmeans1d_cluster(x, k)
            rl, rc, rcost = orc.kmeans1d_reference(x, k)
            kk = min(k, n)
            assert labels.shape == (n,) and cent.shape == (kk,)
            assert abs(_sse(x, labels, kk) - rcost) <= 1e-9 * max(1.0, rcost), (n, k, trial)
            assert np.all(np.diff(cent) >= -1e-12)                       # clusters numbered by ascending centroid
            order = np.argsort(x, kind="stable")
            assert np.all(np.diff(labels[order]) >= 0)                   # contiguous in sorted order
            for q in range(kk):
                if (labels == q).any():
                    assert abs(cent[q] - x[labels == q].mean()) < 1e-9


def test_kmeans1d_large_input_runs_and_separates_a_zero_heavy_distribution():
    # the shape of the real call: the flattened C x C variance matrix, more than half of it exact zeros
    rng = np.random.default_rng(1)
    C = 128
    v = np.triu(rng.gamma(0.5, 1.0, (C, C)), 1)
    labels, cent = cs.kmeans1d_cluster(v.reshape(-1), 50)
    assert cent.shape == (50,) and np.all(np.diff(cent) > 0)
    assert (labels[v.reshape(-1) == 0] == 0).all() and cent[0] < 0.05
    assert abs(_sse(v.reshape(-1), labels, 50) - sum(
        float(((v.reshape(-1)[labels == q] - cent[q]) ** 2).sum()) for q in range(50))) < 1e-6


def test_cov_index_matrix():
    for dim in (2, 3, 8, 19):
        np.testing.assert_array_equal(cs.make_cov_index_matrix(dim).numpy(), orc.isw_cov_index_matrix(dim))


def test_irw_and_isw_bookkeeping_against_the_restatement():
    rng = np.random.default_rng(2)
    C = 12
    irw = cs.CovMatrix_IRW(C, relax_denom=4, device="cpu")
    eye, mask, margin, n_off = irw.get_mask_matrix()
    assert float(n_off) == C * (C - 1) / 2 and float(margin) == (C * (C - 1) // 2) // 4
    np.testing.assert_array_equal(mask.numpy(), np.triu(np.ones((C, C)), 1))
    assert float(cs.CovMatrix_IRW(C, device="cpu").get_mask_matrix()[2]) == 0

    isw = cs.CovMatrix_ISW(C, relax_denom=0, clusters=3, device="cpu")
    acc = np.zeros((C, C))
    for _ in range(4):                                   # four "batches" of an image and its photometric transform
        f = rng.standard_normal((2, C, 6, 5)) * rng.uniform(0.2, 3.0, (1, C, 1, 1))
        var = orc.isw_variance_of_covariance(f)
        acc += var
        isw.set_variance_of_covariance(torch.from_numpy(var).float())
    eye, mask, margin, n_sens = isw.get_mask_matrix()
    ref = orc.isw_mask((acc / 4).astype(np.float32).astype(np.float64), 3)
    np.testing.assert_array_equal(mask.numpy(), ref)
    assert margin == 0 and float(n_sens) == ref.sum() and 0 < ref.sum() < C * (C - 1) / 2
    assert np.all(ref[np.tril_indices(C)] == 0)          # only strictly-upper entries can be sensitive
    # a second calibration round intersects with the first mask (cov_settings.py:68-71)
    isw.set_variance_of_covariance(torch.from_numpy(orc.isw_variance_of_covariance(rng.standard_normal((2, C, 6, 5)))).float())
    isw.set_mask_matrix()
    assert np.all(isw.mask_matrix.numpy() <= ref)
    isw.reset_mask_matrix()
    assert isw.mask_matrix is None


def test_build_cov_matrix_layers():
    layers, kinds = cs.build_cov_matrix_layers([0, 0, 2, 2, 1, 0, 0], [0, 0, 64, 256, 512, 1024, 2048], clusters=3, device="cpu")
    assert kinds == [2, 2, 1] and [l.dim for l in layers] == [64, 256, 512]
    assert isinstance(layers[0], cs.CovMatrix_ISW) and isinstance(layers[2], cs.CovMatrix_IRW)
