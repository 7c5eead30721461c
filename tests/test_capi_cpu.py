"""No-GPU checks of the boundary: the C-ABI library builds, loads, and exports every symbol that
include/mrfp_hip.h declares; host-only entry points answer; the product path refuses to run
without a GPU instead of falling back."""
import ctypes
import os

import pytest
import torch

from mrfp_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cdll():
    build.build()
    return _lib.lib()


def test_header_symbols_exported(cdll):
    protos = _lib.parse_header()
    assert len(protos) >= 24
    for name in protos:
        assert hasattr(cdll, name), name
    assert cdll.mrfp_version() >= 100


def test_host_only_entry_points(cdll):
    assert cdll.mrfp_stats_nslab(16, 384) == 128
    assert cdll.mrfp_stats_nslab(2, 64) == 64
    assert 1 <= cdll.mrfp_ce_nblocks(10) <= 2048 and cdll.mrfp_ce_nblocks(1 << 30) == 2048
    # argument validation happens on the host before any launch
    rc = cdll.mrfp_add(None, None, None, 0, 0, None)
    assert rc != 0 and b"add" in cdll.mrfp_last_error()


def test_row_kernels_give_every_workgroup_the_same_number_of_lines(cdll):
    """mrfp_stats_nslab(B, Ho) = workgroups per image of the statistics / apply / pool kernels (workgroup j walks lines j, j + n, ...):
    at most the cap (2048 workgroups per launch), never more than the lines, and no workgroup with more than one line above another
    -- 192 lines over 128 workgroups (two lines for half of them, one for the rest) was 1.5 % of the normalisation family."""
    cdll.mrfp_stats_nslab.restype = ctypes.c_int64
    cdll.mrfp_stats_nslab.argtypes = [ctypes.c_int64, ctypes.c_int64]
    for B in (1, 2, 16, 64):
        cap = max(1, 2048 // B)
        for Ho in (1, 5, 48, 96, 127, 128, 129, 192, 231, 256, 277, 321, 332, 384, 768, 1024, 2048, 5000):
            n = int(cdll.mrfp_stats_nslab(B, Ho))
            assert 1 <= n <= min(Ho, cap), (B, Ho, n)
            per = -(-Ho // n)                              # lines of the busiest workgroup
            assert per == -(-Ho // min(Ho, cap)), (B, Ho, n)      # ... is what the cap forces, not more
            assert (per - 1) * n < Ho, (B, Ho, n)          # and no workgroup is idle or two lines behind
            assert n * per - Ho < per or per == 1, (B, Ho, n)
    assert int(cdll.mrfp_stats_nslab(16, 192)) == 96 and int(cdll.mrfp_stats_nslab(16, 384)) == 128


def test_fourier_and_whitening_host_queries(cdll):
    """Host-only queries of the Fourier / whitening families: the band-limited low-band path stores floor(radius)+1
    spectrum columns on planned line lengths and the whole half spectrum otherwise; workspace sizes; argument checks."""
    sb = cdll.mrfp_fourier_stored_bins
    assert sb(192, 192, 16.0, 0) == 17 and sb(192, 192, 16.9, 0) == 17 and sb(192, 192, 0.0, 0) == 1
    assert sb(192, 192, 16.0, 1) == 97                      # high band: the whole half spectrum
    assert sb(192, 192, 500.0, 0) == 97 and sb(32, 48, 30.0, 0) == 25
    assert sb(8, 8, 2.0, 0) == 5 and sb(12, 18, 3.0, 0) == 10   # no register plan for these lengths: generic path
    assert cdll.mrfp_fourier_spectrum_bytes(2, 8, 8, 16) == 2 * 8 * 5 * 16 * 8
    ws = cdll.mrfp_group_moments_ws_bytes
    assert ws(16, 192 * 192, 256) > 0 and ws(16, 192 * 192, 256) % (64 * 68 * 4) == 0
    assert ws(2, 100, 24) == 0 and ws(2, 100, 2048) == 0    # C % 16 != 0, C > 1024: unsupported
    assert cdll.mrfp_group_moments(None, None, None, None, None, 0, 1, 1, 16, None) != 0
    assert b"group_moments" in cdll.mrfp_last_error()
    assert cdll.mrfp_group_isqrt_fwd(None, None, 1, 5, None) != 0 and b"group_isqrt" in cdll.mrfp_last_error()


def test_wgrad_workspace_covers_the_accumulator_stationary_kernels(cdll):
    """mrfp_conv_wgrad_ws_bytes / _grouped_ws_bytes know (M, N, Q, count) only: the workspace must hold the slab slots of whichever kernel the launch
    picks -- conv_wgrad_kernel's K' splits, or, for the accumulator-stationary kernels (csrc/conv_wg3.hip: 64 x 9 x 64 classes over 512 workgroups;
    csrc/conv_wg1.hip: 256 x 256 classes over 256), `main chunks per class` + 2 remainder slots per problem."""
    one, grp = cdll.mrfp_conv_wgrad_ws_bytes, cdll.mrfp_conv_wgrad_grouped_ws_bytes
    for f in (one, grp):
        f.restype = ctypes.c_int64
    one.argtypes = [ctypes.c_int64] * 3
    grp.argtypes = [ctypes.c_int64] * 4

    # the grids of the persistent kernels come from ONE constant (csrc/common.hpp: kCUs; conv_common.hpp: kGrid1PerCU / kGrid2PerCU): read
    # it from the source instead of repeating the numbers here
    import re
    src = open(os.path.join(ROOT, "mrfp_amd", "csrc", "common.hpp")).read()
    cus = int(re.search(r"constexpr int kCUs = (\d+);", src).group(1))
    cc = open(os.path.join(ROOT, "mrfp_amd", "csrc", "conv_common.hpp")).read()
    assert "kGrid1PerCU = kCUs;" in cc and "kGrid2PerCU = 2 * kCUs;" in cc
    for f_, lit in (("conv_wg3.hip", r"\b512\s*/\s*count"), ("conv_wg1.hip", r"\b256\s*/\s*count"), ("conv_c64.hip", r"\?\s*512\s*:\s*256")):
        assert not re.search(lit, open(os.path.join(ROOT, "mrfp_amd", "csrc", f_)).read()), "a grid literal came back in " + f_

    def slots3(N, Q, count):
        ncls = (N // 64) * (Q // 576)
        a = (2 * cus // count) // ncls
        return a + 2 if a >= 1 else 0

    def slots1(N, Q, count):
        ncls = (N // 256) * (Q // 256)
        a = (cus // count) // ncls
        return a + 2 if a >= 1 else 0
    for M, N, Q, count in [(16 * 384 * 384, 64, 576, 1), (16 * 384 * 384, 128, 576, 1), (16 * 192 * 192, 256, 2304, 1), (16 * 192 * 192, 256, 2880, 1),
                           (36864, 256, 2304, 22), (36864, 512, 4608, 3), (16 * 192 * 192, 64, 576, 3), (16 * 96 * 96, 128, 1152, 3)]:
        got = int(grp(M, N, Q, count)) if count > 1 else int(one(M, N, Q))
        assert got % (N * Q * 4) == 0 and got >= slots3(N, Q, count) * count * N * Q * 4, (M, N, Q, count, got)
    for M, N, Q, count in [(36864, 1024, 256, 23), (36864, 256, 1024, 22), (36864, 2048, 512, 3), (36864, 512, 2048, 2), (36864, 2048, 1024, 1)]:
        got = int(grp(M, N, Q, count)) if count > 1 else int(one(M, N, Q))
        assert got % (N * Q * 4) == 0 and got >= slots1(N, Q, count) * count * N * Q * 4, (M, N, Q, count, got)
    assert int(one(36864, 19, 256)) > 0 and int(grp(36864, 19, 256, 32)) > 0         # shapes neither kernel takes still get their splits


def test_nearest_tables_match_aten():
    """The host-side index tables the kernels consume == F.interpolate(mode='nearest')."""
    import numpy as np
    import torch.nn.functional as F
    from mrfp_amd import ops
    for size, kw in [(192, dict(scale=1.205)), (231, dict(scale=1.2)), (277, dict(scale=1.2)), (332, dict(size=384)),
                     (384, dict(scale=0.838)), (321, dict(scale=0.798)), (256, dict(size=192)), (64, dict(scale=1.205)),
                     (616, dict(scale=1.2)), (1024, dict(scale=0.838))]:
        probe = torch.arange(size, dtype=torch.float32).view(1, 1, 1, size)
        if "scale" in kw:
            ref = F.interpolate(probe, scale_factor=(1.0, kw["scale"])).flatten().long().numpy()
            out = ops.nearest_out_size(size, kw["scale"])
            tab = ops._nearest_table(size, out, kw["scale"])
        else:
            ref = F.interpolate(probe, size=(1, kw["size"])).flatten().long().numpy()
            out = kw["size"]
            tab = ops._nearest_table(size, out, None)
        assert out == len(ref)
        np.testing.assert_array_equal(tab, ref)
        inv = ops._inverse_table(tab, size).reshape(-1, 2)
        for s in range(size):
            assert (tab[inv[s, 0]:inv[s, 1]] == s).all() and (inv[s, 1] - inv[s, 0]) == (tab == s).sum()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_refuses_cpu():
    from mrfp_amd.deepv3 import MRFPPlus
    m = MRFPPlus(19, criterion=torch.nn.CrossEntropyLoss(ignore_index=255))
    with pytest.raises(_lib.MrfpHipError):
        m(torch.zeros(2, 3, 64, 64), torch.zeros(2, 64, 64, dtype=torch.long))


def test_state_dict_abi():
    """Same 431 keys / shapes / trainable set as the reference module (checkpoint ABI)."""
    import json
    from mrfp_amd.deepv3 import MRFPPlus, simpleDeepV3Plus
    spec = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_spec.json")))
    m = MRFPPlus(19)
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == spec["MRFPPlus"]
    assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == sorted(spec["trainable"])
    assert sorted(n for n, p in m.named_parameters() if not p.requires_grad) == sorted(spec["frozen"])
    p = simpleDeepV3Plus(19)
    assert [[k, list(v.shape)] for k, v in p.state_dict().items()] == spec["simpleDeepV3Plus"]
    with pytest.raises(ValueError):
        MRFPPlus(19, trunk="resnet-18")
    with pytest.raises(ValueError):
        simpleDeepV3Plus(19, trunk="resnet-101")
