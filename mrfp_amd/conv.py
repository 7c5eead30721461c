"""nn.Conv2d forward / backward on the MFMA implicit-GEMM kernels (mrfp_amd/csrc/conv.hip).

The OIHW fp32 Parameter stays the master copy (checkpoint ABI, optimiser state); the kernels read
two derived packs in the activation dtype -- forward [N][R][S][C] and dgrad [C][R][S][N] with flipped
taps -- rebuilt by one small kernel whenever the Parameter's version counter moves (optimizer
step, HRFP re-initialisation, load_state_dict).
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import _lib
from ._lib import call, dt, ptr, stream
from .ops import CL, GRAD_DEFERRED, _chk, empty_cl, grad_sink, notify_grad, zeros_cl

_PACKS = {}      # id(weight Parameter) -> {key: _Pack}; entry dropped when the Parameter dies
import os as _os
FUSE_STATS = [_os.environ.get("MRFP_FUSE_STATS", "1") != "0"]   # conv epilogues emit BatchNorm partial statistics for bias-free convolutions
_LAST_STATS = [None]  # handed from _Conv2d.forward to conv2d() (autograd re-wraps the output tensor object)
# set by a caller right before a convolution whose output goes through a nearest-neighbour resize into a training-mode BatchNorm
# (deepv3.MRFPPlus._hrfp): the ops.NearestPlan of that resize.  The epilogue statistics then count every output pixel as often as
# the resize reads it (mrfp_conv_fwd_wstats), and the BatchNorm skips its statistics pass over the resized tensor.
STAT_RESIZE = [None]
WSTATS = [_os.environ.get("MRFP_WSTATS", "1") != "0"]
WSTATS_HITS = [0]
_EPOCH = [0]     # bumped by writers that bypass autograd's version counters (the fused SGD kernel)


def invalidate_packs():
    _EPOCH[0] += 1


def _epc(dtype) -> int:
    """elements per 16-byte chunk"""
    return 4 if dtype == torch.float32 else 8


def _round_up(n, m):
    return (n + m - 1) // m * m


class _Pack:
    __slots__ = ("version", "wf", "wd", "bias", "key", "wref")


def get_pack(weight: torch.Tensor, bias: Optional[torch.Tensor], dtype, Cphys: int, Nphys: int) -> _Pack:
    N, C, R, S = weight.shape
    key = (dtype, Cphys, Nphys, weight.data_ptr(), bias.data_ptr() if bias is not None else 0)
    ver = (weight._version, bias._version if bias is not None else 0, _EPOCH[0])
    per_w = _PACKS.get(id(weight))
    if per_w is None:
        per_w = {}
        _PACKS[id(weight)] = per_w
        weakref.finalize(weight, _PACKS.pop, id(weight), None)
    pk = per_w.get(key)
    if pk is not None and pk.version == ver:
        return pk
    dev = weight.device
    w32 = weight.detach()
    if w32.dtype != torch.float32 or not w32.is_contiguous():
        w32 = w32.float().contiguous()
    if pk is None:
        pk = _Pack()
        pk.wf = torch.empty(Nphys * R * S * Cphys, dtype=dtype, device=dev)
        # dgrad pack [Cphys][R][S][Nphys]: the rows of pad input channels (c >= C) stay zero, so a dgrad launch over a
        # channel-padded input writes exact zeros into the pad channels of dx
        pk.wd = (torch.zeros(Cphys * R * S * Nphys, dtype=dtype, device=dev) if Cphys != C
                 else torch.empty(C * R * S * Nphys, dtype=dtype, device=dev))
        pk.bias = None
        pk.version = None
        pk.wref = weakref.ref(weight)
    call("mrfp_pack_weight", ptr(w32), ptr(pk.wf), ptr(pk.wd), _lib._DT[dtype], N, C, R, S, Nphys, Cphys, stream())
    if bias is not None:
        if pk.bias is None:                      # allocated once (pad entries stay zero), refreshed in place
            pk.bias = torch.zeros(Nphys, dtype=torch.float32, device=dev)
        pk.bias[:N].copy_(bias.detach())
    # under hipGraph capture (harness.Trainer.enable_graph) the pack kernel is only recorded, not run: the cached pack
    # must stay "stale" for eager code, and the recorded kernel re-packs on every replay
    if not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        pk.version = ver
    per_w[key] = pk
    return pk


_BATCH = {}


def _packable(pk, w, with_bias=False):
    return (w is not None and (with_bias or pk.bias is None) and w.dtype == torch.float32 and w.is_contiguous()
            and w.shape[2] * w.shape[3] <= 9)        # the brick kernel holds up to 3x3 taps; the 7x7 stem packs lazily


def _batched_repack(todo, tag, biases=None, launch=True):
    """ONE mrfp_pack_weights_batched launch per dtype for the (key, pack, weight) triples in `todo`; the job table is cached
    per (tag, dtype) and rebuilt only when the set of packs changes."""
    import numpy as np
    by_dtype = {}
    for key, pk, w in todo:
        by_dtype.setdefault(key[0], []).append((key, pk, w))
    for dtype, items in by_dtype.items():
        sig = tuple((id(pk), w.data_ptr(), pk.wf.data_ptr(), pk.wd.data_ptr()) for _, pk, w in items)
        st = _BATCH.get((tag, dtype))
        if st is None or st["sig"] != sig:
            rec = np.zeros(len(items), dtype=np.dtype([("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("dims", "<i4", (6,))]))
            prefix = np.zeros(len(items) + 1, dtype=np.int64)
            for i, (key, pk, w) in enumerate(items):
                N, C, R, S = w.shape
                _, Cphys, Nphys = key[0], key[1], key[2]
                rec[i]["w"], rec[i]["wf"], rec[i]["wd"] = w.data_ptr(), pk.wf.data_ptr(), pk.wd.data_ptr()
                rec[i]["dims"] = (N, C, R, S, Nphys, Cphys)
                bc = 64 if R * S == 1 else 8                                           # input channels per brick (conv.hip)
                prefix[i + 1] = prefix[i] + ((Nphys + 63) // 64) * ((Cphys + bc - 1) // bc)
            dev = items[0][2].device
            # (pinned staging + asynchronous copies: legal while a hipGraph is being captured too -- the first capture pass of
            #  Trainer.enable_graph() is the first time the HRFP packs exist when the re-initialisation runs; the host copies stay
            #  alive in the table entry for the graph's memcpy nodes)
            hj, hp = torch.from_numpy(rec.view(np.uint8).copy()), torch.from_numpy(prefix)
            if dev.type == "cuda":
                hj, hp = hj.pin_memory(), hp.pin_memory()
            st = {"sig": sig, "host": (hj, hp), "jobs": hj.to(dev, non_blocking=True),
                  "prefix": hp.to(dev, non_blocking=True), "total": int(prefix[-1]), "n": len(items)}
            _BATCH[(tag, dtype)] = st
        if not launch:
            continue
        call("mrfp_pack_weights_batched", ptr(st["jobs"]), ptr(st["prefix"]), st["n"], st["total"], _lib._DT[dtype], stream())
        capturing = torch.cuda.is_current_stream_capturing()
        for key, pk, w in items:
            b = biases.get(id(w)) if biases else None
            if b is not None and pk.bias is not None:           # packs with a bias (the HRFP convolutions): refresh the fp32 copy
                pk.bias[:w.shape[0]].copy_(b.detach())
            if not capturing:
                pk.version = (w._version, b._version if b is not None else 0, _EPOCH[0])


def repack_all():
    """Re-packs, in ONE launch, every cached bias-free pack whose fp32 master changed (called by the harness right after
    the fused SGD kernel rewrote the parameter arena; 124 launches of ~7 us per step otherwise).  Packs with a bias
    (final2) keep the lazy per-layer path; the frozen HRFP weights, re-drawn at the start of a forward, go through
    repack_weights()."""
    todo = []
    for per_w in _PACKS.values():
        for key, pk in per_w.items():
            w = pk.wref() if getattr(pk, "wref", None) is not None else None
            if w is None or not w.requires_grad or not _packable(pk, w):
                continue
            todo.append((key, pk, w))
    if todo:
        _batched_repack(todo, "trainable")


def repack_weights(weights, tag="list", biases=None):
    """The same for an explicit list of weights that were just rewritten (the HRFP branch's convolutions after their
    re-initialisation: reference deepv3.py:290-299 re-draws them at the start of a forward; 28 pack launches per step
    otherwise).  `biases`: the bias Parameter of each weight (or None) -- a pack with a bias gets its fp32 copy refreshed here too.
    Weights without a cached pack yet are left to the lazy path."""
    _REPACK_LISTS[tag] = ([weakref.ref(w) for w in weights], [weakref.ref(b) if b is not None else None for b in biases]
                          if biases is not None else None)
    todo, bmap = _repack_selection(weights, biases)
    # the masters were just rewritten -- possibly through a flat arena they are views of (deepv3._hrfp_arena), which does not bump
    # the parameters' own version counters: every cached pack of these weights that the batched launch does NOT refresh (another
    # dtype / padding, a 7x7 filter, a bias mismatch) is marked stale here and rebuilt lazily at its next use
    sel = {id(pk) for _, pk, _ in todo}
    for w in weights:
        for pk in _PACKS.get(id(w), {}).values():
            if id(pk) not in sel:
                pk.version = None
    if todo:
        _batched_repack(todo, tag, bmap)


_REPACK_LISTS = {}       # tag -> the (weights, biases) of the last repack_weights() call


def prebuild_repack_tables():
    """Builds (without launching) the job tables of every repack_weights() list seen so far, for the packs that exist NOW: called
    by harness.Trainer between the eager warm-up pass of a hipGraph capture and the capture itself -- in the very first step the
    packs of the HRFP convolutions come into being AFTER the re-initialisation ran, so the capture pass would be the first to
    need the table, and a host -> device copy cannot be captured."""
    for tag, (wrefs, brefs) in list(_REPACK_LISTS.items()):
        weights = [r() for r in wrefs]
        if any(w is None for w in weights):          # the model is gone
            del _REPACK_LISTS[tag]
            continue
        biases = [r() if r is not None else None for r in brefs] if brefs is not None else None
        todo, bmap = _repack_selection(weights, biases)
        if todo:
            _batched_repack(todo, tag, bmap, launch=False)


def _repack_selection(weights, biases):
    todo = []
    bmap = {id(w): b for w, b in zip(weights, biases)} if biases is not None else {}
    for w in weights:
        for key, pk in _PACKS.get(id(w), {}).items():
            has_b = bmap.get(id(w)) is not None
            if pk.wf is not None and _packable(pk, w, with_bias=has_b) and (pk.bias is None) == (not has_b) \
                    and key[4] == (bmap[id(w)].data_ptr() if has_b else 0):
                todo.append((key, pk, w))
    return todo, bmap


def _out_size(H, R, stride, pad, dil):
    return (H + 2 * pad - dil * (R - 1) - 1) // stride + 1


# ---- second stream for the weight gradients ---------------------------------------------------------------------
USE_WGRAD_STREAM = [_os.environ.get("MRFP_WGRAD_STREAM", "1") != "0"]
_WGRAD_STREAMS = {}
_WGRAD_PENDING = [False]


def wgrad_stream(device):
    """The side stream weight gradients are computed on (None when disabled or while a hipGraph is being captured)."""
    if not USE_WGRAD_STREAM[0] or device.type != "cuda" or torch.cuda.is_current_stream_capturing():
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _WGRAD_STREAMS.get(key)
    if st is None:
        st = _WGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


_JOIN_QUEUED = [False]


def _join_at_end_of_backward():
    _JOIN_QUEUED[0] = False
    flush_wgrads()
    # what this pass produced is what the next pass OF THE SAME KIND is expected to produce.  Passes are told apart by the
    # geometry of their first weight gradient (the head's convolution: input shape, batch, dtype and class count are in it), so
    # two models / two input shapes alternating in one process each keep their own expectation instead of
    # flushing late or early on the other one's counts.  (Grouping is a speed matter only: every flush is a correct launch.)
    if _WG_PASS_KEY[0] is not None:
        if len(_WG_EXPECT_ALL) >= 64 and _WG_PASS_KEY[0] not in _WG_EXPECT_ALL:
            _WG_EXPECT_ALL.pop(next(iter(_WG_EXPECT_ALL)))
        _WG_EXPECT_ALL[_WG_PASS_KEY[0]] = dict(_WG_SEEN)
    _WG_PASS_KEY[0] = None
    _WG_EXPECT.clear()
    _WG_SEEN.clear()
    join_wgrad_stream()


def join_wgrad_stream(stream=None):
    """Makes `stream` (default: the current one) wait for every weight gradient issued so far."""
    if not _WGRAD_PENDING[0]:
        return
    tgt = stream if stream is not None else torch.cuda.current_stream()
    for st in _WGRAD_STREAMS.values():
        tgt.wait_stream(st)
    if stream is None:
        _WGRAD_PENDING[0] = False


# ---- grouped, deferred weight gradients ----------------------------------------------------------------------------
# A ResNet stage is `blocks` Bottlenecks of ONE geometry (reference network/Resnet.py:579-585 _make_layer; autograd of
# Resnet.py:202-216 yields their weight gradients one launch at a time).  At M = 16*48*48 a single launch has 16-72 output tiles
# and needs 21-32 K' splits to fill 256 CUs -- 12 K' tiles per workgroup behind a prologue, and fp32 slabs of 20-30x the size of
# dW written and read again.  Weight gradients feed nothing but the optimizer, so they can wait: with a gradient arena present
# (harness.FlatArena) backward only QUEUES (x, dy, sink) per launch geometry and mrfp_conv_wgrad_grouped runs a whole queue as one
# launch + one slab reduction -- when a queue is full, when backward leaves a stage (wgrad_boundary), and when backward ends.
GROUP_WGRAD = [_os.environ.get("MRFP_WGRAD_GROUP", "1") != "0"]
_WG_QUEUE = {}              # launch geometry -> [(x, dy, sink, weight)]
_WG_PENDING_BYTES = [0]
# Activations (x, dy) the queues may keep alive beyond the point where an immediate launch would have released them.
# MRFP_WGRAD_GROUP_GB fixes it; by default it is a quarter of the device memory that is free when the first gradient is queued,
# at most 24 GB (the bench step on a 288 GB part: 24; a large-activation configuration -- configs[4] -- on a fuller device gets
# less and flushes earlier instead of raising the peak).
_WG_MAX_BYTES = [int(float(_os.environ["MRFP_WGRAD_GROUP_GB"]) * (1 << 30)) if "MRFP_WGRAD_GROUP_GB" in _os.environ else None]
import collections as _collections
WGRAD_GROUP_LAUNCHES = _collections.deque(maxlen=4096)   # sizes of the last grouped launches (tests / diagnostics; bounded)
_GROUP_MAX = [None]
# How many problems of a geometry one backward pass produces is learnt from the previous pass: a geometry that came ONCE is
# launched at once from then on (nothing to group with -- and the last layers of backward, the stem, would otherwise leave as an
# exposed tail behind the end of the dgrad chain), a repeated one leaves as soon as its expected count is complete.
_WG_EXPECT = {}             # launch geometry -> problems seen in the last complete backward pass of THIS kind (see _WG_PASS_KEY)
_WG_SEEN = {}               # ... in the running one
_WG_PASS_KEY = [None]       # geometry of the first weight gradient of the running backward pass
_WG_EXPECT_ALL = {}         # pass key -> {launch geometry -> problems} of the last complete pass that started with it


_DEBUG_SKIP_WGRAD = _os.environ.get("MRFP_DEBUG_SKIP_WGRAD") == "1"     # timing diagnostics only: weight gradients are NOT computed


def _issue_wgrads(sig, items):
    """One mrfp_conv_wgrad(_grouped) launch for `items` (same geometry) on the weight-gradient stream (or the current one)."""
    import ctypes
    if _DEBUG_SKIP_WGRAD:
        for _, _, _, weight in items:
            GRAD_DEFERRED.discard(id(weight))
            notify_grad(weight)
        return
    (dtype, B, H, W, Cphys, C, N, Nphys, R, S, Ho, Wo, stride, pad_h, pad_w, dil) = sig
    x0 = items[0][0]
    M, Q = B * Ho * Wo, R * S * Cphys
    side = wgrad_stream(x0.device)
    L = _lib.lib()

    def run():
        n = len(items)
        if n == 1:
            x, dy, sink, _ = items[0]
            ws = torch.empty(int(L.mrfp_conv_wgrad_ws_bytes(M, N, Q)), dtype=torch.uint8, device=x.device)
            call("mrfp_conv_wgrad", ptr(x), ptr(dy), ptr(sink), ptr(ws), _lib._DT[dtype], B, H, W, Cphys, C, N, Nphys, R, S,
                 Ho, Wo, stride, pad_h, pad_w, dil, stream())
            return
        ws = torch.empty(int(L.mrfp_conv_wgrad_grouped_ws_bytes(M, N, Q, n)), dtype=torch.uint8, device=x0.device)
        arr = ctypes.c_void_p * n
        xs, dys, dws = arr(*[ptr(i[0]) for i in items]), arr(*[ptr(i[1]) for i in items]), arr(*[ptr(i[2]) for i in items])
        call("mrfp_conv_wgrad_grouped", xs, dys, dws, n, ptr(ws), _lib._DT[dtype], B, H, W, Cphys, C, N, Nphys, R, S,
             Ho, Wo, stride, pad_h, pad_w, dil, stream())

    if side is not None:
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            run()
        for x, dy, _, _ in items:
            dy.record_stream(side)
            x.record_stream(side)
        _WGRAD_PENDING[0] = True
    else:
        run()
    WGRAD_GROUP_LAUNCHES.append(len(items))
    for _, _, _, weight in items:
        GRAD_DEFERRED.discard(id(weight))
        notify_grad(weight)


def flush_wgrads(sig=None):
    """Issues the queued weight gradients (of one launch geometry, or all of them in first-queued order)."""
    keys = [sig] if sig is not None else list(_WG_QUEUE.keys())
    for k in keys:
        items = _WG_QUEUE.pop(k, None)
        if not items:
            continue
        _WG_PENDING_BYTES[0] -= sum(i[0].numel() * i[0].element_size() + i[1].numel() * i[1].element_size() for i in items)
        _issue_wgrads(k, items)


def _queue_wgrad(sig, x, dy, sink, weight):
    if _GROUP_MAX[0] is None:
        _GROUP_MAX[0] = min(int(_lib.lib().mrfp_conv_wgrad_group_max()), int(_os.environ.get("MRFP_WGRAD_GROUP_MAX", "32")))
    if not _JOIN_QUEUED[0]:       # when this backward pass ends: flush every queue, then the caller's stream waits for the side stream
        _JOIN_QUEUED[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_join_at_end_of_backward)
    if _WG_PASS_KEY[0] is None:   # first weight gradient of this pass: pick up the expectation of the passes that started like it
        _WG_PASS_KEY[0] = sig
        _WG_EXPECT.clear()
        _WG_EXPECT.update(_WG_EXPECT_ALL.get(sig, {}))
        _WG_SEEN.clear()
    if _WG_MAX_BYTES[0] is None:
        free, _total = torch.cuda.mem_get_info(x.device)
        _WG_MAX_BYTES[0] = min(24 << 30, max(1 << 30, free // 4))
    seen = _WG_SEEN[sig] = _WG_SEEN.get(sig, 0) + 1
    expect = _WG_EXPECT.get(sig, 0)
    if expect == 1 and seen == 1 and sig not in _WG_QUEUE:
        _issue_wgrads(sig, [(x, dy, sink, weight)])       # a geometry of its own: nothing to wait for
        return
    q = _WG_QUEUE.setdefault(sig, [])
    q.append((x, dy, sink, weight))
    GRAD_DEFERRED.add(id(weight))
    _WG_PENDING_BYTES[0] += x.numel() * x.element_size() + dy.numel() * dy.element_size()
    if len(q) >= _GROUP_MAX[0] or (expect > 1 and seen % expect == 0):
        flush_wgrads(sig)
    elif _WG_PENDING_BYTES[0] > _WG_MAX_BYTES[0]:
        flush_wgrads()


def _in_backward():
    """True while the autograd engine is executing a graph task on this thread (a forward convolution issued from inside a
    backward pass -- activation checkpointing, recomputation in a custom backward -- is legitimate and must not be taken for
    the sign of a dead pass)."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f is not None and f() != -1


def _drop_stale_backward_state(reason="a forward convolution found the end-of-backward callback of an earlier pass still pending"):
    """The last backward pass died with an exception before the engine ran its callbacks (the harness calls this from the
    `finally` of its forward/backward, conv2d() when it is called OUTSIDE any backward pass with the callback still marked as
    queued).  Its queued weight gradients are forgotten -- their step is lost anyway -- so that the next backward registers its
    own callback; dropping a non-empty queue is reported, never silent."""
    dropped = sum(len(v) for v in _WG_QUEUE.values())
    _JOIN_QUEUED[0] = False
    _WG_QUEUE.clear()
    _WG_SEEN.clear()
    _WG_EXPECT.clear()
    _WG_PASS_KEY[0] = None
    _WG_PENDING_BYTES[0] = 0
    GRAD_DEFERRED.clear()
    if dropped:
        import warnings
        warnings.warn("mrfp_amd.conv: %d queued weight gradients of a backward pass that did not finish were dropped (%s)"
                      % (dropped, reason), RuntimeWarning, stacklevel=3)


def backward_failed():
    """To be called when a backward pass raised (harness.Trainer does, from its `finally`): clears the deferred weight-gradient
    state at once instead of leaving it for the next forward convolution to find."""
    if _JOIN_QUEUED[0] or _WG_QUEUE:
        _drop_stale_backward_state("the backward pass raised")


def wgrad_boundary(t):
    """Marks a stage boundary on activation `t` (the input of a ResNet stage / of the head): when backward has produced t's
    gradient, everything behind it has queued its weight gradients, and the queues are issued -- the stage's gradient buckets
    complete there (harness.GradSync launches buckets in index order, so later arrival is tolerated) instead of at the very end
    of backward.  Returns t."""
    if GROUP_WGRAD[0] and t.requires_grad and torch.is_grad_enabled():
        t.register_hook(_boundary_hook)
    return t


def _boundary_hook(_grad):
    flush_wgrads()
    return None


GATED_SKIP_HITS = [0]       # dgrad launches that applied a residual tail's gate to their addend (tests)


def ungate(t):
    """The plain gradient for a tensor that carries a `_mrfp_gate` (sign mask, version) tag: t * [mask bit set].  Tensors
    without the tag (and None) pass through."""
    g = getattr(t, "_mrfp_gate", None) if t is not None else None
    if g is None:
        return t
    if g[1] != t._version:
        raise _lib.MrfpHipError("a gated skip gradient was modified in place before it reached its consumer")
    B, C, H, W = t.shape
    out = empty_cl(B, C, H, W, t.dtype, t.device)
    call("mrfp_affine_bwd_mask", ptr(t), None, ptr(g[0]), ptr(out), None, dt(t), B, H, W, C, None, None, None, 0, stream())
    return out


def L_single(B, H, W, Cphys, Ho, Wo, Nphys, esz):
    """both operands of a weight-gradient problem fit one 3.75 GB buffer-descriptor range (a grouped launch needs that)"""
    return max(B * H * W * Cphys, B * Ho * Wo * Nphys) * esz < 0xF0000000


class _Conv2d(torch.autograd.Function):
    """want_skip: also return an alias of x for a skip connection; the gradient arriving on that alias is added by
    the dgrad kernel's epilogue (no separate accumulation pass over the activation)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad_h, pad_w, dil, Nphys, want_skip):
        B, Cphys, H, W = x.shape
        N, C, R, S = weight.shape
        Ho, Wo = _out_size(H, R, stride, pad_h, dil), _out_size(W, S, stride, pad_w, dil)
        pk = get_pack(weight, bias, x.dtype, Cphys, Nphys)
        y = empty_cl(B, Nphys, Ho, Wo, x.dtype, x.device)
        stats = None
        L = _lib.lib()
        plan, STAT_RESIZE[0] = STAT_RESIZE[0], None
        single = bool(L.mrfp_conv_single_launch(B, H * W * Cphys * x.element_size()))
        wtab = None
        if plan is not None and WSTATS[0] and FUSE_STATS[0] and single and (plan.Hs, plan.Ws) == (Ho, Wo) and R * S > 1:
            wtab = plan.multiplicity(B)          # statistics of the nearest-resized output (see STAT_RESIZE)
        if wtab is not None or (bias is None and FUSE_STATS[0] and single):
            # no bias = a convolution that feeds a normalisation layer: let the epilogue produce its statistics
            nblk = int(L.mrfp_conv_stats_blocks(dt(x), B, H, W, Cphys, Nphys, R, S, Ho, Wo, stride, pad_h, pad_w, dil, 1))
            stats = torch.empty(int(L.mrfp_conv_stats_rows(nblk)) * 2 * Nphys, dtype=torch.float32, device=x.device)
        _lib.NOTE[0] = (C, N)
        if wtab is not None:
            call("mrfp_conv_fwd_wstats", ptr(x), ptr(pk.wf), ptr(pk.bias), ptr(y), dt(x), B, H, W, Cphys, Nphys, Nphys, R, S,
                 Ho, Wo, stride, pad_h, pad_w, dil, ptr(wtab), ptr(stats), stream())
            WSTATS_HITS[0] += 1
        else:
            call("mrfp_conv_fwd", ptr(x), ptr(pk.wf), ptr(pk.bias), ptr(y), dt(x), B, H, W, Cphys, Nphys, Nphys, R, S,
                 Ho, Wo, stride, pad_h, pad_w, dil, 1, None, ptr(stats), stream())
        if stats is not None:      # the rows the BatchNorm finalize should read (compacted for large launches)
            first, cnt = int(L.mrfp_conv_stats_final_first(nblk)), int(L.mrfp_conv_stats_final_count(nblk))
            final = stats[first * 2 * Nphys:(first + cnt) * 2 * Nphys]
            if wtab is not None:
                # (element count = that of the RESIZED tensor; no per-image use; the plan identifies the one consumer they serve)
                _LAST_STATS[0] = (final, cnt, B * plan.Ho * plan.Wo, stats, nblk, 0, plan)
            else:
                # (rows to hand to a BatchNorm finalize, their count, the element count; + the RAW per-row-block rows and their count:
                #  an InstanceNorm consumer needs them per image -- ops._InstanceNormAct)
                rb = int(L.mrfp_conv_stats_block_rows(dt(x), B, H, W, Cphys, Nphys, R, S, Ho, Wo, stride, pad_h, pad_w, dil, 1))
                _LAST_STATS[0] = (final, cnt, B * Ho * Wo, stats, nblk, rb)
        else:
            _LAST_STATS[0] = None
        ctx.save_for_backward(x, weight, bias)
        ctx.cfg = (stride, pad_h, pad_w, dil, Nphys, Ho, Wo)
        ctx.set_materialize_grads(False)
        if want_skip:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, weight, bias = ctx.saved_tensors
        cell = getattr(ctx, "_mrfp_cell", None)
        if cell is not None and cell[1]:
            # a residual tail handed its UNMASKED incoming gradient to this alias (ops._BatchNormAct.backward): only the tagged
            # tensor itself may arrive here.  Anything else means a consumer the use count did not see (a raw torch op on the
            # alias) made autograd sum gradients into an untagged tensor -- fail loudly, the sum would be silently wrong
            g = getattr(dskip, "_mrfp_gate", None) if dskip is not None else None
            if g is None or g[1] != dskip._version:
                raise _lib.MrfpHipError("a gated skip gradient reached its convolution without its gate: the skip alias was "
                                        "also consumed by an operator outside mrfp_amd.ops (set MRFP_GATED_SKIP=0)")
            cell[1] = False
        if dy is None:            # only the skip alias was used downstream
            return ungate(dskip), None, None, None, None, None, None, None, None
        stride, pad_h, pad_w, dil, Nphys, Ho, Wo = ctx.cfg
        dy = _chk(dy, "dy")
        B, Cphys, H, W = x.shape
        N, C, R, S = weight.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            pk = get_pack(weight, bias, x.dtype, Cphys, Nphys)
            dx = empty_cl(B, Cphys, H, W, x.dtype, x.device)
            gate = None
            if dskip is not None:
                g = getattr(dskip, "_mrfp_gate", None)
                if g is not None:
                    # the skip-connection gradient of a residual tail arrives UNMASKED with that tail's sign mask attached
                    # (ops._BatchNormAct.backward): the dgrad epilogue applies the gate while it adds -- dy * [y > 0] is never
                    # written or re-read.  Anything this launch cannot do that way gets the plain gradient.
                    if (g[1] == dskip._version and dskip.dtype == x.dtype and x.element_size() == 2
                            and Cphys % 8 == 0 and dskip.shape == dx.shape and dskip.is_contiguous(memory_format=CL)):
                        gate = g[0]
                    else:
                        dskip = ungate(dskip)
                dskip = _chk(dskip, "dskip")
                if dskip.dtype != x.dtype:
                    dskip = dskip.to(x.dtype)
            _lib.NOTE[0] = (N, C)
            if gate is not None:
                call("mrfp_conv_fwd_gated", ptr(dy), ptr(pk.wd), None, ptr(dx), dt(dy), B, Ho, Wo, Nphys, Cphys, Cphys, R, S, H, W,
                     1, dil * (R - 1) - pad_h, dil * (S - 1) - pad_w, dil, stride, ptr(dskip), ptr(gate), stream())
                GATED_SKIP_HITS[0] += 1
            else:
                call("mrfp_conv_fwd", ptr(dy), ptr(pk.wd), None, ptr(dx), dt(dy), B, Ho, Wo, Nphys, Cphys, Cphys, R, S, H, W,
                     1, dil * (R - 1) - pad_h, dil * (S - 1) - pad_w, dil, stride, ptr(dskip), None, stream())
        if ctx.needs_input_grad[1]:
            M, Q = B * Ho * Wo, R * S * Cphys
            sink = grad_sink(weight)      # the parameter's slot in the flat gradient arena, when the harness owns it
            side = wgrad_stream(x.device) if sink is not None else None
            if sink is not None and GROUP_WGRAD[0] and L_single(B, H, W, Cphys, Ho, Wo, Nphys, x.element_size()):
                _queue_wgrad((x.dtype, B, H, W, Cphys, C, N, Nphys, R, S, Ho, Wo, stride, pad_h, pad_w, dil), x, dy, sink, weight)
                dw = None
            elif side is not None:
                # The weight gradient feeds nothing but the optimizer: it runs on a second HIP stream, concurrently with
                # the dgrad chain of the main stream (its workgroups fill the tails / small-kernel gaps of that chain).
                # Ordering: side waits for dy (an event on the main stream); the consumers of the arena (optimizer step,
                # gradient all-reduce) wait for the side stream (harness.Trainer / GradSync, join_wgrad_stream()).
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ws = torch.empty(int(_lib.lib().mrfp_conv_wgrad_ws_bytes(M, N, Q)), dtype=torch.uint8, device=x.device)
                    call("mrfp_conv_wgrad", ptr(x), ptr(dy), ptr(sink), ptr(ws), dt(x), B, H, W, Cphys, C, N, Nphys, R, S,
                         Ho, Wo, stride, pad_h, pad_w, dil, stream())
                dy.record_stream(side)
                x.record_stream(side)
                _WGRAD_PENDING[0] = True
                if not _JOIN_QUEUED[0]:       # when this backward pass ends, the caller's stream waits for the side stream
                    _JOIN_QUEUED[0] = True
                    torch.autograd.Variable._execution_engine.queue_callback(_join_at_end_of_backward)
                notify_grad(weight)
                dw = None
            else:
                ws = torch.empty(int(_lib.lib().mrfp_conv_wgrad_ws_bytes(M, N, Q)), dtype=torch.uint8, device=x.device)
                dw = sink if sink is not None else torch.empty((N, C, R, S), dtype=torch.float32, device=x.device)
                call("mrfp_conv_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), dt(x), B, H, W, Cphys, C, N, Nphys, R, S, Ho, Wo,
                     stride, pad_h, pad_w, dil, stream())
                if sink is not None:
                    notify_grad(weight)
                    dw = None
                elif dw.dtype != weight.dtype:
                    dw = dw.to(weight.dtype)
        if bias is not None and ctx.needs_input_grad[2]:
            from .ops import _stats_fwd
            nslab, sws = _stats_fwd(dy, None)
            out = torch.empty(4 * Nphys, dtype=torch.float32, device=x.device)
            call("mrfp_bn_finalize", ptr(sws), B, nslab, B * Ho * Wo, Nphys, None, None, 0.0, 0.0, None, None,
                 ptr(out[:Nphys]), ptr(out[Nphys:2 * Nphys]), ptr(out[2 * Nphys:3 * Nphys]), ptr(out[3 * Nphys:]), stream())
            db = (out[:N] * float(B * Ho * Wo)).to(bias.dtype)       # column mean * count = column sum
        return dx, dw, db, None, None, None, None, None, None


class _SharedConv1x1Pair(torch.autograd.Function):
    """(conv1x1(x1, W), conv1x1(x2, W) + b): ONE weight applied to two activations of different resolution.  Used by the
    MRFP+ head, where final2(Upsample(dec1) + OCout_dec) is evaluated as Upsample(final2(dec1)) + final2(OCout_dec) (a 1x1
    convolution commutes with bilinear interpolation), so the 2x upsample runs on the class scores instead of on 256
    channels.  One Function for both uses keeps the shared weight gradient a single sink write + a single ready
    notification (the gradient-arena protocol of _Conv2d assumes one use per weight)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, Nphys):
        N, C, R, S = weight.shape
        if R != 1 or S != 1 or x1.shape[1] != C or x2.shape[1] != C:
            raise _lib.MrfpHipError("shared conv pair: 1x1 weight and unpadded inputs expected")
        pk = get_pack(weight, bias, x1.dtype, C, Nphys)
        ys = []
        for x, b in ((x1, None), (x2, pk.bias)):
            B, _, H, W = x.shape
            y = empty_cl(B, Nphys, H, W, x.dtype, x.device)
            call("mrfp_conv_fwd", ptr(x), ptr(pk.wf), ptr(b), ptr(y), dt(x), B, H, W, C, Nphys, Nphys, 1, 1, H, W, 1, 0, 0, 1, 1,
                 None, None, stream())
            ys.append(y)
        ctx.save_for_backward(x1, x2, weight, bias)
        ctx.Nphys = Nphys
        ctx.set_materialize_grads(False)
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, dy1, dy2):
        x1, x2, weight, bias = ctx.saved_tensors
        N, C, _, _ = weight.shape
        Nphys = ctx.Nphys
        pk = get_pack(weight, bias, x1.dtype, C, Nphys)
        dxs, dws = [], []
        db = None
        for x, dy, need_dx in ((x1, dy1, ctx.needs_input_grad[0]), (x2, dy2, ctx.needs_input_grad[1])):
            if dy is None:
                dxs.append(None)
                continue
            dy = _chk(dy, "dy")
            B, _, H, W = x.shape
            dx = None
            if need_dx:
                dx = empty_cl(B, C, H, W, x.dtype, x.device)
                call("mrfp_conv_fwd", ptr(dy), ptr(pk.wd), None, ptr(dx), dt(dy), B, H, W, Nphys, C, C, 1, 1, H, W, 1, 0, 0, 1, 1,
                     None, None, stream())
            dxs.append(dx)
            if ctx.needs_input_grad[2]:
                M = B * H * W
                ws = torch.empty(int(_lib.lib().mrfp_conv_wgrad_ws_bytes(M, N, C)), dtype=torch.uint8, device=x.device)
                dw = torch.empty((N, C, 1, 1), dtype=torch.float32, device=x.device)
                call("mrfp_conv_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), dt(x), B, H, W, C, C, N, Nphys, 1, 1, H, W, 1, 0, 0, 1,
                     stream())
                dws.append(dw)
        if bias is not None and ctx.needs_input_grad[3] and dy2 is not None:
            from .ops import _stats_fwd
            d2 = _chk(dy2, "dy")
            B2, _, H2, W2 = d2.shape
            nslab, sws = _stats_fwd(d2, None)
            out = torch.empty(4 * Nphys, dtype=torch.float32, device=d2.device)
            call("mrfp_bn_finalize", ptr(sws), B2, nslab, B2 * H2 * W2, Nphys, None, None, 0.0, 0.0, None, None,
                 ptr(out[:Nphys]), ptr(out[Nphys:2 * Nphys]), ptr(out[2 * Nphys:3 * Nphys]), ptr(out[3 * Nphys:]), stream())
            db = (out[:N] * float(B2 * H2 * W2)).to(bias.dtype)       # column mean * count = column sum
        dw = None
        if dws:
            total = dws[0] if len(dws) == 1 else dws[0].add_(dws[1])
            sink = grad_sink(weight)
            if sink is not None:
                sink.copy_(total)
                notify_grad(weight)
            else:
                dw = total.to(weight.dtype)
        return dxs[0], dxs[1], dw, db, None


def shared_conv1x1_pair(x1, x2, weight, bias, phys_out):
    return _SharedConv1x1Pair.apply(_chk(x1), _chk(x2), weight, bias, int(phys_out))


def pad_input_channels(x: torch.Tensor, dtype) -> torch.Tensor:
    """Network input NCHW fp32 -> NHWC `dtype` with the channel count padded to a 16-byte chunk."""
    if not x.is_cuda:
        raise _lib.MrfpHipError("input must live on the GPU (got %s): the HIP path has no CPU fallback" % x.device)
    B, C, H, W = x.shape
    Cpad = _round_up(C, _epc(dtype))
    xs = x.detach()
    if xs.dtype != torch.float32 or not xs.is_contiguous():
        xs = xs.float().contiguous()
    y = empty_cl(B, Cpad, H, W, dtype, x.device)
    call("mrfp_nchw_to_nhwc_pad", ptr(xs), ptr(y), _lib._DT[dtype], B, C, H, W, Cpad, stream())
    return y


def conv2d(x, weight, bias, stride, padding, dilation, phys_out: Optional[int] = None, want_skip: bool = False):
    """x: [B,Cphys,H,W] channels-last (Cphys >= weight.shape[1], extra channels must be zero);
    returns [B,N,Ho,Wo], or the channel-padded [B,phys_out,Ho,Wo] buffer when phys_out is given.
    want_skip: returns (y, x_skip) -- see _Conv2d."""
    st = stride[0] if isinstance(stride, (tuple, list)) else int(stride)
    ph, pw = (padding if isinstance(padding, (tuple, list)) else (int(padding), int(padding)))
    dl = dilation[0] if isinstance(dilation, (tuple, list)) else int(dilation)
    N, C = weight.shape[0], weight.shape[1]
    epc = _epc(x.dtype)
    if x.shape[1] % epc != 0:                        # e.g. a raw 3-channel image: pad (copy) to a chunk
        xp = zeros_cl(x.shape[0], _round_up(x.shape[1], epc), x.shape[2], x.shape[3], x.dtype, x.device)
        xp[:, :x.shape[1]] = x
        x = xp
    if x.shape[1] < C:
        raise _lib.MrfpHipError("conv2d: input has %d channels, weight expects %d" % (x.shape[1], C))
    Nphys = phys_out if phys_out is not None else _round_up(N, epc)
    _LAST_STATS[0] = None
    if _JOIN_QUEUED[0] and not _in_backward():
        _drop_stale_backward_state()
    out = _Conv2d.apply(x, weight, bias, st, ph, pw, dl, Nphys, want_skip)
    y, skip = out if want_skip else (out, None)
    if skip is not None:
        skip._mrfp_skip_alias = True      # its gradient goes to this convolution's dgrad epilogue and nowhere else
        # [operators that consumed the alias so far (ops._chk): a gated gradient needs exactly one, a gated gradient was issued]
        skip._mrfp_uses = [0, False]
        if y.grad_fn is not None:
            y.grad_fn._mrfp_cell = skip._mrfp_uses     # (the autograd node IS the ctx of _Conv2d.backward)
    if _LAST_STATS[0] is not None and (phys_out is not None or Nphys == N):
        y._mrfp_colstats = _LAST_STATS[0]        # consumed by ops.batch_norm_act (statistics pass skipped)
    _LAST_STATS[0] = None
    if phys_out is None and Nphys != N:
        y = y[:, :N].contiguous(memory_format=CL)
    return (y, skip) if want_skip else y
