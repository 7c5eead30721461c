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
from .ops import CL, _chk, empty_cl, grad_sink, notify_grad, zeros_cl

_PACKS = {}      # id(weight Parameter) -> {key: _Pack}; entry dropped when the Parameter dies
import os as _os
FUSE_STATS = [_os.environ.get("MRFP_FUSE_STATS", "1") != "0"]   # conv epilogues emit BatchNorm partial statistics for bias-free convolutions
_LAST_STATS = [None]  # handed from _Conv2d.forward to conv2d() (autograd re-wraps the output tensor object)
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


def _packable(pk, w):
    return (w is not None and pk.bias is None and w.dtype == torch.float32 and w.is_contiguous()
            and w.shape[2] * w.shape[3] <= 9)        # the brick kernel holds up to 3x3 taps; the 7x7 stem packs lazily


def _batched_repack(todo, tag):
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
            st = {"sig": sig, "jobs": torch.from_numpy(rec.view(np.uint8).copy()).to(dev),
                  "prefix": torch.from_numpy(prefix).to(dev), "total": int(prefix[-1]), "n": len(items)}
            _BATCH[(tag, dtype)] = st
        call("mrfp_pack_weights_batched", ptr(st["jobs"]), ptr(st["prefix"]), st["n"], st["total"], _lib._DT[dtype], stream())
        if not torch.cuda.is_current_stream_capturing():
            for key, pk, w in items:
                pk.version = (w._version, 0, _EPOCH[0])


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


def repack_weights(weights, tag="list"):
    """The same for an explicit list of weights that were just rewritten (the HRFP branch's convolutions after their
    re-initialisation: reference deepv3.py:290-299 re-draws them at the start of a forward; 28 pack launches per step
    otherwise).  Weights without a cached pack yet are left to the lazy path."""
    todo = []
    for w in weights:
        for key, pk in _PACKS.get(id(w), {}).items():
            if _packable(pk, w) and pk.wf is not None:
                todo.append((key, pk, w))
    if todo:
        _batched_repack(todo, tag)


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
        if bias is None and FUSE_STATS[0] and L.mrfp_conv_single_launch(B, H * W * Cphys * x.element_size()):
            # no bias = a convolution that feeds a normalisation layer: let the epilogue produce its statistics
            nblk = int(L.mrfp_conv_stats_blocks(dt(x), B, H, W, Cphys, Nphys, R, S, Ho, Wo, stride, pad_h, pad_w, dil, 1))
            stats = torch.empty(int(L.mrfp_conv_stats_rows(nblk)) * 2 * Nphys, dtype=torch.float32, device=x.device)
        call("mrfp_conv_fwd", ptr(x), ptr(pk.wf), ptr(pk.bias), ptr(y), dt(x), B, H, W, Cphys, Nphys, Nphys, R, S,
             Ho, Wo, stride, pad_h, pad_w, dil, 1, None, ptr(stats), stream())
        if stats is not None:      # the rows the BatchNorm finalize should read (compacted for large launches)
            first, cnt = int(L.mrfp_conv_stats_final_first(nblk)), int(L.mrfp_conv_stats_final_count(nblk))
            _LAST_STATS[0] = (stats[first * 2 * Nphys:(first + cnt) * 2 * Nphys], cnt, B * Ho * Wo)
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
            if side is not None:
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
    out = _Conv2d.apply(x, weight, bias, st, ph, pw, dl, Nphys, want_skip)
    y, skip = out if want_skip else (out, None)
    if skip is not None:
        skip._mrfp_skip_alias = True      # its gradient goes to this convolution's dgrad epilogue and nowhere else
        skip._mrfp_uses = [0]             # operators that consumed the alias so far (ops._chk): a gated gradient needs exactly one
    if _LAST_STATS[0] is not None and (phys_out is not None or Nphys == N):
        y._mrfp_colstats = _LAST_STATS[0]        # consumed by ops.batch_norm_act (statistics pass skipped)
    _LAST_STATS[0] = None
    if phys_out is None and Nphys != N:
        y = y[:, :N].contiguous(memory_format=CL)
    return (y, skip) if want_skip else y
