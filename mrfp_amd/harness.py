"""Train-step and eval harness: the semantics of the reference's driver loops (reference
main.py:822-839 optimiser + poly LR, 857-864 train step, 887-913 eval loop) re-designed for one
process per MI355X:

  * all trainable parameters, their gradients and the momentum live in three flat fp32 arenas
    (parameters keep their names / shapes / state_dict ABI as views), so the optimiser is ONE fused
    HIP kernel and the data-parallel exchange works on contiguous buckets without packing copies;
  * gradient buckets (contiguous arena ranges, filled in reverse forward order) are all-reduced by
    RCCL over xGMI on a side HIP stream as soon as autograd has finished the last tensor of a bucket,
    overlapping the exchange with the rest of backward; no other collective exists on the data path;
  * BatchNorm / NP+ statistics stay per replica: with 16 images per GPU that is exactly the
    statistical population the reference's single-GPU run sees.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib
from ._lib import call, ptr, stream


def poly_lr_factor(it: int, max_iter: int = 40000, power: float = 0.9) -> float:
    """reference main.py:832-839 (LRPolicy)."""
    return math.pow(1 - it / max_iter, power)


class FlatArena:
    """Trainable parameters and their gradients as views into two flat fp32 buffers (every tensor starts
    16-byte aligned).  Device-agnostic host logic: the data-parallel buckets are ranges of `flat_g`."""

    def __init__(self, model: torch.nn.Module):
        every = list(model.parameters())
        self.params = [p for p in every if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        # the reference hands model.parameters() -- frozen HRFP tensors included -- to torch.optim.SGD (main.py:826), so
        # its optimizer.state_dict() numbers parameters by their position in THAT list
        self.n_all = len(every)
        self.all_index = [i for i, p in enumerate(every) if p.requires_grad]
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.n = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + p.numel()].view(p.shape)
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
            p._mrfp_direct = True          # backward kernels may write this gradient slot directly (ops.grad_sink)

    def zero_grad(self):
        self.flat_g.zero_()
        for p, o in zip(self.params, self.offsets):   # keep .grad pointing into the arena
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)


class FlatSGD(FlatArena):
    """torch.optim.SGD(lr, momentum=0.9, weight_decay=5e-4) + LambdaLR(poly 0.9) as ONE fused HIP kernel
    over the flat arenas (reference main.py:826-839, 863-864)."""

    def __init__(self, model: torch.nn.Module, lr=1e-2, momentum=0.9, weight_decay=5e-4, max_iter=40000, power=0.9):
        super().__init__(model)
        self.flat_m = torch.zeros_like(self.flat_p)
        self.base_lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.max_iter, self.power = max_iter, power
        self.it = 0
        self.has_momentum = False        # torch.optim.SGD: the first step copies the gradient into the momentum buffer

    @property
    def lr(self):
        return self.base_lr * poly_lr_factor(self.it, self.max_iter, self.power)

    # ---- checkpoint ABI: the layout of torch.optim.SGD.state_dict() (reference main.py:867 stores it as 'optimizer') ----
    def state_dict(self):
        """torch.optim.SGD.state_dict() of the reference's optimizer after the same number of steps: one param group
        over ALL model parameters (frozen ones have no state entry), per-parameter `momentum_buffer` cut from the
        momentum arena, `lr` = current scheduled rate, `initial_lr` = base rate (LambdaLR adds it).  The extra key
        `mrfp_iteration` carries the scheduler position, which the reference does not save (its resume restarts the
        poly schedule); loading a plain torch state dict recovers it from lr / initial_lr."""
        state = {}
        if self.has_momentum:
            for i, p, o in zip(self.all_index, self.params, self.offsets):
                state[i] = {"momentum_buffer": self.flat_m[o:o + p.numel()].view(p.shape).detach().clone().cpu()}
        group = {"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.weight_decay,
                 "nesterov": False, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "initial_lr": self.base_lr, "params": list(range(self.n_all))}
        return {"state": state, "param_groups": [group], "mrfp_iteration": self.it}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        self.momentum, self.weight_decay = float(group["momentum"]), float(group["weight_decay"])
        self.base_lr = float(group.get("initial_lr", group["lr"]))
        if "mrfp_iteration" in sd:
            self.it = int(sd["mrfp_iteration"])
        else:                                   # a checkpoint written by torch.optim.SGD + LambdaLR: invert the poly factor
            f = float(group["lr"]) / self.base_lr if self.base_lr > 0 else 1.0
            self.it = int(round(self.max_iter * (1.0 - min(max(f, 0.0), 1.0) ** (1.0 / self.power))))
        state = sd.get("state", {})
        # torch numbers the saved parameters 0..P-1 in param_groups[0]["params"] order = model.parameters() order
        slot = {j: t for t, j in enumerate(self.all_index)}          # position in model.parameters() -> arena slot
        self.flat_m.zero_()
        n = 0
        for key, st in state.items():
            t = slot.get(group["params"].index(int(key)) if int(key) in group["params"] else -1)
            buf = st.get("momentum_buffer")
            if t is None or buf is None:
                continue
            p, o = self.params[t], self.offsets[t]
            if buf.numel() != p.numel():
                raise _lib.MrfpHipError("optimizer state %s: momentum_buffer has %d values, the parameter %d"
                                        % (key, buf.numel(), p.numel()))
            self.flat_m[o:o + p.numel()].copy_(buf.reshape(-1).to(self.flat_m.device, torch.float32))
            n += 1
        # torch.optim.SGD keeps no state for a parameter that never received a gradient: its momentum stays zero here
        # (what torch's first step on it would start from, up to the dampening-free first-step copy)
        self.missing_state = len(self.params) - n if n else 0
        self.has_momentum = n > 0

    def step(self, gscale: float = 1.0):
        if self.flat_p.device.type != "cuda":
            raise _lib.MrfpHipError("FlatSGD.step needs the arenas on the GPU: the HIP path has no CPU fallback")
        call("mrfp_sgd_step", ptr(self.flat_p), ptr(self.flat_g), ptr(self.flat_m), self.n, float(self.lr),
             float(self.momentum), float(self.weight_decay), float(gscale), int(not self.has_momentum), stream())
        self.has_momentum = True
        # the fused kernel wrote the arena behind autograd's version counters: invalidate the derived
        # weight packs explicitly (mrfp_amd/conv.py rebuilds them on next use)
        from . import conv
        conv.invalidate_packs()
        conv.repack_all()                             # one launch for every bias-free trainable pack
        self.it += 1                                  # scheduler.step()


class GradSync:
    """Bucketed RCCL all-reduce of the flat gradient arena, overlapped with backward."""

    def __init__(self, opt: FlatArena, bucket_mb: float = 32.0, group=None):
        self.opt, self.group = opt, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # MRFP_FORCE_SYNC=1: run the bucket / side-stream / collective machinery even with one rank (rehearsal of the
        # multi-GPU path on a single-GPU box)
        self.enabled = self.world > 1 or (dist.is_initialized() and os.environ.get("MRFP_FORCE_SYNC") == "1")
        self.buckets: List[tuple] = []
        if not self.enabled:
            return
        cap = int(bucket_mb * (1 << 20) / 4)
        # buckets are contiguous arena ranges, cut from the END (gradients arrive in reverse forward order)
        end = opt.n
        members: List[int] = []
        self.bucket_of = [0] * len(opt.params)
        for i in range(len(opt.params) - 1, -1, -1):
            members.append(i)
            if end - opt.offsets[i] >= cap or i == 0:
                self.buckets.append((opt.offsets[i], end, list(members)))
                end, members = opt.offsets[i], []
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self.bucket_of[i] = b
        self.pending = [0] * len(self.buckets)
        self.seen = [False] * len(opt.params)
        self.next = 0                    # buckets are launched strictly in index order on every rank
        self.works = []
        self.on_gpu = opt.flat_g.is_cuda
        self.side = torch.cuda.Stream() if self.on_gpu else None
        self._index = {id(p): i for i, p in enumerate(opt.params)}
        from . import ops
        for i, p in enumerate(opt.params):
            p.register_post_accumulate_grad_hook(self._make_autograd_hook(i, ops.GRAD_DEFERRED))
        ops.GRAD_NOTIFY[0] = self._notify     # gradients written straight into the arena by the HIP backward kernels

    def _notify(self, param):
        i = self._index.get(id(param))
        if i is not None:
            self._make_hook(i)(param)

    def _make_autograd_hook(self, i, deferred):
        """autograd's post-accumulate hook: fires when the parameter's backward node has RUN -- also when that node returned None
        because its kernel writes the arena directly, and also when it only QUEUED the weight-gradient launch (conv._queue_wgrad:
        grouped, deferred weight gradients).  A queued gradient has not been written: it is reported by ops.notify_grad when its
        launch has been issued, and ignored here."""
        inner = self._make_hook(i)

        def hook(param):
            if id(param) in deferred:
                return
            inner(param)
        return hook

    def _make_hook(self, i):
        def hook(_param):
            # A parameter can be reported twice in one step: by the HIP backward kernel that wrote its gradient
            # straight into the arena (ops.notify_grad) and by autograd's post-accumulate hook (this PyTorch fires
            # it even when the backward returned None for the parameter).  Count each parameter once.
            if self.seen[i]:
                return
            self.seen[i] = True
            self.pending[self.bucket_of[i]] -= 1
            self._launch_ready()
        return hook

    def _launch_ready(self):
        """Collectives must be issued in the same order on every rank, whatever order the gradients arrive in (ranks
        may draw different perturbation toggles, a tensor may get no gradient on one rank): a bucket is launched only
        when every bucket before it has been launched -- bucket 0 holds the LAST parameters, whose gradients arrive
        first, so in the common case this is still "as soon as the bucket is complete"."""
        while self.next < len(self.buckets) and self.pending[self.next] == 0:
            self._launch(self.next)
            self.next += 1

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        if not self.on_gpu:                      # gloo / CPU arenas (tests): no streams involved
            self.works.append(dist.all_reduce(self.opt.flat_g[lo:hi], group=self.group, async_op=True))
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        from . import conv
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            conv.join_wgrad_stream(self.side)        # the bucket's weight gradients are written on the wgrad stream
            self.works.append(dist.all_reduce(self.opt.flat_g[lo:hi], group=self.group, async_op=True))

    def begin(self):
        if self.enabled:
            self.pending = [len(m) for _, _, m in self.buckets]
            self.seen = [False] * len(self.opt.params)
            self.next = 0
            self.works = []

    def finish(self):
        """Blocks the compute stream until every bucket has been reduced; returns the 1/world scale."""
        from . import conv
        conv.flush_wgrads()                          # (queued grouped weight gradients: backward's end callback has issued them already)
        conv.join_wgrad_stream()                     # weight gradients are computed on a second stream (mrfp_amd/conv.py)
        if not self.enabled:
            return 1.0
        for b in range(self.next, len(self.buckets)):   # buckets with tensors that got no gradient this step: same order
            self.pending[b] = 0
        self._launch_ready()
        for w in self.works:
            w.wait()
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.side)
        return 1.0 / self.world


def sync_replicas(model, arena: Optional[FlatArena] = None, group=None, src: int = 0):
    """What DistributedDataParallel does at construction: every rank starts from rank `src`'s parameters and buffers
    (MRFPPlus.__init__ draws its initial weights from the unseeded global RNG, and a missing pretrained checkpoint keeps
    them: without this, replicas would differ for ever because only gradients are averaged).  Also makes the three
    perturbation toggles of MRFPPlus.forward agree across ranks for the whole run: the reference seeds python's
    `random` identically everywhere (main.py:38); here rank `src` draws one seed and every rank gets a private
    `random.Random(seed)` for the toggles -- no per-step collective, no host synchronisation.
    The perturbation DRAWS, by contrast, differ per rank (reference deepv3.py:272-275, 291-306: NP+
    normals and the HRFP re-initialisation come from the torch generators): every torch generator of this process is
    seeded `base + rank` with one base broadcast from `src`, so N replicas apply N different perturbations."""
    if arena is not None:
        dist.broadcast(arena.flat_p, src, group=group)
    with torch.no_grad():
        for p in model.parameters():
            if arena is None or not p.requires_grad:
                dist.broadcast(p.data, src, group=group)
        for b in model.buffers():
            dist.broadcast(b.data, src, group=group)
    from . import conv
    conv.invalidate_packs()
    rng = getattr(model, "rng", None)
    if rng is not None and hasattr(rng, "seed_toggles"):
        import random
        seed = [random.getrandbits(62) if dist.get_rank(group) == src else 0]
        dist.broadcast_object_list(seed, src, group=group)
        rng.seed_toggles(seed[0])
    base = [random_base_seed() if dist.get_rank(group) == src else 0]
    dist.broadcast_object_list(base, src, group=group)
    torch.manual_seed(base[0] + dist.get_rank(group))          # CPU + every device generator of this process
    return base[0]


def random_base_seed() -> int:
    """Base of the per-rank torch seeds: MRFP_SEED when set (reproducible runs), else drawn from the OS."""
    import random
    v = os.environ.get("MRFP_SEED")
    return int(v) if v is not None else random.SystemRandom().getrandbits(48)


class Trainer:
    """zero_grad -> forward -> backward (+ overlapped all-reduce) -> fused SGD step -> LR step."""

    def __init__(self, model, lr=1e-2, momentum=0.9, weight_decay=5e-4, max_iter=40000, bucket_mb=32.0,
                 loss_scale=None):
        """loss_scale: static scale of the backward pass for float16 activations (the per-pixel CE gradient is
        1/#pixels ~ 1e-7, below float16's normal range); default 65536 when cfg.MODEL.ACT_DTYPE is float16, else 1.
        The scale enters through the gradient of the loss (the CE backward kernel multiplies by it) and leaves in the
        fused SGD kernel's gradient scale, so parameters see exactly the unscaled step."""
        from .config import cfg
        self.model = model
        self.opt = FlatSGD(model, lr, momentum, weight_decay, max_iter)
        self.sync = GradSync(self.opt, bucket_mb)
        # (MRFP_FORCE_SYNC=1: the replica synchronisation -- parameter / buffer broadcasts, the shared toggle seed, the per-rank
        #  torch seeds -- also runs at world size 1: the rehearsal of every collective of the multi-GPU path on one GPU over RCCL)
        if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("MRFP_FORCE_SYNC") == "1"):
            sync_replicas(model, self.opt)
        if loss_scale is None:
            loss_scale = 65536.0 if cfg.MODEL.ACT_DTYPE == torch.float16 else 1.0
        self.loss_scale = float(loss_scale)
        self.graph = False

    def _fwd_bwd(self, img, label):
        from . import conv
        self.opt.zero_grad()
        self.sync.begin()
        ok = False
        try:
            loss = self.model(img, label, training=True)
            if self.loss_scale != 1.0:
                loss.backward(torch.full_like(loss, self.loss_scale))
            else:
                loss.backward()
            ok = True
        finally:
            if not ok:          # a pass that raised leaves queued weight gradients and a pending end-of-backward callback behind
                conv.backward_failed()
        return loss

    def step(self, img, label):
        if self.graph:
            return self._graph_step(img, label)
        loss = self._fwd_bwd(img, label)
        gscale = self.sync.finish()
        from . import conv, ops
        conv.join_wgrad_stream()                       # weight gradients are computed on a second stream
        self.opt.step(gscale / self.loss_scale)
        if ops._SYNC_BN_FLAG:                          # cfg.MODEL.SYNC_BN over several ranks: the device-side count check (no host wait)
            ops.sync_bn_poll()
        return loss

    # ---- hipGraph mode --------------------------------------------------------------------------------------------
    # zero_grad + forward + backward (~2900 kernel launches, ~28 ms of Python + ctypes per ResNet-101 step) are captured
    # once per (perturbation toggle combination, input shape) and replayed with one hipGraphLaunch; the optimizer step
    # (learning rate and first-step flag are by-value kernel arguments) and the batched weight repack stay eager.
    # Single-process only: the overlapped all-reduce of GradSync is driven by Python hooks.
    def enable_graph(self):
        if self.sync.enabled:
            raise _lib.MrfpHipError("Trainer graph mode is single-process (the overlapped all-reduce is hook-driven)")
        self.graph = True
        self._graphs = {}
        self._bns = [m for m in self.model.modules() if hasattr(m, "_nbt_pending")]
        return self

    def _graph_step(self, img, label):
        rng = self.model.rng
        tog = tuple(rng.toggles())
        key = (tuple(t < 0.5 for t in tog), tuple(img.shape), img.dtype, tuple(label.shape))
        entry = self._graphs.get(key)

        class _Fixed:                       # the toggles drawn above, everything else from the model's own rng
            def __init__(self, base, t):
                self._b, self._t = base, t

            def toggles(self):
                return self._t

            def __getattr__(self, name):
                return getattr(self._b, name)

        if entry is None:
            static_img, static_lab = img.clone(), label.clone()
            self.model.rng = _Fixed(rng, tog)
            try:
                # ONE capture stream for every key: autograd's gradient accumulators remember the stream they first ran on, and a
                # capture on another stream is forked / joined once per parameter (a cross-queue barrier each at replay)
                if getattr(self, "_graph_stream", None) is None:
                    self._graph_stream = torch.cuda.Stream()
                side = self._graph_stream
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):           # eager warm-up on a side stream (lazy tables, attributes, packs)
                    loss = self._fwd_bwd(static_img, static_lab).detach()
                torch.cuda.current_stream().wait_stream(side)
                from . import conv
                conv.join_wgrad_stream()
                self.opt.step(1.0 / self.loss_scale)
                conv.prebuild_repack_tables()            # (job tables of the batched re-packs: no host -> device copy inside the capture)
                # capture on the warm-up's stream, with the warm-up's autograd graph gone (.detach() above): a gradient
                # accumulator that remembers another stream makes autograd fork / join the capture once per parameter, and
                # the replay of such a graph pays a cross-queue barrier per fork (35 vs 15 ms on ResNet-50 8x512^2)
                g = torch.cuda.CUDAGraph(keep_graph=True)       # (the hipGraph_t stays readable: graph_topology())
                if getattr(self, "_debug_capture_on_fresh_stream", False):      # tools/graph_dbg.py only: the round-2 capture, for evidence
                    side = torch.cuda.Stream()
                    side.wait_stream(self._graph_stream)
                with torch.cuda.graph(g, stream=side):
                    static_loss = self._fwd_bwd(static_img, static_lab).detach()    # only the value is read at replay: no autograd graph
                                                                                      # (and its accumulator nodes) kept alive per key
            finally:
                self.model.rng = rng
            for m in self._bns:                           # the capture pass ran the Python side of every layer once more
                if m.training and m.num_batches_tracked is not None:
                    m._nbt_pending -= 1
            entry = (g, static_img, static_lab, static_loss)
            self._graphs[key] = entry
            return loss                                   # this call was the eager warm-up step
        g, static_img, static_lab, static_loss = entry
        static_img.copy_(img)
        static_lab.copy_(label)
        g.replay()
        for m in self._bns:
            if m.training and m.num_batches_tracked is not None:
                m._nbt_pending += 1
        self.opt.step(1.0 / self.loss_scale)
        return static_loss


def graph_topology(g):
    """Nodes / dependency edges of a captured torch.cuda.CUDAGraph(keep_graph=True) read back through the HIP runtime
    (hipGraphGetNodes / hipGraphGetEdges): {"nodes", "edges", "forks" (nodes with more than one successor), "joins" (more than
    one predecessor), "roots"}.  A capture that stayed on ONE stream is a chain: forks == joins == 0, one root.  Every stream the
    capture forked to shows up as a fork / join pair -- and each pair becomes a cross-queue dependency at replay (the round-2
    capture had one pair per parameter: see DESIGN.md section 5)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so.7")
    graph = ctypes.c_void_p(int(g.raw_cuda_graph()))
    n = ctypes.c_size_t(0)
    hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    hip.hipGraphGetEdges.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    if hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) != 0:
        raise _lib.MrfpHipError("hipGraphGetNodes failed")
    e = ctypes.c_size_t(0)
    if hip.hipGraphGetEdges(graph, None, None, ctypes.byref(e)) != 0:
        raise _lib.MrfpHipError("hipGraphGetEdges failed")
    src, dst = (ctypes.c_void_p * max(1, e.value))(), (ctypes.c_void_p * max(1, e.value))()
    if e.value and hip.hipGraphGetEdges(graph, src, dst, ctypes.byref(e)) != 0:
        raise _lib.MrfpHipError("hipGraphGetEdges failed")
    out_deg, in_deg = {}, {}
    for i in range(e.value):
        out_deg[src[i]] = out_deg.get(src[i], 0) + 1
        in_deg[dst[i]] = in_deg.get(dst[i], 0) + 1
    return {"nodes": int(n.value), "edges": int(e.value), "forks": sum(1 for v in out_deg.values() if v > 1),
            "joins": sum(1 for v in in_deg.values() if v > 1), "roots": int(n.value) - len(in_deg)}


@torch.no_grad()
def evaluate(model, batches, num_classes=19):
    """reference main.py:887-913: model.eval(), per-image arg-max + confusion histogram (on the device,
    one 19x19 int64 D2H at the end instead of two full-logit copies per image), mIoU as metrics.py:60-85."""
    from . import metrics, ops
    model.eval()
    hist, dropped = None, 0
    for img, label in batches:
        if img.shape[2:] != label.shape[1:]:        # reference main.py:894, 910-912
            dropped += 1
            continue
        logits = model(img, training=False)
        hist, _ = ops.argmax_hist(logits, label, hist)
    if dist.is_initialized() and dist.get_world_size() > 1:
        if hist is None:                              # this rank dropped all of its batches: still take part
            dev = next(model.parameters()).device
            hist = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=dev)
        dist.all_reduce(hist)
    h = hist.cpu().numpy() if hist is not None else None
    return h, (metrics.miou_from_hist(h) if h is not None else 0.0), dropped


# ------------------------------------------------------------------------------------------
# checkpoints in the reference's format (reference main.py:867-869, 884-886)
# ------------------------------------------------------------------------------------------
def save_checkpoint(path, model, epoch, optimizer=None):
    """{'epoch', 'state_dict', 'optimizer'} as reference main.py:867-869 writes it, with the `module.` prefix
    nn.DataParallel gives the keys.  `optimizer`: a FlatSGD / Trainer (its torch.optim.SGD-layout state_dict() is
    stored), a torch optimizer, or an already-built state dict."""
    sd = {"module." + k: v.detach().cpu() for k, v in model.state_dict().items()}
    ck = {"epoch": epoch, "state_dict": sd}
    if optimizer is not None:
        opt = getattr(optimizer, "opt", optimizer)                  # Trainer -> its FlatSGD
        ck["optimizer"] = opt.state_dict() if hasattr(opt, "state_dict") else opt
    torch.save(ck, path)


def load_checkpoint(path_or_dict, model, strict=True, optimizer=None, trust_pickle=False):
    """Loads a reference checkpoint (keys with or without the `module.` prefix) into the HIP model and, when
    `optimizer` (FlatSGD / Trainer) is given and the checkpoint has an 'optimizer' entry, the momentum buffers and the
    schedule position (reference main.py:884-886 + the 'optimizer' entry of main.py:867).  The reference's format
    ({'epoch', 'state_dict', 'optimizer'} of tensors, numbers, lists and dicts) loads under torch's safe unpickler;
    `trust_pickle=True` opts into arbitrary pickles for files of known origin only."""
    ck = (torch.load(path_or_dict, map_location="cpu", weights_only=not trust_pickle)
          if isinstance(path_or_dict, str) else path_or_dict)
    sd = ck["state_dict"] if "state_dict" in ck else ck
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    with torch.no_grad():
        missing = model.load_state_dict(sd, strict=strict)
    from . import conv
    conv.invalidate_packs()
    if optimizer is not None and isinstance(ck, dict) and ck.get("optimizer") is not None:
        getattr(optimizer, "opt", optimizer).load_state_dict(ck["optimizer"])
    return ck.get("epoch", None) if isinstance(ck, dict) else None, missing
