"""Where does the capture of ops._AlgebraGraph abort when it is begun inside the autograd machinery?  (first finding.)
Each variant runs in its own child process under faulthandler; the parent prints the exit status and the tail of the child's stderr.
    python tools/experiments/graph_dbg4.py            # all variants
    python tools/experiments/graph_dbg4.py <variant>  # one, in this process
Variants: plain (the shipped placement) | nograd (plain frame, grad mode off) | infn (inside Function.forward) |
infn_grad (inside Function.forward under torch.enable_grad()) | infn_fwdonly (inside Function.forward, forward graph only) |
inbwd (inside a backward pass: Function.backward)."""
import subprocess
import sys

VARIANTS = ["plain", "nograd", "infn_fwdonly", "infn_grad", "infn", "inbwd"]


def child(which):
    import faulthandler
    faulthandler.enable()
    import torch
    sys.path.insert(0, ".")
    from mrfp_amd import ops
    from mrfp_amd.network.sync_switchwhiten import SwitchWhiten2d
    dev = "cuda:0"
    sw = SwitchWhiten2d(64, num_pergroup=16, sw_type=2).to(dev).train()
    N, C, H, W = 4, 64, 12, 10
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    hw = float(H * W)

    def algebra(s, M):
        sq = torch.diagonal(M, dim1=-2, dim2=-1).sum((1, 2))
        return sw._transform(s / hw, M, sq, N, C, hw)
    params = [sw.sw_mean_weight, sw.sw_var_weight, sw.weight, sw.bias]
    bufs = (sw.running_mean, sw.running_cov)

    def build(fwd_only=False):
        s, M = ops._gm_call(x, x)
        if fwd_only:
            sl, Ml = s.clone().requires_grad_(True), M.clone().requires_grad_(True)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    with torch.enable_grad():
                        algebra(sl, Ml)
            torch.cuda.current_stream().wait_stream(side)
            gg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gg):
                with torch.enable_grad():
                    algebra(sl, Ml)
            return gg
        return ops._AlgebraGraph(algebra, s, M, params, bufs)

    class InFwd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t, mode):
            if mode == "infn_grad":
                with torch.enable_grad():
                    build()
            else:
                build(fwd_only=(mode == "infn_fwdonly"))
            return t * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2, None

    class InBwd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t * 2

        @staticmethod
        def backward(ctx, g):
            with torch.enable_grad():
                build()
            return g * 2

    t = torch.ones(4, device=dev, requires_grad=True)
    if which == "plain":
        build()
    elif which == "nograd":
        with torch.no_grad():
            build()
    elif which in ("infn", "infn_grad", "infn_fwdonly"):
        InFwd.apply(t, which)
    elif which == "inbwd":
        InBwd.apply(t).sum().backward()
    torch.cuda.synchronize()
    print(which, "OK", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    for v in VARIANTS:
        try:
            r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True, timeout=240)
            rc, err, out = r.returncode, r.stderr, r.stdout
        except subprocess.TimeoutExpired as e:
            rc, err, out = "timeout", (e.stderr or b"").decode() if isinstance(e.stderr, bytes) else str(e.stderr), ""
        print("=== %s: rc %s  stdout %r" % (v, rc, out.strip()[-200:]), flush=True)
        if rc != 0:
            print("\n".join(err.strip().splitlines()[-40:]), flush=True)
