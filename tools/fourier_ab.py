"""A/B of the band-limited inverse row pass (MRFP_FFT_MFMA=1 matrix cores / 0 fp32 direct sum): result difference against the fp32
reference of the oracle definition (torch.fft on the GPU, fp32) and timing.  Each setting runs in a child process (the switch is
read once per process):  gpurun -- python tools/fourier_ab.py [B C H W]"""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(B, C, H, W):
    from mrfp_amd import _lib, ops
    from mrfp_amd._lib import call, dt, ptr, stream
    dev = "cuda:0"
    torch.manual_seed(0)
    x = ops.empty_cl(B, C, H, W, torch.bfloat16, dev)
    x.copy_(torch.randn(B, C, H, W, device=dev) * 3 + 1)
    y = ops.empty_cl(B, C, H, W, torch.bfloat16, dev)
    nbytes = int(_lib.lib().mrfp_fourier_spectrum_bytes(B, H, W, C))
    S = torch.empty(nbytes, dtype=torch.uint8, device=dev); S3 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ratio = torch.empty(B * H * (W // 2 + 1) * C, dtype=torch.float32, device=dev)
    perm = torch.roll(torch.arange(B, device=dev), 1).contiguous()
    twH, twW = ops._twiddles(H, dev), ops._twiddles(W, dev)

    def run():
        call("mrfp_fourier_mix", ptr(x), ptr(y), ptr(perm), ptr(S), ptr(S3), ptr(ratio), 0, ptr(twH), ptr(twW),
             dt(x), B, H, W, C, 16.0, 1.0, 0, stream())
    run()
    torch.cuda.synchronize()
    # fp32 reference of the definition (fourier.hip header), from the SAME bf16 input
    xf = x.float()
    F = torch.fft.rfft2(xf)
    A = F.abs(); Ap = A[perm]
    kh = torch.arange(H, device=dev); dh = torch.minimum(kh, H - kh).float(); kw = torch.arange(W // 2 + 1, device=dev).float()
    band = (dh[:, None] ** 2 + kw[None, :] ** 2) <= 256.0
    rat = torch.where(band[None, None] & (A > 1e-20), Ap / A.clamp_min(1e-30), torch.ones_like(A))
    ref = torch.fft.irfft2(F * rat, s=(H, W))
    d = (y.float() - ref)
    ulp = (ref.abs().clamp_min(1e-3) * 2.0 ** -8)
    print("  max |y-ref| %.4g   rel-l2 %.3e   max err in output ulps %.2f   exact-rounding rate %.4f" % (
        d.abs().max().item(), (d.norm() / ref.norm()).item(), (d.abs() / ulp).max().item(),
        (y == ref.to(torch.bfloat16)).float().mean().item()), flush=True)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("  %.4f ms / call = %.3f of 8 TB/s (3 planes)" % (ms, 3 * x.numel() * 2 / (ms * 1e-3) / 8e12), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(*[int(v) for v in sys.argv[2:6]])
    else:
        shape = sys.argv[1:5] if len(sys.argv) >= 5 else ["16", "128", "192", "192"]
        for env in ({"MRFP_FFT_MFMA": "0"}, {"MRFP_FFT_MFMA": "1"}, {"MRFP_FFT_MFMA": "1", "MRFP_FFT_MFMA_WGS": "512"},
                    {"MRFP_FFT_MFMA": "1", "MRFP_FFT_MFMA_WGS": "2048"}):
            print(env, shape, flush=True)
            subprocess.call([sys.executable, __file__, "child"] + shape, env=dict(os.environ, **env))
