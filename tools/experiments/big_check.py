"""conv_big.hip (MRFP_CONV_BIG=2: wherever legal) against torch: outputs with bias / addend, tails in M and N, stride, dilation;
then timing of a few bench shapes against the default kernels (run once with MRFP_CONV_BIG=0 and once with 2)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from mrfp_amd import conv, _lib
from mrfp_amd._lib import call, ptr, dt, stream
torch.manual_seed(0)
CL = torch.channels_last

def run(B, C, H, W, N, k, st, pad, dil, bias=True, addend=False, reps=0):
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=CL)
    w = (torch.randn(N, C, k, k, device="cuda") * (2.0 / (C * k * k)) ** 0.5)
    b = torch.randn(N, device="cuda") * 0.1 if bias else None
    Ho = (H + 2 * pad - dil * (k - 1) - 1) // st + 1
    Wo = (W + 2 * pad - dil * (k - 1) - 1) // st + 1
    pk = conv.get_pack(w, b, x.dtype, C, N)
    y = torch.empty(B, N, Ho, Wo, device="cuda", dtype=x.dtype).contiguous(memory_format=CL)
    ad = torch.randn(B, N, Ho, Wo, device="cuda").bfloat16().contiguous(memory_format=CL) if addend else None
    def go():
        call("mrfp_conv_fwd", ptr(x), ptr(pk.wf), ptr(pk.bias) if bias else None, ptr(y), dt(x), B, H, W, C, N, N, k, k, Ho, Wo, st, pad, pad, dil, 1,
             ptr(ad), None, stream())
    go()
    torch.cuda.synchronize()
    if reps:
        t0 = time.perf_counter()
        for _ in range(reps): go()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("  %-44s %.3f ms  %.0f TFLOP/s" % ((B, C, H, W, N, k, st, pad, dil), ms, 2.0 * B * Ho * Wo * N * C * k * k / ms / 1e9))
        return
    ref = F.conv2d(x.float(), w.bfloat16().float(), b, st, pad, dil)
    if addend:
        ref = ref.bfloat16().float() + ad.float()
    err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    print("%-44s bias %d addend %d  max rel err %.2e  %s" % ((B, C, H, W, N, k, st, pad, dil), bias, addend, err, "ok" if err < 1.5e-2 else "FAIL"))
    return err < 1.5e-2

print("MRFP_CONV_BIG =", os.environ.get("MRFP_CONV_BIG"))
ok = True
for case in [(4, 64, 96, 96, 256, 3, 1, 1, 1), (2, 128, 100, 90, 320, 3, 1, 2, 2), (16, 512, 48, 48, 2048, 1, 1, 0, 1), (3, 64, 67, 45, 264, 3, 1, 1, 1),
             (2, 64, 128, 128, 512, 3, 2, 1, 1), (1, 192, 33, 65, 256, 1, 1, 0, 1), (2, 64, 60, 60, 256, 5, 1, 2, 1)]:
    ok &= bool(run(*case))
    ok &= bool(run(*case, bias=False, addend=True))
print("all ok" if ok else "SOME FAILED")
if ok:
    for case in [(16, 128, 384, 384, 256, 3, 1, 1, 1), (16, 256, 192, 192, 256, 3, 1, 1, 1), (16, 256, 48, 48, 2048, 3, 1, 6, 6), (16, 512, 48, 48, 2048, 1, 1, 0, 1),
                 (16, 128, 332, 332, 256, 3, 1, 2, 2), (16, 1024, 48, 48, 2048, 1, 1, 0, 1)]:
        run(*case, reps=20)
