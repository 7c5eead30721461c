"""The two-group long-K pointwise kernel with the early start on the first weight quarter (default build) against the round-5 form (-DMRFP_PWK2_EARLY=0 variant
library): plain, fused statistics, addend, gated addend -- output (and statistics rows) must be EQUAL bit for bit; several M (whole ranges,
ragged tail, one tile per workgroup), repeated launches.  (The switch lives in tools/experiments/conv_pwk_early.patch -- apply it first; the same
script checked conv_pwk_defer.patch with -DMRFP_PWK2_DEFER=0.  Both: 40 of 40 hashes equal, gpurun_out/r6_pwk_*_check.txt.)   python tools/experiments/pwk_early_check.py   (builds the variant itself)"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child():
    import torch
    from mrfp_amd import _lib, conv
    from mrfp_amd._lib import call, ptr, stream
    L = _lib.lib()
    out = []
    for (B, H, W, N) in [(16, 48, 48, 256), (3, 31, 29, 256), (1, 16, 8, 128), (5, 48, 40, 384)]:
        C = 1024
        g = torch.Generator().manual_seed(B * 7 + N)
        x = torch.randn(B, C, H, W, generator=g).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(N, C, 1, 1, generator=g) * 0.05).cuda()
        pk = conv.get_pack(w, None, torch.bfloat16, C, N)
        add = torch.randn(B, N, H, W, generator=g).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
        mask = torch.randint(0, 256, (B * H * W * N // 8,), generator=g, dtype=torch.uint8).cuda()
        geo = (B, H, W, C, N, N, 1, 1, H, W, 1, 0, 0, 1, 1)
        for rep in range(2):
            y = torch.empty_like(add)
            call("mrfp_conv_fwd", ptr(x), ptr(pk.wf), None, ptr(y), _lib.BF16, *geo, None, None, stream())
            out.append(y.clone())
            nblk = int(L.mrfp_conv_stats_blocks(_lib.BF16, B, H, W, C, N, 1, 1, H, W, 1, 0, 0, 1, 1))
            st = torch.zeros(int(L.mrfp_conv_stats_rows(nblk)) * 2 * N, dtype=torch.float32, device="cuda")
            y = torch.empty_like(add)
            call("mrfp_conv_fwd", ptr(x), ptr(pk.wf), None, ptr(y), _lib.BF16, *geo, None, ptr(st), stream())
            out += [y.clone(), st.clone()]
            y = torch.empty_like(add)
            call("mrfp_conv_fwd", ptr(x), ptr(pk.wf), None, ptr(y), _lib.BF16, *geo, ptr(add), None, stream())
            out.append(y.clone())
            y = torch.empty_like(add)
            call("mrfp_conv_fwd_gated", ptr(x), ptr(pk.wf), None, ptr(y), _lib.BF16, *geo, ptr(add), ptr(mask), stream())
            out.append(y.clone())
        ref = torch.nn.functional.conv2d(x.float(), w.bfloat16().float())
        print("shape", (B, H, W, N), "plain rel err vs torch fp32 %.2e" % ((out[-5].float() - ref).abs().max() / ref.abs().max()).item(), flush=True)
    torch.cuda.synchronize()
    for t in out:
        print(hashlib.sha256(t.cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16])


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
        sys.exit(0)
    from mrfp_amd import build
    lib = build.build_variant("noearly", ("conv_pwk",), ["-DMRFP_PWK2_EARLY=0"])
    res = []
    for env in ({}, {"MRFP_HIP_LIB": lib}):
        r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=dict(os.environ, **env))
        print(("default build" if not env else "round-5 form"), "rc", r.returncode)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("shape")))
        if r.returncode:
            print(r.stderr[-1500:])
        res.append([l for l in r.stdout.splitlines() if not l.startswith("shape")])
    print("hash lists equal:", res[0] == res[1] and len(res[0]) > 0, len(res[0]))
    # repeated launches inside one build must agree too (rep 0 vs rep 1 of each shape)
    sys.exit(0 if res[0] == res[1] and len(res[0]) > 0 else 1)
