"""Builds libmrfp_hip.so (gfx950) in-tree with hipcc.  `python -m mrfp_amd.build`.

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmrfp_hip.so")
ARCH = "gfx950"
# (-fno-slp-vectorize -- no v_pk_add_f32 / v_pk_mul_f32 from the SLP vectoriser -- was measured in round 3: the bench step is the
#  same with and without, 57.61 / 57.63 ms on one box; profiles/r03_experiments.md)
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))



def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "mrfp_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(CSRC, src[:-4] + ".o")
    if _stale(obj, src):
        cmd = ["hipcc", *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, always=()) -> str:
    """force: recompile everything.  always: translation units that are recompiled (and the library relinked) even if
    their object file looks current."""
    srcs = _sources()
    for s in srcs:
        if force or s in always:
            o = os.path.join(CSRC, s[:-4] + ".o")
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


def build_variant(name: str, units, flags) -> str:
    """csrc/libmrfp_hip_<name>.so: the translation units in `units` recompiled with the extra `flags`, every other object of the
    default build reused.  Used for A/B runs of build-time switches on one GPU box (MRFP_HIP_LIB=<path>) and by the
    conservative-wait build the tests compare the counted-vmcnt kernels with (-DMRFP_VMCNT0=1)."""
    build()
    objs = []
    for s in _sources():
        o = os.path.join(CSRC, s[:-4] + ".o")
        if s[:-4] in units:
            o = os.path.join(CSRC, "%s_%s.variant.o" % (s[:-4], name))
            src = os.path.join(CSRC, s)
            if _stale(o, s):                     # same dependency list as the main build: the source, every .hpp, include/mrfp_hip.h
                r = subprocess.run(["hipcc", *FLAGS, *flags, "-c", src, "-o", o], capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed for %s (%s):\n%s\n%s" % (s, name, r.stdout, r.stderr))
        objs.append(o)
    lib = os.path.join(CSRC, "libmrfp_hip_%s.so" % name)
    if not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        r = subprocess.run(["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed (%s):\n%s\n%s" % (name, r.stdout, r.stderr))
    return lib


# conservative-wait build: every counted `s_waitcnt vmcnt(N)` of the asynchronous LDS-DMA rings is vmcnt(0) (conv_common.hpp)
VM0_UNITS = ("conv_pw", "conv_pwk", "conv_c64", "conv_wgrad", "conv_wg1")


def build_vm0() -> str:
    return build_variant("vm0", VM0_UNITS, ["-DMRFP_VMCNT0=1"])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
