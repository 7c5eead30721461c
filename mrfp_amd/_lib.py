"""ctypes binding of libmrfp_hip.so (C ABI declared in include/mrfp_hip.h).

The prototypes are parsed from the header itself, so the header is the single source of truth
for the boundary.  There is NO fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch  # noqa: F401  (loads torch's libamdhip64.so.7 first so both sides share one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "mrfp_hip.h")
# MRFP_HIP_LIB: an alternative build of the same library (A/B runs of build-time kernel variants on one GPU box)
LIBPATH = os.environ.get("MRFP_HIP_LIB") or os.path.join(_HERE, "csrc", "libmrfp_hip.so")

F32, BF16, F16 = 0, 1, 2
_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}


class MrfpHipError(RuntimeError):
    pass


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes])} for every `mrfp_*` prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int64_t|int)\s+(mrfp_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = {"int": ctypes.c_int, "int64_t": ctypes.c_int64}.get(ret.strip(), ctypes.c_char_p)
        argtypes = []
        names = []
        for a in args.split(","):
            a = a.strip()
            if a in ("", "void"):
                continue
            names.append(re.split(r"[\s\*]+", a)[-1])
            if "*" in a:
                argtypes.append(ctypes.c_void_p)
            elif a.startswith("int64_t"):
                argtypes.append(ctypes.c_int64)
            elif a.startswith("float"):
                argtypes.append(ctypes.c_float)
            elif a.startswith("int"):
                argtypes.append(ctypes.c_int)
            else:
                raise MrfpHipError("cannot parse argument %r of %s" % (a, name))
        protos[name] = (restype, argtypes)
        ARG_NAMES[name] = names
    return protos


ARG_NAMES = {}       # entry point -> argument names as the header spells them (measurement tooling: bench.py)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise MrfpHipError(
                "libmrfp_hip.so is not built (%s). Run `python -m mrfp_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no fallback path." % LIBPATH)
        cdll = ctypes.CDLL(LIBPATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(cdll, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = restype, argtypes
        _lib = cdll
    return _lib


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise MrfpHipError("unsupported activation dtype %s (float32 / bfloat16 / float16 only)" % t.dtype)


def ptr(t):
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    """hipStream_t of torch's current stream on the current device (also inside `torch.cuda.stream(...)` blocks, on
    autograd worker threads and under graph capture).  The raw-handle query costs ~0.3 us; building a
    torch.cuda.Stream object per launch cost ~9 us x 2900 launches per step."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def source_hash() -> str:
    """sha256 (16 hex digits) over the kernel sources, the C header and the operator layer: what a measured figure (PMC traffic in
    profiles/) is valid for.  Works without .git (the GPU box has none)."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = [HEADER] + [os.path.join(_HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(_HERE, "csrc")))
                        if f.endswith((".hip", ".hpp"))]
    files += [os.path.join(_HERE, f) for f in ("ops.py", "conv.py", "deepv3.py", "harness.py")]
    for f in files:
        h.update(os.path.relpath(f, root).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_FN = {}
HOOK = [None]        # measurement only (bench.py): hook(name, args) runs right BEFORE the entry point is called
NOTE = [None]        # measurement only: what the caller knows beyond the C arguments of its NEXT call -- (logical input channels,
                     # logical output channels) of a convolution launch over channel-padded buffers; read and cleared by the hook


def call(name: str, *args):
    """Calls an int-returning entry point and raises with mrfp_last_error() on failure."""
    if HOOK[0] is not None:
        HOOK[0](name, args)
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise MrfpHipError("%s failed (%d): %s" % (name, rc, lib().mrfp_last_error().decode()))
