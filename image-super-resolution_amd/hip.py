"""ctypes binding of libffsr_hip.so -- the C ABI declared in include/ffsr.h.

Prototypes are parsed from the header so the binding cannot drift from the declared ABI.  There is NO
fallback: if the shared library is missing, ``lib()`` raises (the product path must fail loudly).
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FFSR_LIB") or os.path.join(_HERE, "libffsr_hip.so")   # FFSR_LIB: A/B builds of the same library (tools/ab)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ffsr.h")

_CTYPES = {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong, "double": ctypes.c_double}
_lib = None


class FfsrError(RuntimeError):
    pass


def parse_header(path: str = HEADER_PATH):
    """-> {name: [ctypes types]} for every `int ffsr_*(...)` declaration."""
    text = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(ffsr_\w+)\s*\(([^)]*)\)\s*;", text):
        args = []
        for a in m.group(2).split(","):
            a = " ".join(a.split())
            if "*" in a:
                args.append(ctypes.c_void_p)
            else:
                ty = a.rsplit(" ", 1)[0].replace("const ", "").strip()
                args.append(_CTYPES[ty])
        protos[m.group(1)] = args
    return protos


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FfsrError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C image-super-resolution_amd/csrc` (there is no CPU fallback)")
        # torch ships its own HIP runtime: it must be the one this process binds (the library launches on torch's streams),
        # so make sure it is loaded before our .so pulls in a libamdhip64 by SONAME
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in parse_header().items():
            fn = getattr(handle, name)      # AttributeError if the .so does not export a declared symbol
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        if os.environ.get("FFSR_GEMM_MODE") == "bf16":      # plain-bf16 products (see ops.set_gemm_mode)
            handle.ffsr_set_gemm_terms(1)
        _lib = handle
    return _lib


def call(name: str, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise FfsrError(f"{name} returned {rc} ({'invalid argument' if rc == -1 else 'HIP launch error'})")
