"""ctypes wrapper of oracle/csrc/scan_ref.c (TEST INFRASTRUCTURE; same "parity unpinned" status as scan.py).

The pure-torch loop in scan.py is the readable definition; this C build is the same recurrence, used where the
Python loop is too slow (bench.py's cpu_baseline, full-size oracle runs).  tests/test_oracle_golden.py checks that
the two agree."""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(_HERE), "csrc", "scan_ref.c")
LIB = os.path.join(os.path.dirname(_HERE), "_build", "libscan_ref.so")
_lib = None


def _cpu_tag():
    """identifies the instruction set the library was compiled for (-march=native): the flags line of /proc/cpuinfo"""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((ln for ln in f if ln.startswith("flags")), "")
    except OSError:
        flags = ""
    return hashlib.sha1(flags.encode()).hexdigest()


def build(force=False):
    """gcc -O3 -march=native -fopenmp (+ libmvec's vector expf).  The .so travels with the repo snapshot to the GPU box,
    whose host CPU may differ from the build container's: a tag file records the CPU flags the library was built for and
    a mismatch triggers a rebuild there."""
    tag_path, tag = LIB + ".tag", _cpu_tag()
    stale = not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC)
    if not stale:
        try:
            stale = open(tag_path).read().strip() != tag
        except OSError:
            stale = True
    if force or stale:
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fno-math-errno", "-shared", "-fPIC", SRC, "-o", LIB,
                               "-lmvec", "-lm"])
        with open(tag_path, "w") as f:
            f.write(tag)
    return LIB


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.ffsr_oracle_selective_scan.restype = ctypes.c_int
    return _lib


def selective_scan_c(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                     return_last_state=False):
    assert z is None and not return_last_state
    f = lambda t: None if t is None else t.detach().float().contiguous()
    u, delta, A, B, C, D, delta_bias = map(f, (u, delta, A, B, C, D, delta_bias))
    Bsz, Dm, L = u.shape
    y = torch.empty_like(u)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    rc = _load().ffsr_oracle_selective_scan(p(u), p(delta), p(A), p(B), p(C), p(D), p(delta_bias), p(y), Bsz, Dm, L,
                                            A.shape[1], B.shape[1], int(delta_softplus))
    assert rc == 0
    return y
