"""What does the residual read cost in the tile kernels' epilogues?  Same GEMM with and without `res` (planes input and fp32 input),
at MambaIR's out_proj shape and a 3x3 shape.   python tools/res_cost.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
dev = "cuda"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


g = torch.Generator().manual_seed(0)
M = 352 * 512
for (K, N, k) in ((360, 180, 1), (180, 180, 1), (64, 180, 3), (180, 540, 1)):
    x = torch.randn(1, 352, 512, K, generator=g).to(dev)
    w = torch.randn(N, K, k, k, generator=g) / (K * k * k) ** 0.5
    cv = ops.pack_conv(w, torch.zeros(N), dev)
    res = torch.randn(1, 352, 512, N, generator=g).to(dev)
    xp = ops.split_planes(x)
    a = timeit(lambda: ops.conv2d(xp, cv))
    b = timeit(lambda: ops.conv2d(xp, cv, res=res))
    c = timeit(lambda: ops.conv2d(x, cv))
    d = timeit(lambda: ops.conv2d(x, cv, res=res))
    floor = M * N * 4 / 5e6
    print(f"K={K} N={N} k={k}: planes {a:6.1f} us, + res {b:6.1f} (+{b - a:5.1f}; the read alone is {floor:4.1f} us at 5 TB/s)   "
          f"fp32 input {c:6.1f} us, + res {d:6.1f} (+{d - c:5.1f})", flush=True)
