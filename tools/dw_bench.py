"""Depthwise 3x3 kernels at the shapes of the 340x510 pipeline: NAFNet's gated form (4 levels) and MambaIR's SiLU form; reports
microseconds and the algorithmic HBM rate (read + write once).   python tools/dw_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
dev = "cuda"


def timeit(fn, n=10):
    for _ in range(2):
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
for (H, W, C) in ((1408, 2048, 64), (1024, 1024, 64), (512, 512, 128), (704, 1024, 128), (352, 512, 256), (176, 256, 512), (88, 128, 1024)):
    x = torch.randn(1, H, W, 2 * C, generator=g).to(dev)
    dw = ops.pack_dwconv(torch.randn(2 * C, 1, 3, 3, generator=g), torch.randn(2 * C, generator=g), dev)
    us = timeit(lambda: ops.dw3x3_gate_pool(x, dw))
    print(f"gate  {H}x{W}x{2 * C} -> {C}: {us:7.1f} us  {H * W * 3 * C * 4 / us / 1e6:.2f} TB/s", flush=True)
H, W, C = 352, 512, 360
xz = torch.randn(1, H, W, 2 * C, generator=g).to(dev)
dw = ops.pack_dwconv(torch.randn(C, 1, 3, 3, generator=g), torch.randn(C, generator=g), dev)
us = timeit(lambda: ops.dwconv2d(xz[..., :C], dw, act=ops.ACT_SILU))
print(f"silu  {H}x{W}x{C} (row stride {2 * C}): {us:7.1f} us  {H * W * 2 * C * 4 / us / 1e6:.2f} TB/s")
