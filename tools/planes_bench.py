"""Micro-benchmark: ffsr_conv2d_planes vs the fp32-input split-bf16 kernel on the hot shapes (diagnostic tool).
usage: python tools/planes_bench.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")

SHAPES = [  # (H, W, Cin, N, k)
    (352, 512, 180, 540, 1), (352, 512, 360, 180, 1), (352, 512, 180, 360, 1), (352, 512, 180, 180, 3), (1360, 2040, 128, 128, 3),
]
_OLD = [
    (352, 512, 180, 360, 1), (352, 512, 180, 540, 1), (352, 512, 360, 180, 1), (352, 512, 180, 720, 1),
    (352, 512, 180, 180, 1), (352, 512, 180, 180, 3), (352, 512, 180, 60, 3), (352, 512, 60, 180, 3),
    (1360, 2040, 128, 128, 3), (1408, 2048, 64, 128, 1), (1408, 2048, 64, 64, 1),
]
VARIANTS = [(128, 64, 2), (128, 128, 2), (128, 192, 2), (256, 128, 3), (256, 192, 2)]


COLD = os.environ.get("COLD", "0") == "1"
_flush = None


def timeit(fn, reps):
    global _flush
    fn()
    torch.cuda.synchronize()
    if COLD:   # every rep starts from flushed caches (1 GiB of writes in between), timed on its own
        if _flush is None:
            _flush = torch.empty(1 << 28, device="cuda")
        tot = 0.0
        for i in range(reps):
            _flush.fill_(float(i))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        return tot / reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = "cuda"
    for H, W, Cin, N, k in SHAPES:
        x = torch.randn(1, H, W, Cin, device=dev)
        w = torch.randn(N, Cin, k, k) * 0.05
        cv = ops.pack_conv(w, torch.randn(N), dev)
        flops = 2.0 * H * W * N * Cin * k * k
        out = ops.conv2d(x, cv, tile_hint=64)
        ms = timeit(lambda: ops.conv2d(x, cv, tile_hint=64, out=out), reps)
        line = f"M={H * W:8d} K={Cin * k * k:5d} N={N:4d} k{k}: f32in {ms * 1e3:7.1f}us {flops / ms / 1e9:6.1f}TF |"
        xp = ops.split_planes(x)
        ms = timeit(lambda: ops.split_planes(x, out=xp), reps)
        line += f" split {ms * 1e3:6.1f}us |"
        for bm, bn, st in VARIANTS:
            if bn > 64 and (N + bn - 1) // bn * bn > (N + 63) // 64 * 64 + 64:
                continue
            o2 = ops.conv2d(xp, cv, bm=bm, bn=bn, stages=st)
            err = ((o2 - out).abs().max() / out.abs().max()).item()
            ms = timeit(lambda: ops.conv2d(xp, cv, bm=bm, bn=bn, stages=st, out=o2), reps)
            line += f" {bm}x{bn}s{st} {ms * 1e3:7.1f}us {flops / ms / 1e9:6.1f}TF" + (f" [err {err:.0e}]" if err > 1e-5 else "")
        pl = ops.conv2d(xp, cv, out_planes=True, want_f32=False)
        ms = timeit(lambda: ops.conv2d(xp, cv, out_planes=pl, want_f32=False), reps)
        line += f" | planes-out {ms * 1e3:7.1f}us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
