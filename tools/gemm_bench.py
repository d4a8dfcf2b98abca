"""Micro-benchmark of ffsr_conv2d_f32 on the shapes that dominate the hot path (diagnostic tool).
usage: python tools/gemm_bench.py [reps]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")

SHAPES = [  # (H, W, Cin, N, k)
    (352, 512, 180, 360, 1), (352, 512, 180, 540, 1), (352, 512, 360, 180, 1), (352, 512, 180, 720, 1),
    (352, 512, 308, 308, 1), (352, 512, 180, 180, 3), (352, 512, 180, 60, 3), (1360, 2040, 128, 128, 3),
    (1408, 2048, 64, 128, 1),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    hints = [int(x) for x in os.environ.get("HINTS", "0,1,2,3").split(",")]
    dev = "cuda"
    for H, W, Cin, N, k in SHAPES:
        x = torch.randn(1, H, W, Cin, device=dev)
        w = torch.randn(N, Cin, k, k) * 0.05
        cv = ops.pack_conv(w, torch.randn(N), dev)
        flops = 2.0 * H * W * N * Cin * k * k
        line = f"M={H * W:8d} K={Cin * k * k:5d} N={N:4d} k{k}:"
        for hint in hints:
            if hint == 3 and N > 64:
                continue
            out = ops.conv2d(x, cv, tile_hint=hint)
            torch.cuda.synchronize()
            if hint == hints[0]:
                ref = out.clone()
            else:
                line += f" [err {((out - ref).abs().max() / ref.abs().max()).item():.1e}]"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv2d(x, cv, tile_hint=hint, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            line += f"  h{hint}: {ms * 1e3:7.1f}us {flops / ms / 1e9:6.1f}TF"
        print(line, flush=True)


if __name__ == "__main__":
    main()
