"""3x3 convs on the planes GEMM: per-tap tiles (stages 2) against the tap-strip variant (stages 4), at the shapes of the path."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
SHAPES = [(352, 512, 180, 60, None), (352, 512, 180, 45, None), (352, 512, 180, 180, None),
          (1360, 2040, 128, 128, None), (704, 1024, 64, 64, None)]
for H, W, Cin, N, _ in SHAPES:
    x = torch.randn(1, H, W, Cin, device="cuda")
    cv = ops.pack_conv(torch.randn(N, Cin, 3, 3) * 0.05, torch.randn(N), "cuda", pad=1)
    xp = ops.split_planes(x)
    line = f"{H}x{W} Cin {Cin:4d} N {N:4d}:"
    base = None
    for bn, stages in ((0, 0), (64, 2), (64, 3), (64, 4), (128, 2), (128, 3), (128, 4)):
        if bn == 64 and N > 128:
            continue
        out = ops.conv2d(xp, cv, bm=128 if bn else 0, bn=bn, stages=stages)
        if base is None:
            base = out.clone()
        err = (out - base).abs().max().item()
        for _ in range(3):
            ops.conv2d(xp, cv, bm=128 if bn else 0, bn=bn, stages=stages, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d(xp, cv, bm=128 if bn else 0, bn=bn, stages=stages, out=out)
        e1.record()
        torch.cuda.synchronize()
        line += f"  [{'auto' if not bn else f'bn{bn} st{stages}'}] {e0.elapsed_time(e1) * 100:7.1f} us (d {err:.1e})"
    print(line, flush=True)
