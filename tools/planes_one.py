"""Run one conv shape N times on the planes GEMM (for rocprofv3 --pmc): python tools/planes_one.py H W Cin N k bn stages reps"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
H, W, Cin, N, k, bn, stages, reps = (int(v) for v in sys.argv[1:9])
x = torch.randn(1, H, W, Cin, device="cuda")
cv = ops.pack_conv(torch.randn(N, Cin, k, k) * 0.05, torch.randn(N), "cuda")
xp = ops.split_planes(x)
out = ops.conv2d(xp, cv, bn=bn, stages=stages)
for _ in range(reps):
    ops.conv2d(xp, cv, bn=bn, stages=stages, out=out)
torch.cuda.synchronize()
