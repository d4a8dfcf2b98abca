"""Run one conv shape N times (for rocprofv3 --pmc): python tools/gemm_one.py H W Cin N k hint reps"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
H, W, Cin, N, k, hint, reps = (int(v) for v in sys.argv[1:8])
x = torch.randn(1, H, W, Cin, device="cuda")
cv = ops.pack_conv(torch.randn(N, Cin, k, k) * 0.05, torch.randn(N), "cuda")
out = ops.conv2d(x, cv, tile_hint=hint)
for _ in range(reps):
    ops.conv2d(x, cv, tile_hint=hint, out=out)
torch.cuda.synchronize()
