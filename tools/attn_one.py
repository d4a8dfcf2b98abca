"""Run the DRCT window-attention kernel alone (for rocprofv3): python tools/attn_one.py C heads shift reps"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
C, heads, shift, reps = (int(v) for v in sys.argv[1:5])
variant = int(sys.argv[5]) if len(sys.argv) > 5 else 0
B, H, W = 1, 352, 512
qkv = torch.randn(B * H * W, 3 * C, device="cuda")
bias = torch.randn(31 * 31, heads, device="cuda")
out = ops.window_attn(qkv, bias, B, H, W, C, heads, 16, shift, (C // heads) ** -0.5, variant=variant)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.window_attn(qkv, bias, B, H, W, C, heads, 16, shift, (C // heads) ** -0.5, out=out, variant=variant)
e1.record()
torch.cuda.synchronize()
hd = C // heads
flops = 2.0 * 2 * 256 * 256 * hd * heads * (H // 16) * (W // 16)
print(f"v{variant} C={C} heads={heads} hd={hd} shift={shift}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us  {flops / (e0.elapsed_time(e1) / reps) / 1e9:.1f} TFLOP/s (algorithmic)")
