"""Time GRL's window attention kernel at the 352x512 size."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
dev = "cuda"
B, H, W, heads, hd = 1, 352, 512, 3, 30
C = 180
qkv = torch.randn(B * H * W, 3 * C, device=dev)
bias = torch.randn(heads, 64, 64, device=dev)
logit = torch.rand(heads, device=dev) + 0.5
out = torch.empty(B * H * W, C, device=dev)
for shift in (0, 4):
    for exact in (True, True):
        ops.grl_window_attn(qkv, 0, bias, logit, out, 0, B, H, W, heads, hd, shift)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.grl_window_attn(qkv, 0, bias, logit, out, 0, B, H, W, heads, hd, shift)
        e1.record()
        torch.cuda.synchronize()
        print(f"shift {shift} window kernel: {e0.elapsed_time(e1) / 10 * 1e3:7.1f} us", flush=True)
