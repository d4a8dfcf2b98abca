"""Time GRL's two attention kernels (cosine window attention, anchored stripe attention) at the 352x512 size.
usage (GPU box): python tools/grl_attn_bench.py     (tools/grl_attn_pmc.sh collects their counters over this program)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
dev = "cuda"
B, H, W, heads, hd = 1, 352, 512, 3, 30
C = 180
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * H * W, 3 * C, generator=g).to(dev)
anchor = torch.randn(B, H // 2, W // 2, C // 2, generator=g).to(dev)
bias = torch.randn(heads, 64, 64, generator=g).to(dev)
b1 = torch.randn(heads, 16, 64, generator=g).to(dev)      # [heads, anchor keys (4x4), window queries]
b2 = torch.randn(heads, 64, 16, generator=g).to(dev)
logit = (torch.rand(heads, generator=g) + 0.5).to(dev)
out = torch.empty(B * H * W, C, device=dev)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for shift in (0, 4):
    us = timeit(lambda: ops.grl_window_attn(qkv, 0, bias, logit, out, 0, B, H, W, heads, hd, shift))
    print(f"window kernel, shift {shift}: {us:7.1f} us", flush=True)
try:
    us = timeit(lambda: ops.grl_stripe_attn(qkv, 3 * C // 2, anchor, b1, b2, logit, logit, out, C // 2, B, H, W, heads, hd))
    print(f"stripe kernel: {us:7.1f} us", flush=True)
except Exception as e:          # bias table shapes are the model's business: report and go on
    print("stripe kernel not run:", e)
