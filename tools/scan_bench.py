"""Micro-benchmark of the 4-direction selective scan at the 340x510 (padded 352x512) size for several chunk lengths."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("FFSR_AB_LIB"):          # A/B against another build of the library (tools/ab/*.so)
    importlib.import_module("image-super-resolution_amd.hip").LIB_PATH = os.path.abspath(os.environ["FFSR_AB_LIB"])
ops = importlib.import_module("image-super-resolution_amd.ops")
dev = "cuda"
B, H, W, Dm, R = 1, 352, 512, 360, 12
L = H * W
u = torch.randn(B * L, Dm, device=dev)
xdbl = torch.randn(B * L, 4 * (R + 32), device=dev) * 0.5
dtw = torch.randn(4, Dm, R, device=dev) * 0.1
dtb = torch.randn(4, Dm, device=dev) * 0.1
A = -torch.exp(torch.randn(4 * Dm, 16, device=dev) * 0.3)
Dv = torch.randn(4 * Dm, device=dev)
ref = None
for chunk in [int(v) for v in (sys.argv[1:] or ["256"])]:
    y = ops.selective_scan4(u, xdbl, dtw, dtb, A, Dv, B, H, W, Dm, R, chunk=chunk)
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    err = (y - ref).abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.selective_scan4(u, xdbl, dtw, dtb, A, Dv, B, H, W, Dm, R, chunk=chunk)
    e1.record()
    torch.cuda.synchronize()
    n = (L + chunk - 1) // chunk
    print(f"chunk {chunk:5d} nchunk {n:5d} waves {24 * n:6d}: {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us (3 kernels)  max diff vs first {err:.2e}", flush=True)
