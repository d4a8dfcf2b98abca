"""Cold-cache micro-benchmark of the streaming (HBM-bound) kernels: achieved GB/s vs the bytes they must move."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
dev = "cuda"
flush = torch.empty(1 << 28, device=dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for i in range(reps):
        flush.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps


def report(name, ms, nbytes):
    print(f"{name:58s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:7.0f} GB/s", flush=True)


M = 352 * 512
for C, ld in ((180, 180), (180, 308), (308, 308), (360, 360)):
    x = torch.randn(M, ld, device=dev)[:, :C]
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    out = torch.empty(M, C, device=dev)
    report(f"layernorm [{M},{C}] ld {ld} -> f32", timeit(lambda: ops.layernorm(x, g, b, out=out)), M * C * 8)
    pl = ops.Planes(1, 1, M, C, dev)
    report(f"layernorm [{M},{C}] ld {ld} -> planes", timeit(lambda: ops.layernorm(x, g, b, out_planes=pl, want_f32=False)), M * (C * 4 + pl.Cp * 4))
x = torch.randn(1, 1408, 2048, 64, device=dev)
g, b = torch.randn(64, device=dev), torch.randn(64, device=dev)
out = torch.empty_like(x)
report("layernorm NAFNet level 0 [2.88M,64]", timeit(lambda: ops.layernorm(x, g, b, out=out)), x.numel() * 8)
xm = torch.randn(1, 352, 512, 360, device=dev)
dw = ops.pack_dwconv(torch.randn(360, 1, 3, 3), torch.randn(360), dev)
o = torch.empty_like(xm)
report("dwconv 3x3 + SiLU [352x512, 360]", timeit(lambda: ops.dwconv2d(xm, dw, act=ops.ACT_SILU, out=o)), xm.numel() * 8)
for (H, W, C) in ((1408, 2048, 64), (704, 1024, 128), (352, 512, 256)):
    t = torch.randn(1, H, W, 2 * C, device=dev)
    dwg = ops.pack_dwconv(torch.randn(2 * C, 1, 3, 3), torch.randn(2 * C), dev)
    report(f"dw3x3 + SimpleGate + pool [{H}x{W}, {2 * C}->{C}]", timeit(lambda: ops.dw3x3_gate_pool(t, dwg)), t.numel() * 4 * 1.5)
y4 = torch.randn(4, M, 360, device=dev)
z = torch.randn(M, 720, device=dev)[:, 360:]
g, b = torch.randn(360, device=dev), torch.randn(360, device=dev)
pl = ops.Planes(1, 1, M, 360, dev)
report("mamba_norm_gate -> planes", timeit(lambda: ops.mamba_norm_gate(y4, z, g, b, out_planes=pl, want_f32=False)), M * 360 * 4 * 5 + M * 384 * 4)
a, bb = torch.randn(M, 180, device=dev), torch.randn(M, 180, device=dev)
o2 = torch.empty_like(a)
report("scale_add [M,180]", timeit(lambda: ops.scale_add(a, bb, out=o2)), M * 180 * 12)
report("colmean [352x512,180]", timeit(lambda: ops.colmean(a.reshape(1, 352, 512, 180))), M * 180 * 4)
