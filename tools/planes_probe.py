"""Phase timing of the planes GEMM with in-kernel s_memtime stamps (diagnostic build tools/libplanes_probe.so =
ffsr_gemm_planes.hip compiled with -DFFSR_PLANES_PROBE).  usage: python tools/planes_probe.py M K N [k] [bm bn stages]"""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("image-super-resolution_amd.ops")
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libplanes_probe.so"))
M, K, N = (int(v) for v in sys.argv[1:4])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 1
bm, bn, stages = (int(v) for v in sys.argv[5:8]) if len(sys.argv) > 7 else ops.planes_tile(M, N, K * k * k)
H, W = (352, M // 352) if k > 1 else (1, M)
x = torch.randn(1, H, W, K, device="cuda")
cv = ops.pack_conv(torch.randn(N, K, k, k) * 0.05, torch.randn(N), "cuda")
xp = ops.split_planes(x)
out = ops.new_map(1, H, W, N, "cuda")
tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
stamps = torch.zeros(tiles, 8, dtype=torch.int64, device="cuda")
lib.ffsr_planes_probe_set(ctypes.c_void_p(stamps.data_ptr()))
flush = torch.empty(1 << 28, device="cuda")
vp = ctypes.c_void_p
f = ctypes.c_float
for it in range(3):
    flush.fill_(float(it))            # 1 GiB of writes: nothing of the operands is left in L2 / Infinity Cache
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.ffsr_conv2d_planes(vp(xp.hi.data_ptr()), vp(xp.lo.data_ptr()), xp.Cp, vp(cv.phi.data_ptr()), vp(cv.plo.data_ptr()),
                                cv.phi.shape[0], vp(ops.zero_page("cuda").data_ptr()), vp(cv.bias.data_ptr()), vp(out.data_ptr()),
                                None, None, None, None, None, 0, 1, H, W, N, ops.ld(out), 0, k, k, 1, k // 2, k // 2, 0, f(0.0),
                                f(1.0), f(1.0), bm, bn, stages, None)
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0, rc
s = stamps.cpu().double()
names = ["address set-up", "first tile landed", "1st K step + 2nd tile landed", "rest of main loop", "epilogue (stores issued)"]
print(f"M={M} K={K * k * k} N={N} tile {bm}x{bn}s{stages} tiles={tiles}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us (cold caches)")
d = s[:, 1:6] - s[:, 0:5]
for i, n in enumerate(names):
    print(f"  {n:30s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f}  (shader cycles)")
tot = s[:, 5] - s[:, 0]
span = (s[:, 5].max() - s[:, 0].min()).item()
print(f"  total per WG median {tot.median().item():.0f}; kernel span {span:.0f} cycles = {span / e0.elapsed_time(e1) / 1e3:.0f} cycles/us")
st = (s[:, 0] - s[:, 0].min()).sort().values
print("  WG start quantiles (cycles):", [int(st[int(q * (len(st) - 1))].item()) for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0)])
