"""Phase timing of the bf16x3 GEMM kernel with in-kernel s_memtime stamps (diagnostic)."""
import ctypes, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("image-super-resolution_amd.ops")
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libgemm_probe.so"))
M, K, N = (int(v) for v in sys.argv[1:4])
x = torch.randn(M, K, device="cuda")
cv = ops.pack_conv(torch.randn(N, K) * 0.05, None, "cuda")
out = torch.empty(M, N, device="cuda")
tiles = ((M + 127) // 128) * ((N + 63) // 64)
stamps = torch.zeros(tiles, 8, dtype=torch.int64, device="cuda")
z = ops.zero_page("cuda")
for _ in range(3):
    rc = lib.probe_gemm(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(cv.whi.data_ptr()), ctypes.c_void_p(cv.wlo.data_ptr()),
                        cv.whi.shape[1], ctypes.c_void_p(z.data_ptr()), ctypes.c_void_p(out.data_ptr()), M, K, N,
                        ctypes.c_void_p(stamps.data_ptr()), None)
    torch.cuda.synchronize()
assert rc == 0
s = stamps.cpu().double()
d = s[:, 1:7] - s[:, 0:6]
names = ["prologue addr", "first tile load+store", "compute kt=0", "store kt=1 (exposed load wait)", "rest of main loop", "epilogue"]
print(f"M={M} K={K} N={N} tiles={tiles}  (s_memtime ticks = 100 MHz refclk? or shader cycles; relative shares matter)")
for i, n in enumerate(names):
    print(f"  {n:34s} median {d[:, i].median().item():9.0f}  mean {d[:, i].mean().item():9.0f}")
tot = s[:, 6] - s[:, 0]
print(f"  total per WG median {tot.median().item():.0f}; kernel span {(s[:, 6].max() - s[:, 0].min()).item():.0f}")
