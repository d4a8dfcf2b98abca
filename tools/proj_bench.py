"""ffsr_tok_proj_f32 at MambaIR's shape (M = 352 x 512 tokens, 4 x 360 -> 180) for both workgroup sizes; compares with the
separate launches it replaces.   python tools/proj_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
dev = "cuda"
M, K, N = 352 * 512, 360, 180
g = torch.Generator().manual_seed(0)
y4 = torch.randn(4, M, K, generator=g).to(dev)
xz = torch.randn(M, 2 * K, generator=g).to(dev)
x = torch.randn(M, N, generator=g).to(dev)
w0 = torch.randn(N, K, generator=g) / K ** 0.5
pg, pb, g2, be2, skip = (torch.rand(n, generator=g).to(dev) + 0.5 for n in (K, K, N, N, N))
tg = ops.pack_tok_gemm(w0, None, dev, check=False)
cv = ops.pack_conv(w0.reshape(N, K, 1, 1), None, dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def fused():
    return ops.tok_proj(y4[0], tg, xdirs=4, xstride=y4.stride(0), z=xz[:, K:], pro_ln=(pg, pb), res=x, rvec=skip, post_ln=(g2, be2),
                        out_pre_ln=True, out_planes=True)


def plain():
    return ops.tok_proj(y4[0], tg, res=x, rvec=skip, post_ln=(g2, be2), out_pre_ln=True, out_planes=True)


def separate():
    gt = ops.mamba_norm_gate(y4, xz[:, K:], pg, pb, out_planes=True, want_f32=False)
    y = ops.linear(gt, cv, res=x, rvec=skip)
    return ops.layernorm(y, g2, be2, out_planes=True, want_f32=False)


if len(sys.argv) > 2:          # one case only (for rocprofv3 --pmc): python tools/proj_bench.py <waves> fused|plain
    ops.TOK_WAVES = int(sys.argv[1])
    print(f"waves {sys.argv[1]} {sys.argv[2]}: {timeit(fused if sys.argv[2] == 'fused' else plain, 10):.1f} us")
    sys.exit(0)
for w in (8, 4):
    ops.TOK_WAVES = w
    print(f"waves {w}: fused {timeit(fused):.1f} us   projection only (no prologue) {timeit(plain):.1f} us", flush=True)
print(f"separate launches (norm_gate + planes GEMM + LayerNorm): {timeit(separate):.1f} us")
