"""Experiment: LN -> fc1 (GELU) -> fc2 (+res) over the whole token range vs in token chunks (intermediates that fit the
256 MiB Infinity Cache between producer and consumer).  usage: mlp_chunk_bench.py [C] [ratio]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("image-super-resolution_amd.ops")
dev = "cuda"
C = int(sys.argv[1]) if len(sys.argv) > 1 else 180
ratio = int(sys.argv[2]) if len(sys.argv) > 2 else 2
M = 352 * 512
x = torch.randn(M, C, device=dev)
g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
f1 = ops.pack_conv(torch.randn(ratio * C, C) * 0.05, torch.randn(ratio * C), dev)
f2 = ops.pack_conv(torch.randn(C, ratio * C) * 0.05, torch.randn(C), dev)
flush = torch.empty(1 << 28, device=dev)


def mlp(nch):
    out = torch.empty(M, C, device=dev)
    per = (M // nch + 127) // 128 * 128
    for c0 in range(0, M, per):
        xs = x[c0:c0 + per]
        n = ops.layernorm(xs, g, b, out_planes=True, want_f32=False)
        h = ops.linear(n, f1, act=ops.ACT_GELU, out_planes=True, want_f32=False)
        ops.linear(h, f2, res=xs, out=out[c0:c0 + per])
    return out


ref = mlp(1)
for nch in (1, 2, 3, 4, 8, 1, 2, 4):
    assert torch.equal(mlp(nch), ref)
    tot = 0.0
    for i in range(5):
        flush.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); mlp(nch); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    print(f"C={C} ratio={ratio} chunks={nch}: {tot / 5 * 1e3:8.1f} us per MLP (LN + fc1 + fc2)", flush=True)
