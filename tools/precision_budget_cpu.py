"""VERDICT r1 item 7, the 1-term ("plain bf16") point of the precision budget, estimated on the CPU oracle: every nn.Linear /
nn.Conv2d of the path evaluated with BOTH operands rounded to bf16 and fp32 accumulation (what a hi*hi-only MFMA mode of
the GEMM kernels would compute; attention contractions, norms, the scan and everything elementwise stay fp32), against the
unmodified fp32 oracle.  Same weights / input as tools/precision_budget.py.  usage: python tools/precision_budget_cpu.py"""
import importlib
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    from ffsr_oracle import pipeline, nafnet as onaf
    from ffsr_oracle.scan_c import selective_scan_c
    W = importlib.import_module("image-super-resolution_amd.weights")
    g = torch.Generator().manual_seed(22)
    x = torch.rand(1, 3, 64, 64, generator=g)
    x = F.avg_pool2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), 5, 1)
    lr = torch.floor((x - x.amin()) / (x.amax() - x.amin()) * 256).clamp(0, 255) / 255.0
    weights = W.random_weights(seed=50)
    with torch.no_grad():
        want = pipeline.process_image(weights, lr, scan_fn=selective_scan_c)
        want_naf, _ = onaf.nafnet_sr(weights["nafnet"], lr)
        conv2d, linear = F.conv2d, F.linear
        rb = lambda t: t.to(torch.bfloat16).float()
        for name, round_x in (("bf16x2-like (weights rounded to bf16)", False), ("bf16 1-term (weights AND activations rounded)", True)):
            F.conv2d = lambda x_, w_, *a, **k: conv2d(rb(x_) if round_x else x_, rb(w_), *a, **k)
            F.linear = lambda x_, w_, *a, **k: linear(rb(x_) if round_x else x_, rb(w_), *a, **k)
            try:
                got = pipeline.process_image(weights, lr, scan_fn=selective_scan_c)
                got_naf, _ = onaf.nafnet_sr(weights["nafnet"], lr)
            finally:
                F.conv2d, F.linear = conv2d, linear
            e = (got - want).abs().max().item()
            psnr = 10 * math.log10(1.0 / max(((got - want) ** 2).mean().item(), 1e-30))
            print(f"{name:48s} full path 64x64: max abs {e:.2e}, PSNR vs fp32 {psnr:5.1f} dB;  NAFNet 64x64: max abs "
                  f"{(got_naf - want_naf).abs().max().item():.2e}", flush=True)


if __name__ == "__main__":
    main()
