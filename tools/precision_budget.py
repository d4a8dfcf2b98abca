"""what each arithmetic mode of the GEMM / conv kernels costs in error, measured instead of assumed.
Full-depth experts + fusion on one 64x64 LR tile against the CPU oracle (max-abs, PSNR between the two paths), plus
NAFNet alone at 256x256 (BASELINE config 2):
  f32     exact v_mfma_f32_32x32x2_f32 (fmaf chains)
  bf16x3  hi*hi + hi*lo + lo*hi on the bf16 MFMA (default)
  bf16x2  weights rounded to bf16 (lo plane zero): a_hi*w_hi + a_lo*w_hi  -- same kernels, FFSR_WEIGHT_LO=0
usage (GPU box): python tools/precision_budget.py"""
import importlib
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))


def main():
    from ffsr_oracle import pipeline, nafnet as onaf
    from ffsr_oracle.scan_c import selective_scan_c
    from test_gpu_models import lr_image
    W = importlib.import_module("image-super-resolution_amd.weights")
    E = importlib.import_module("image-super-resolution_amd.engine")
    N = importlib.import_module("image-super-resolution_amd.nafnet")
    ops = importlib.import_module("image-super-resolution_amd.ops")
    weights = W.random_weights(seed=50)
    lr = lr_image(22, 1, 64, 64)
    lr2 = lr_image(21, 1, 256, 256)
    with torch.no_grad():
        want = pipeline.process_image(weights, lr, scan_fn=selective_scan_c)
        want2, _ = onaf.nafnet_sr(weights["nafnet"], lr2)
    for name, mode, wlo in (("f32", "f32", True), ("bf16x3", "bf16x3", True), ("bf16x2 (weights rounded to bf16)", "bf16x3", False)):
        ops.set_gemm_mode(mode)
        ops.set_weight_lo(wlo)
        eng = E.Engine(weights, "cuda")
        got = E.map_to_nchw(eng.process(E.nchw_to_map(lr, "cuda")))
        e = (got - want).abs().max().item()
        psnr = 10 * math.log10(1.0 / max(((got - want) ** 2).mean().item(), 1e-30))
        sr2, _ = eng.nafnet(E.nchw_to_map(lr2, "cuda"))
        e2 = (E.map_to_nchw(sr2) - want2).abs().max().item()
        print(f"{name:36s} full path 64x64: max|hip-oracle| {e:.2e}, PSNR(hip, oracle) {psnr:6.1f} dB;  NAFNet 256x256: max abs {e2:.2e}",
              flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
