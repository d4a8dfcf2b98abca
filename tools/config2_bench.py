"""BASELINE config 2: NAFNet-width64 + pixel-shuffle x4 alone on a 256x256 LR image (1024x1024 out), the three arithmetic
modes (bf16x3 = default, f32 = exact, bf16 = plain bf16 operands: the precision BASELINE names): time per image (events, 20 reps
after 3 warm-ups), output MP/s, max-abs error and PSNR vs the CPU oracle.
usage (GPU box): python tools/config2_bench.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))


def main():
    from ffsr_oracle import nafnet as onaf
    from test_gpu_models import lr_image
    W = importlib.import_module("image-super-resolution_amd.weights")
    E = importlib.import_module("image-super-resolution_amd.engine")
    N = importlib.import_module("image-super-resolution_amd.nafnet")
    ops = importlib.import_module("image-super-resolution_amd.ops")
    sd = W.nafnet_state_dict(seed=81)
    lr = lr_image(21, 1, 256, 256)
    with torch.no_grad():
        want, _ = onaf.nafnet_sr(sd, lr)
    for mode in ("bf16x3", "f32", "bf16"):
        ops.set_gemm_mode(mode)
        net = N.NAFNetSR(sd, "cuda")
        x = E.nchw_to_map(lr, "cuda")
        for _ in range(3):
            sr, _ = net(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            sr, _ = net(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        got = E.map_to_nchw(sr)
        err = (got - want).abs().max().item()
        psnr = -10.0 * torch.log10(((got - want) ** 2).mean()).item()
        print(f"config 2 [{mode}]: {ms:.2f} ms per 256x256 LR image = {1.048576 / ms * 1e3:.1f} output-MP/s "
              f"({2.02 / ms * 1e3:.0f} TFLOP/s algorithmic), max|hip - oracle| = {err:.2e}, PSNR(hip, oracle) = {psnr:.1f} dB", flush=True)
    ops.set_gemm_mode(os.environ.get("FFSR_GEMM_MODE", "bf16x3"))


if __name__ == "__main__":
    main()
