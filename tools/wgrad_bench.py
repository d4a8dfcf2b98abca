"""Weight gradient of the refine stack's 128 -> 128 3x3 convolution at config 5's size (32 x 256 x 256 pixels):
ffsr_conv_wgrad_f32 (f32 MFMA) against ffsr_conv_wgrad_bf16x3 (transposing-read bf16 kernel), time and error vs fp64 on a slice."""
import importlib
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
hip = importlib.import_module("image-super-resolution_amd.hip")
dev = "cuda"
torch.manual_seed(0)
for (B, H, W, Cin, N) in [(32, 256, 256, 128, 128), (32, 64, 64, 128, 128), (8, 256, 256, 128, 128), (32, 256, 256, 76, 64), (32, 256, 256, 96, 32), (32, 256, 256, 128, 3),
                          (32, 256, 256, 3, 128), (32, 256, 256, 32, 3), (32, 256, 256, 16, 1)]:
    x = torch.randn(B, H, W, (Cin + 3) // 4 * 4, device=dev)
    dy = torch.randn(B, H, W, (N + 3) // 4 * 4, device=dev)
    part = torch.empty(1 << 26, device=dev)
    res = {}
    for name in ("ffsr_conv_wgrad_f32", "ffsr_conv_wgrad_bf16x3"):
        def run():
            dw = torch.zeros(N, Cin, 3, 3, device=dev)
            hip.call(name, x.data_ptr(), x.shape[3], dy.data_ptr(), dy.shape[3], dw.data_ptr(), None, part.data_ptr(), part.numel(), B, H, W, Cin, N,
                     3, 3, 1, 1, torch.cuda.current_stream().cuda_stream)
            return dw
        res[name] = run()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        us = (time.time() - t0) / 5 * 1e6
        print(f"{name} B{B} {H}x{W} {Cin}->{N}: {us:.0f} us = {2.0 * B * H * W * Cin * N * 9 / us / 1e6:.1f} TFLOP/s", flush=True)
    if Cin % 128 == 0 and N % 128 == 0:      # both operands as planes (LDS-DMA staged kernel; FFSR_WGRAD_DMA=0: register staged)
        ops = importlib.import_module("image-super-resolution_amd.ops")
        xp, dp = ops.split_planes(x), ops.split_planes(dy)
        def runp():
            dw = torch.zeros(N, Cin, 3, 3, device=dev)
            hip.call("ffsr_conv_wgrad_bf16x3_planes", xp.hi.data_ptr(), xp.lo.data_ptr(), xp.Cp, None, 0, dp.hi.data_ptr(), dp.lo.data_ptr(),
                     dp.Cp, dw.data_ptr(), None, part.data_ptr(), part.numel(), B, H, W, Cin, N, 3, 3, 1, 1, torch.cuda.current_stream().cuda_stream)
            return dw
        rp = runp()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            runp()
        torch.cuda.synchronize()
        us = (time.time() - t0) / 5 * 1e6
        print(f"planes x planes B{B} {H}x{W} {Cin}->{N}: {us:.0f} us = {2.0 * B * H * W * Cin * N * 9 / us / 1e6:.1f} TFLOP/s; "
              f"equal to fp32-input bf16x3: {torch.equal(rp, res['ffsr_conv_wgrad_bf16x3'])}", flush=True)
    a, b = res["ffsr_conv_wgrad_f32"], res["ffsr_conv_wgrad_bf16x3"]
    print("  max rel diff bf16x3 vs f32:", ((a - b).abs().max() / a.abs().max()).item(), flush=True)
