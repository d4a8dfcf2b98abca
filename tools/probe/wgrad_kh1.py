"""Probe: the bf16 weight-gradient kernel with one kernel row (1x3) against three (3x3) on 32 x 256 x 256 pixels, 128 -> 128.
One row = a third of the workgroups and no rows shared between workgroups through L2: 703 vs 2130 us = exactly a third, so the
3x3 layer does not lose time to cross-workgroup row sharing; the per-workgroup pipeline is the limit."""
import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
hip = importlib.import_module("image-super-resolution_amd.hip")
dev = "cuda"
B, H, W, Cin, N = 32, 256, 256, 128, 128
x = torch.randn(B, H, W, Cin, device=dev); dy = torch.randn(B, H, W, N, device=dev)
part = torch.empty(1 << 26, device=dev)
for KH, ph in ((3, 1), (1, 0)):
    def run():
        dw = torch.zeros(N, Cin, KH, 3, device=dev)
        hip.call("ffsr_conv_wgrad_bf16x3", x.data_ptr(), Cin, dy.data_ptr(), N, dw.data_ptr(), None, part.data_ptr(), part.numel(), B, H, W, Cin, N, KH, 3, ph, 1, torch.cuda.current_stream().cuda_stream)
    run(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): run()
    torch.cuda.synchronize(); print("KH", KH, (time.time() - t0) / 5 * 1e6, "us", flush=True)
