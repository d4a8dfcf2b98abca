"""Parity at the geometry of BASELINE.json's configs 2 and 3 (the non-bench configs are parity-test cases), and the edge
sizes of the per-image path:

  configs[1]  NAFNet-width64 (full depth: enc 2,2,4,8 / mid 12 / dec 2,2,2,2) alone on a 256x256 LR image -> 1024x1024
  configs[2]  full 4-expert + 7-phase fusion path on a batch of sixteen 64x64 LR tiles, PSNR parity vs the CPU path

(configs[0], fusion-only with bicubic stand-in experts, is test_gpu_models.test_fusion_golden_64_and_odd against the vectors
captured from the reference; configs[3] is the sharded bench itself plus tests/test_shard_gloo.py.)
The oracle is the torch-CPU restatement under oracle/ (checker only).  north_star tolerance: max abs <= 1e-3 in fp32.
"""
import importlib
import math
import os

import pytest
import torch

from test_gpu_models import DEV, TOL, err, lr_image, mod, run_expert

pytestmark = pytest.mark.gpu


_CFG2 = {}


def _config2_case():
    """the oracle pass (2 TFLOP on the host cores, ~20 s) is shared by the two arithmetic modes"""
    if not _CFG2:
        from ffsr_oracle import nafnet as onaf
        sd = mod("weights").nafnet_state_dict(seed=81)
        lr = lr_image(21, 1, 256, 256)
        with torch.no_grad():
            _CFG2["case"] = (sd, lr) + tuple(onaf.nafnet_sr(sd, lr))
    return _CFG2["case"]


@pytest.mark.parametrize("mode", ["bf16x3", "f32", "bf16"])
def test_config2_nafnet_alone_256x256(mode):
    """bf16x3 (default) and f32 meet north_star's 1e-3 max-abs.  bf16 = BASELINE config 2's named precision (plain bf16 operands,
    one MFMA per product, fp32 accumulate; the reference's GPU route runs under autocast): its max-abs is ~2e-2 (measured:
    profiles/r02_precision_budget.txt 2.3e-2 at 64x64, printed below), so it is a precision OPTION -- the bar here is image
    fidelity: PSNR(hip, oracle) measured 50.4 dB with these random weights (bar: >= 45 dB), i.e. the arithmetic noise is ~20 dB
    below the ~30 dB error of super-resolution itself, but NOT "identical to 3 s.f." -- which is why it is not the default."""
    ops = mod("ops")
    sd, lr, want_sr, want_feat = _config2_case()
    ops.set_gemm_mode(mode)
    try:
        assert ops.gemm_mode_name() == mode
        sr, feat = run_expert(mod("nafnet").NAFNetSR(sd, DEV), lr)
    finally:
        ops.set_gemm_mode(os.environ.get("FFSR_GEMM_MODE", "bf16x3"))
    assert tuple(sr.shape) == (1, 3, 1024, 1024) and tuple(feat.shape) == tuple(want_feat.shape)
    e_sr, e_feat = err(sr, want_sr), err(feat, want_feat)
    psnr = -10.0 * torch.log10(((sr - want_sr) ** 2).mean()).item()
    print(f"[{mode}] NAFNet 256x256: sr max abs err {e_sr:.2e}, feature max abs err {e_feat:.2e}, PSNR(hip, oracle) {psnr:.1f} dB")
    if mode == "bf16":
        assert psnr >= 45.0 and e_sr < 0.1, (psnr, e_sr)     # measured on MI355X: 50.4 dB, max-abs 1.7e-2 (8-bit significands)
    else:
        assert e_sr < TOL and e_feat < TOL * max(1.0, want_feat.abs().max().item()), (mode, e_sr, e_feat)


def test_config3_full_path_batch16_64x64():
    """B = 16 through Engine.process in one call.  Every tile must equal the same tile processed alone (the reference
    loop is batch 1, io.py:330-345); tiles 0 and 15 are also checked against the CPU oracle, with the PSNR between the
    two paths reported (identical images to 3 s.f. needs PSNR(hip, oracle) far above the ~30 dB of SR itself)."""
    from ffsr_oracle import pipeline
    from ffsr_oracle.scan_c import selective_scan_c
    W, E = mod("weights"), mod("engine")
    weights = W.random_weights(seed=50)
    eng = E.Engine(weights, DEV)
    lr = lr_image(22, 16, 64, 64)
    got = E.map_to_nchw(eng.process(E.nchw_to_map(lr, DEV)))
    assert tuple(got.shape) == (16, 3, 256, 256)
    for i in (0, 7, 15):
        one = E.map_to_nchw(eng.process(E.nchw_to_map(lr[i:i + 1], DEV)))
        assert err(got[i:i + 1], one) < 1e-5, i
    for i in (0, 15):
        want = pipeline.process_image(weights, lr[i:i + 1], scan_fn=selective_scan_c)
        e = err(got[i:i + 1], want)
        mse = ((got[i:i + 1] - want) ** 2).mean().item()
        psnr = 10 * math.log10(1.0 / max(mse, 1e-20))
        print(f"tile {i}: max abs err {e:.2e}, PSNR(hip, oracle) {psnr:.1f} dB")
        assert e < TOL and psnr > 80.0, (i, e, psnr)


# ------------------------------------------------------------------ edge sizes of the per-image path
@pytest.mark.parametrize("h,w", [(9, 9), (17, 33), (16, 48)])
def test_smallest_and_ragged_sizes_vs_oracle(h, w):
    """9x9 is the smallest image the reference accepts (reflect pad 7 < 9, one 16x16 window after pad16, io.py:71-83);
    17x33 pads by 15 on both axes; 16x48 needs no padding at all."""
    from ffsr_oracle import pipeline
    W, E = mod("weights"), mod("engine")
    weights = W.random_weights(seed=90, small=True)
    eng = E.Engine(weights, DEV)
    lr = lr_image(23, 1, h, w)
    want = pipeline.process_image(weights, lr, naf_cfg=dict(enc_blks=(1, 1, 1, 1), mid_blks=1, dec_blks=(1, 1, 1, 1)))
    got = E.map_to_nchw(eng.process(E.nchw_to_map(lr, DEV)))
    assert tuple(got.shape) == (1, 3, 4 * h, 4 * w)
    assert err(got, want) < TOL, (h, w, err(got, want))


def test_image_too_small_for_reflect_pad_raises():
    """torch's reflect pad refuses a pad >= the dimension (8 rows -> pad 8), so the reference raises from _pad16
    (io.py:71-78); the engine raises too instead of producing an image."""
    W, E, hip = mod("weights"), mod("engine"), mod("hip")
    eng = E.Engine(W.random_weights(seed=90, small=True), DEV)
    with pytest.raises(RuntimeError):
        torch.nn.functional.pad(torch.zeros(1, 3, 8, 40), (0, 8, 0, 8), mode="reflect")
    with pytest.raises(hip.FfsrError):
        eng.process(E.nchw_to_map(lr_image(24, 1, 8, 40), DEV))


def test_large_image_680x1020_and_the_size_guard():
    """four times BASELINE's pixels (680x1020 LR -> 2720x4080: NAFNet level-0 maps of 1.4 G elements, 2.8 GB plane
    operands): 64-bit addressing end to end.  The size-independent check (the oracle cannot run this size in minutes) -- the default arithmetic
    agrees with the exact mode to 1e-3 -- on reduced-depth experts of the real width (every kernel of the path runs).
    Beyond 2^32 bytes per plane operand the C ABI refuses (32-bit byte offsets) instead of wrapping."""
    W, E, ops, hip = mod("weights"), mod("engine"), mod("ops"), mod("hip")
    weights = W.random_weights(seed=91, small=True)
    lr = lr_image(25, 1, 680, 1020)
    outs = {}
    for mode in ("bf16x3", "f32"):
        ops.set_gemm_mode(mode)
        try:
            eng = E.Engine(weights, DEV)
            outs[mode] = eng.process(E.nchw_to_map(lr, DEV))[..., :3].float().cpu()
            del eng
            torch.cuda.empty_cache()
        finally:
            ops.set_gemm_mode(os.environ.get("FFSR_GEMM_MODE", "bf16x3"))
    a, b = outs["bf16x3"], outs["f32"]
    assert a.shape == (1, 2720, 4080, 3) and torch.isfinite(a).all() and a.min().item() >= 0.0 and a.max().item() <= 1.0
    diff = err(a, b)
    print(f"680x1020: max |fast - exact| = {diff:.3e}")
    assert diff < TOL, diff
    # a plane operand of 2^32 bytes or more is rejected before any launch (M = 2^24 rows x 128 channels x 2 B)
    z = ops.zero_page(DEV)
    cv = ops.pack_conv(torch.zeros(8, 128, 1, 1), None, DEV)
    tiny = torch.zeros(64, device=DEV)
    with pytest.raises(hip.FfsrError, match="invalid argument"):
        hip.call("ffsr_conv2d_planes", tiny.data_ptr(), tiny.data_ptr(), 128, cv.phi.data_ptr(), cv.plo.data_ptr(),
                 cv.phi.shape[0], z.data_ptr(), None, tiny.data_ptr(), None, None, None, None, None, 0, 1, 4096, 4096, 8, 8, 0,
                 1, 1, 1, 0, 0, 0, 0.0, 1.0, 1.0, 128, 64, 2, torch.cuda.current_stream().cuda_stream)
