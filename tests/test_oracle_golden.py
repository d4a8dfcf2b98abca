"""CPU: the oracle restatement reproduces the golden vectors captured from the reference
(oracle/make_golden.py ran the reference's own modules from /root/reference).  Tolerances are fp32
round-off only (the restatement uses the same torch ops)."""
import torch
from conftest import load_golden
from ffsr_oracle import drct, grl, nafnet, mambair, fusion, pipeline

TOL = 5e-6


def _close(a, b, tol=TOL):
    assert a.shape == b.shape
    assert (a - b).abs().max().item() <= tol


def test_nafnet_small():
    g = load_golden("nafnet_small.pt")
    sr, feat = nafnet.nafnet_sr(g["sd"], g["lr"], **g["cfg"])
    _close(sr, g["sr"]); _close(feat, g["feat"])


def test_drct_small():
    g = load_golden("drct_small.pt")
    sr, feat = drct.drct_forward(g["sd"], g["lr"])
    _close(sr, g["sr"]); _close(feat, g["feat"])


def test_grl_small():
    g = load_golden("grl_small.pt")
    sr, feat = grl.grl_forward(g["sd"], g["lr"])
    _close(sr, g["sr"]); _close(feat, g["feat"])


def test_mambair_small():
    # the scan inside is oracle/ffsr_oracle/scan.py on BOTH sides (mamba-ssm is absent): this pins
    # everything around the scan; the scan arithmetic itself is "parity unpinned".
    g = load_golden("mambair_small.pt")
    sr, feat = mambair.mambair_forward(g["sd"], g["lr"])
    _close(sr, g["sr"]); _close(feat, g["feat"], 2e-5)


def test_fusion_full_both_sizes():
    g = load_golden("fusion_full.pt")
    for tag, c in g["cases"].items():
        imgs = {k: v.float() for k, v in c["imgs"].items()}
        feats = {k: v.float() for k, v in c["feats"].items()}
        _close(fusion.fusion_forward(g["sd"], c["lr"], imgs, feats), c["out"])
        for a, b in zip(fusion.frequency_bands(g["sd"], c["lr"]), c["bands"]):
            _close(a, b)


def test_fusion_with_improvements_switched_off():
    """model.fusion.improvements of configs/train_config.yaml (io.py:186-193): each improvement off alone, and all off --
    the reference's own outputs for the network it builds then (oracle/make_golden.py golden_flags)"""
    g = load_golden("fusion_flags.pt")
    sd = dict(load_golden("fusion_full.pt")["sd"])
    sd.update(g["simple"])
    imgs = {k: v.float() for k, v in g["imgs"].items()}
    feats = {k: v.float() for k, v in g["feats"].items()}
    assert len(g["variants"]) == 7
    for v in g["variants"]:
        have = {k: sd[k] for k in v["keys"]}                    # only what the variant's network owns
        _close(fusion.fusion_forward(have, g["lr"], imgs, feats, flags=v["flags"]), v["out"])


def test_host_logic_40x56():
    h = load_golden("host_40x56.pt")
    w = {n: load_golden(f"{f}_small.pt")["sd"] for n, f in
         (("drct", "drct"), ("grl", "grl"), ("nafnet", "nafnet"), ("mamba", "mambair"))}
    lr = pipeline.uint2tensor4(h["img"].numpy())
    lp, (hh, ww) = pipeline.pad16(lr)
    assert (hh, ww) == (40, 56) and lp.shape[-2:] == (48, 64)
    _close(lp, h["lr_padded"], 0.0)
    imgs, feats, lr_in = pipeline.run_experts(w, lr, naf_cfg=load_golden("nafnet_small.pt")["cfg"])
    for n, key in (("drct", "drct"), ("grl", "grl"), ("nafnet", "naf"), ("mamba", "mamba")):
        _close(imgs[n], h[f"{key}_sr"].float(), 1e-3)      # fixture stored as fp16
        _close(feats[n], h[f"{key}_feat"].float(), 4e-3)


def test_uint8_rounding_is_half_to_even():
    t = torch.tensor([[[[0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255, 1.2, -0.3]]]]).repeat(1, 3, 2, 1)
    u = pipeline.tensor2uint(t)
    assert u.shape == (2, 6, 3) and u[1, :, 2].tolist() == [0, 2, 2, 254, 255, 0]


def test_scan_matches_naive_loop():
    from ffsr_oracle.scan import selective_scan_ref
    g = torch.Generator().manual_seed(0)
    B, Dm, N, L, G = 1, 8, 4, 37, 2
    u, dt = torch.randn(B, Dm, L, generator=g), torch.randn(B, Dm, L, generator=g)
    A = -torch.rand(Dm, N, generator=g) - 0.1
    Bm, Cm = torch.randn(B, G, N, L, generator=g), torch.randn(B, G, N, L, generator=g)
    D, bias = torch.randn(Dm, generator=g), torch.randn(Dm, generator=g)
    y = selective_scan_ref(u, dt, A, Bm, Cm, D, delta_bias=bias, delta_softplus=True)
    ref = torch.zeros_like(y)
    for d in range(Dm):
        h = torch.zeros(N, dtype=torch.float64)
        for t in range(L):
            x = float(dt[0, d, t] + bias[d])
            delta = x if x > 20 else float(torch.log1p(torch.exp(torch.tensor(x, dtype=torch.float64))))
            h = torch.exp(delta * A[d].double()) * h + delta * Bm[0, d // (Dm // G), :, t].double() * float(u[0, d, t])
            ref[0, d, t] = float((h * Cm[0, d // (Dm // G), :, t].double()).sum()) + float(D[d] * u[0, d, t])
    assert (y - ref).abs().max() < 1e-4


def test_c_scan_matches_torch_scan():
    from ffsr_oracle.scan import selective_scan_ref
    from ffsr_oracle.scan_c import selective_scan_c
    g = torch.Generator().manual_seed(1)
    B, Dm, N, L, G = 2, 24, 16, 50, 4
    u, dt = torch.randn(B, Dm, L, generator=g), torch.randn(B, Dm, L, generator=g)
    A = -torch.rand(Dm, N, generator=g) - 0.1
    Bm, Cm = torch.randn(B, G, N, L, generator=g), torch.randn(B, G, N, L, generator=g)
    D, bias = torch.randn(Dm, generator=g), torch.randn(Dm, generator=g)
    a = selective_scan_c(u, dt, A, Bm, Cm, D, delta_bias=bias, delta_softplus=True)
    b = selective_scan_ref(u, dt, A, Bm, Cm, D, delta_bias=bias, delta_softplus=True)
    assert (a - b).abs().max() < 1e-4


def test_metrics_oracle_matches_reference_values():
    """tests/golden/metrics.pt holds calculate_psnr / calculate_ssim of the imported reference
    (src/utils/metrics.py:76-186) on two seeded pairs, crop 0 / 4, RGB / BT.601-Y."""
    from ffsr_oracle import metrics as om
    g = load_golden("metrics.pt")
    assert len(g["cases"]) == 8
    for c in g["cases"]:
        a, b = g["pairs"][c["pair"]]
        assert abs(om.psnr(a, b, c["crop_border"], c["test_y_channel"]) - c["psnr"]) < 1e-4, c
        assert abs(om.ssim(a, b, c["crop_border"], c["test_y_channel"]) - c["ssim"]) < 1e-6, c


def test_u8_metrics_oracle_known_answers():
    """the evaluation script's uint8 variant (utils/utils_image.py:148-189; parity unpinned: cv2 / skimage are absent):
    OpenCV's fixed-point luma on values with known results, PSNR of a constant offset, SSIM of identical images"""
    import numpy as np
    from ffsr_oracle import metrics as om
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]]], dtype=np.uint8)
    assert om.rgb2y_opencv_u8(px)[0].tolist() == [255, 0, 76, 150, 29, 128]     # 0.299 / 0.587 / 0.114, rounded
    rng = np.random.RandomState(0)
    a = rng.randint(0, 250, (40, 52, 3)).astype(np.uint8)
    p, s = om.psnr_ssim_u8(a, a.copy())
    assert p == float("inf") and abs(s - 1.0) < 1e-12
    p, s = om.psnr_ssim_u8(a, a + 5, crop_border=0, test_y_channel=False)          # mse = 25 exactly
    assert abs(p - 20 * np.log10(255.0 / 5.0)) < 1e-12 and 0.9 < s < 1.0
    p4, _ = om.psnr_ssim_u8(a, a + 5, crop_border=4, test_y_channel=True)          # Y of a uniform +5 shift is +5 (+-1 rounding)
    assert abs(p4 - 20 * np.log10(255.0 / 5.0)) < 0.5


def test_fusion_train_mode_oracle_matches_reference_forward_and_gradients():
    """tests/golden/fusion_train.pt holds the reference's model.train() forward (dropout 0), the L1 loss and
    loss.backward()'s gradient of every parameter (oracle/make_golden.py golden_train).  The oracle's train mode,
    differentiated by torch autograd, must reproduce them: this is what pins the checker of the HIP backward pass."""
    import torch.nn.functional as F
    g, sd = load_golden("fusion_train.pt"), load_golden("fusion_full.pt")["sd"]
    sdo = {k: (v.clone().requires_grad_(True) if k in g["grads"] else v.clone()) for k, v in sd.items()}
    imgs, feats = {k: v.float() for k, v in g["imgs"].items()}, {k: v.float() for k, v in g["feats"].items()}
    sr = fusion.fusion_forward(sdo, g["lr"], imgs, feats, train=True)
    _close(sr.detach(), g["sr"], 2e-5)
    loss = F.l1_loss(sr.clamp(0, 1), g["hr"])
    assert abs(loss.item() - g["loss"].item()) < 1e-6
    loss.backward()
    assert len(g["grads"]) == 198 and sum(v.numel() for v in g["grads"].values()) == 1433217
    for k, want in g["grads"].items():
        err = (sdo[k].grad - want).abs().max().item()
        assert err <= 1e-3 * want.abs().max().item() + 1e-12, (k, err)
    for k, want in g["stats"].items():
        _close(sdo[k].detach(), want, 1e-5)
    # eval mode is untouched by the train flag's plumbing: the clamps are back
    with torch.no_grad():
        assert fusion.fusion_forward(sd, g["lr"], imgs, feats).max().item() <= 1.0
