"""Generate tests/golden/*.pt by running the REFERENCE (imported from /root/reference through
oracle/ref_harness.py) on seeded inputs, and check the oracle restatement against it.

Run in the build container only:   python oracle/make_golden.py
Fixtures are data (weights as tensors, inputs, expected outputs) -- no reference source.
Reduced-size experts keep the fixtures small; the fusion net is the full 1.43 M-parameter model.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_harness import load_reference  # noqa: E402
from ffsr_oracle import drct, grl, nafnet, mambair, fusion, pipeline  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


def randomize_(module, seed):
    """reference-style init, then make every branch live (zero-init scalers, biases,
    BN running statistics randomised)."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for k, v in sd.items():
        if not v.is_floating_point():
            continue
        if k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif k.endswith("running_mean"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        elif k.endswith((".beta", ".gamma")):
            v.copy_(torch.randn(v.shape, generator=g) * 0.3)
        elif k.endswith("bias") and float(v.abs().max()) == 0.0:
            v.copy_(torch.randn(v.shape, generator=g) * 0.05)
        elif k.endswith("relative_position_bias_table"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.5)
        elif k.endswith(("skip_scale", "skip_scale2", "norm.weight", "norm1.weight", "norm2.weight",
                         "ln_1.weight", "ln_2.weight", "out_norm.weight", "norm_start.weight",
                         "norm_end.weight")) and v.dim() == 1:
            v.copy_(1.0 + 0.2 * torch.randn(v.shape, generator=g))
        elif k.endswith(("A_logs",)):
            v.copy_(v + 0.3 * torch.randn(v.shape, generator=g))
        elif k.endswith(("Ds",)):
            v.copy_(1.0 + 0.3 * torch.randn(v.shape, generator=g))
    for k, v in sd.items():          # fp16-exact weights -> fixtures store them as fp16 (half the bytes)
        if v.is_floating_point():
            v.copy_(v.half().float())
    module.load_state_dict(sd)
    return module


def half(sd):
    return {k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()}


def load_sd(path):
    return {k: (v.float() if v.is_floating_point() else v) for k, v in torch.load(path)["sd"].items()}


def lr_input(seed, b, h, w):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(b, 3, h, w, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"), 5, 1)
    x = (x - x.amin()) / (x.amax() - x.amin())
    return torch.floor(x * 256).clamp(0, 255) / 255.0


def floats(sd):
    return {k: v.detach().clone() for k, v in sd.items()}


def report(name, got, want, tol):
    err = (got - want).abs().max().item()
    print(f"  {name:34s} max|oracle-ref| = {err:.3e}   (|ref|max {want.abs().max().item():.3f})")
    assert err <= tol, f"{name}: oracle deviates from reference by {err}"
    return err


def hook_output(module):
    box = {}
    module.register_forward_hook(lambda m, i, o: box.__setitem__("out", o.detach()))
    return box


def hook_input(module):
    box = {}
    module.register_forward_hook(lambda m, i, o: box.__setitem__("out", i[0].detach()))
    return box


@torch.no_grad()
def main():
    os.makedirs(GOLDEN, exist_ok=True)
    ref = load_reference()
    torch.manual_seed(0)

    # ---------------- NAFNet (reduced width/depth, same topology) ----------------
    print("NAFNetSR")
    m = ref.create_nafnet_sr_model(upscale=4, width=8, middle_blk_num=2, enc_blk_nums=[1, 1, 2, 2],
                                   dec_blk_nums=[1, 1, 1, 1]).eval()
    randomize_(m, 11)
    box = hook_input(m.nafnet.ending)
    x = lr_input(1, 1, 24, 40)
    want = m(x)
    sd = floats(m.nafnet.state_dict())
    got, feat = nafnet.nafnet_sr(sd, x, enc_blks=(1, 1, 2, 2), mid_blks=2, dec_blks=(1, 1, 1, 1))
    report("sr", got, want, 2e-5)
    report("feat(ending input)", feat, box["out"], 2e-5)
    torch.save({"sd": half(sd), "lr": x, "sr": want, "feat": box["out"],
                "cfg": {"enc_blks": (1, 1, 2, 2), "mid_blks": 2, "dec_blks": (1, 1, 1, 1)}},
               os.path.join(GOLDEN, "nafnet_small.pt"))

    # ---------------- DRCT (embed 60, 2 RDG) ----------------
    print("DRCT")
    m = ref.DRCT(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30,
                 conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[6, 6], embed_dim=60,
                 num_heads=[6, 6], mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv").eval()
    randomize_(m, 12)
    box = hook_output(m.conv_after_body)
    x = lr_input(2, 1, 32, 48)
    want = m(x)
    sd = {k: v for k, v in floats(m.state_dict()).items()
          if not k.endswith(("attn_mask", "relative_position_index"))}
    got, feat = drct.drct_forward(sd, x)
    report("sr", got, want, 5e-5)
    report("feat(conv_after_body)", feat, box["out"], 5e-5)
    torch.save({"sd": half(sd), "lr": x, "sr": want, "feat": box["out"]}, os.path.join(GOLDEN, "drct_small.pt"))

    # ---------------- GRL (embed 60, depths [2, 3]) ----------------
    print("GRL")
    m = ref.create_grl_model(embed_dim=60, depths=[2, 3], num_heads_w=[3, 3], num_heads_s=[3, 3]).eval()
    randomize_(m, 13)
    box = hook_output(m.conv_after_body)
    x = lr_input(3, 1, 32, 48)
    want = m(x)
    sd = {k: v for k, v in floats(m.state_dict()).items()
          if not k.startswith(("table_", "index_", "mask_"))}
    got, feat = grl.grl_forward(sd, x)
    report("sr", got, want, 5e-5)
    report("feat(conv_after_body)", feat, box["out"], 5e-5)
    torch.save({"sd": half(sd), "lr": x, "sr": want, "feat": box["out"]}, os.path.join(GOLDEN, "grl_small.pt"))

    # ---------------- MambaIR (embed 48, depths (2,)); scan = our restatement (unpinned) ----------------
    print("MambaIR")
    m = ref.MambaIR(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30,
                    conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=(2,), embed_dim=48,
                    mlp_ratio=2.0, drop_path_rate=0.1, upsampler="pixelshuffle", resi_connection="1conv").eval()
    randomize_(m, 14)
    box = hook_output(m.conv_after_body)
    x = lr_input(4, 1, 16, 32)
    want = m(x)
    sd = floats(m.state_dict())
    got, feat = mambair.mambair_forward(sd, x)
    report("sr", got, want, 5e-5)
    report("feat(conv_after_body)", feat, box["out"], 5e-5)
    torch.save({"sd": half(sd), "lr": x, "sr": want, "feat": box["out"]}, os.path.join(GOLDEN, "mambair_small.pt"))

    # ---------------- fusion (full model), 64x64 and an odd size ----------------
    print("fusion")
    m = ref.CompleteEnhancedFusionSR(expert_ensemble=None).eval()
    randomize_(m, 15)
    sd = {k: v for k, v in floats(m.state_dict()).items() if not k.endswith("num_batches_tracked")}
    cases = {}
    for tag, (h, w), seed in (("t64", (64, 64), 5), ("odd", (35, 51), 6)):
        g = torch.Generator().manual_seed(seed)
        lr = lr_input(seed, 1, h, w)
        bic = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
        imgs = {n: (bic + 0.03 * torch.randn(bic.shape, generator=g)).clamp(0, 1) for n in fusion.EXPERTS}
        feats = {n: torch.randn(1, 64 if n == "nafnet" else 180, h, w, generator=g) for n in fusion.EXPERTS}
        want = m.forward_with_precomputed(lr, imgs, feats)
        got = fusion.fusion_forward(sd, lr, imgs, feats)
        report(f"fusion {tag} {h}x{w}", got, want, 2e-5)
        # phase-2 bands separately (non-pow2 FFT, reflect pads)
        _, raw = m.freq_decomp(lr, return_raw_bands=True)
        for i, (a, b) in enumerate(zip(fusion.frequency_bands(sd, lr), raw)):
            report(f"  band {i}", a, b, 1e-5)
        cases[tag] = {"lr": lr, "imgs": {k: v.half() for k, v in imgs.items()},
                      "feats": {k: v.half() for k, v in feats.items()}, "out": want,
                      "bands": [b.clone() for b in raw]}
        # stored inputs are fp16-exact so the fixture stays small: re-run reference on the rounded inputs
        imgs16 = {k: v.half().float() for k, v in imgs.items()}
        feats16 = {k: v.half().float() for k, v in feats.items()}
        cases[tag]["out"] = m.forward_with_precomputed(lr, imgs16, feats16)
        report(f"fusion {tag} (fp16-exact inputs)", fusion.fusion_forward(sd, lr, imgs16, feats16), cases[tag]["out"], 2e-5)
    torch.save({"sd": half(sd), "cases": cases}, os.path.join(GOLDEN, "fusion_full.pt"))

    # ---------------- host logic: pad16 / crops / NAFNet feature resample / uint8 ----------------
    print("process_image (small experts + full fusion), 40x56 uint8 input")
    import numpy as np
    rng = np.random.RandomState(7)
    img = (lr_input(7, 1, 40, 56)[0].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)
    lr = pipeline.uint2tensor4(img)
    # the reference-side sequence of io._process_image, using reference modules
    naf = ref.create_nafnet_sr_model(upscale=4, width=8, middle_blk_num=2, enc_blk_nums=[1, 1, 2, 2],
                                     dec_blk_nums=[1, 1, 1, 1]).eval()
    naf.nafnet.load_state_dict(load_sd(os.path.join(GOLDEN, "nafnet_small.pt")))
    dr = ref.DRCT(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30,
                  conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=[6, 6], embed_dim=60,
                  num_heads=[6, 6], mlp_ratio=2, upsampler="pixelshuffle", resi_connection="1conv").eval()
    dr.load_state_dict(load_sd(os.path.join(GOLDEN, "drct_small.pt")), strict=False)
    gr = ref.create_grl_model(embed_dim=60, depths=[2, 3], num_heads_w=[3, 3], num_heads_s=[3, 3]).eval()
    gr.load_state_dict(load_sd(os.path.join(GOLDEN, "grl_small.pt")), strict=False)
    ma = ref.MambaIR(upscale=4, in_chans=3, img_size=64, window_size=16, compress_ratio=3, squeeze_factor=30,
                     conv_scale=0.01, overlap_ratio=0.5, img_range=1.0, depths=(2,), embed_dim=48,
                     mlp_ratio=2.0, drop_path_rate=0.1, upsampler="pixelshuffle", resi_connection="1conv").eval()
    ma.load_state_dict(load_sd(os.path.join(GOLDEN, "mambair_small.pt")))
    print("  (skipped: fusion align layers expect 180-ch features; host logic is pinned per expert below)")
    lp, (h, w) = pipeline.pad16(lr)
    F = torch.nn.functional
    host = {"img": torch.from_numpy(img), "lr_padded": lp}
    bd, bg, bn, bm = hook_output(dr.conv_after_body), hook_output(gr.conv_after_body), \
        hook_input(naf.nafnet.ending), hook_output(ma.conv_after_body)
    host["drct_sr"] = dr(lp).clamp(0, 1)[:, :, :h * 4, :w * 4]
    host["drct_feat"] = bd["out"][:, :, :h, :w]
    host["grl_sr"] = gr(lp).clamp(0, 1)[:, :, :h * 4, :w * 4]
    host["grl_feat"] = bg["out"][:, :, :h, :w]
    host["naf_sr"] = naf(lp).clamp(0, 1)[:, :, :h * 4, :w * 4]
    host["naf_feat"] = F.interpolate(bn["out"], size=(h, w), mode="bilinear", align_corners=False)
    host["mamba_sr"] = ma(lp).clamp(0, 1)[:, :, :h * 4, :w * 4]
    host["mamba_feat"] = bm["out"][:, :, :h, :w]
    weights = {"drct": load_sd(os.path.join(GOLDEN, "drct_small.pt")),
               "grl": load_sd(os.path.join(GOLDEN, "grl_small.pt")),
               "nafnet": load_sd(os.path.join(GOLDEN, "nafnet_small.pt")),
               "mamba": load_sd(os.path.join(GOLDEN, "mambair_small.pt"))}
    cfg = torch.load(os.path.join(GOLDEN, "nafnet_small.pt"))["cfg"]
    imgs, feats, lr_in = pipeline.run_experts(weights, lr, naf_cfg=cfg)
    for n, key in (("drct", "drct"), ("grl", "grl"), ("nafnet", "naf"), ("mamba", "mamba")):
        report(f"{n} sr (pad16/crop/clamp)", imgs[n], host[f"{key}_sr"], 5e-5)
        report(f"{n} feat", feats[n], host[f"{key}_feat"], 5e-5)
    torch.save({k: (v.half() if v.dtype == torch.float32 and k not in ("lr_padded",) else v)
                for k, v in host.items()}, os.path.join(GOLDEN, "host_40x56.pt"))
    print("golden fixtures written to", GOLDEN)
    for f in sorted(os.listdir(GOLDEN)):
        print(f"  {f}: {os.path.getsize(os.path.join(GOLDEN, f)) / 1e6:.2f} MB")


def _load_reference_file(rel, name):
    """one reference source file imported from where it lies (no package __init__ executed)"""
    import importlib.util
    from ref_harness import REFERENCE_ROOT
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    return mod


def metric_pairs():
    """two seeded image pairs [1,3,H,W]: smooth image + noise (values also leave [0,1]: the metrics clamp)"""
    pairs = []
    for seed, (h, w), noise in ((21, (48, 64), 0.02), (22, (37, 53), 0.08)):
        g = torch.Generator().manual_seed(seed)
        a = lr_input(seed, 1, h, w)
        b = a + noise * torch.randn(a.shape, generator=g)
        pairs.append((a * 1.1 - 0.05, b))
    return pairs


@torch.no_grad()
def golden_metrics():
    """calculate_psnr / calculate_ssim_torch / calculate_ssim of src/utils/metrics.py (torch + numpy only;
    scikit-image is absent here, so calculate_ssim takes its torch branch) on two seeded pairs -> tests/golden/metrics.pt (inputs + expected values)"""
    from ffsr_oracle import metrics as om
    ref = _load_reference_file("src/utils/metrics.py", "ref_src_utils_metrics")
    out = []
    for i, (a, b) in enumerate(metric_pairs()):
        for crop in (0, 4):
            for y in (False, True):
                p = ref.calculate_psnr(a, b, crop_border=crop, test_y_channel=y)
                s = ref.calculate_ssim(a, b, crop_border=crop, test_y_channel=y)
                po, so = om.psnr(a, b, crop, y), om.ssim(a, b, crop, y)
                print(f"  pair {i} crop {crop} y {int(y)}: psnr {p:.6f} (oracle {po:.6f})  ssim {s:.8f} (oracle {so:.8f})")
                assert abs(p - po) < 1e-4 and abs(s - so) < 1e-6
                out.append({"pair": i, "crop_border": crop, "test_y_channel": y, "psnr": p, "ssim": s})
    torch.save({"pairs": [(a.clone(), b.clone()) for a, b in metric_pairs()], "cases": out},
               os.path.join(GOLDEN, "metrics.pt"))


def train_case(seed, b, h, w):
    """seeded cached-feature batch (SURVEY 8d config 5 pattern): lr, 4 expert images / features (fp16-exact), hr target"""
    g = torch.Generator().manual_seed(seed)
    lr = lr_input(seed, b, h, w)
    bic = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    imgs = {n: (bic + 0.03 * torch.randn(bic.shape, generator=g)).clamp(0, 1).half().float() for n in fusion.EXPERTS}
    feats = {n: torch.randn(b, 64 if n == "nafnet" else 180, h, w, generator=g).half().float() for n in fusion.EXPERTS}
    hr = (bic + 0.05 * torch.randn(bic.shape, generator=g)).clamp(0, 1)
    return lr, imgs, feats, hr


def golden_train():
    """the reference's training-mode forward + backward (train.py:323-336) on a seeded batch ->
    tests/golden/fusion_train.pt.  model.train() with the two nn.MultiheadAttention dropouts set to 0 (dropout
    disabled); weights = the state_dict stored in fusion_full.pt (not stored again).
    Stored: inputs, sr, loss = L1(sr.clamp(0,1), hr), the gradient of EVERY parameter, the BatchNorm running statistics
    after the step's forward.  Also checks the oracle's train mode + torch autograd against the reference."""
    ref = load_reference()
    m = ref.CompleteEnhancedFusionSR(expert_ensemble=None)
    missing = m.load_state_dict(load_sd(os.path.join(GOLDEN, "fusion_full.pt")), strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
    m.train()
    m.cross_band.band_attention.dropout = 0.0
    m.collaborative.cross_attn.dropout = 0.0
    lr, imgs, feats, hr = train_case(31, 2, 32, 32)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.endswith("num_batches_tracked")}
    sr = m.forward_with_precomputed(lr, imgs, feats)
    loss = torch.nn.functional.l1_loss(sr.clamp(0, 1), hr)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    stats = {k: v.detach().clone() for k, v in m.state_dict().items() if k.endswith(("running_mean", "running_var"))}
    # the oracle in train mode, differentiated by torch autograd
    sdo = {k: (v.clone().requires_grad_(True) if k in grads else v.clone()) for k, v in sd0.items()}
    sro = fusion.fusion_forward(sdo, lr, imgs, feats, train=True)
    report("train-mode sr", sro.detach(), sr.detach(), 2e-5)
    torch.nn.functional.l1_loss(sro.clamp(0, 1), hr).backward()
    worst = 0.0
    for k, gr in grads.items():
        go = sdo[k].grad
        assert go is not None, k
        rel = (go - gr).abs().max().item() / max(gr.abs().max().item(), 1e-12)
        worst = max(worst, rel)
        assert rel < 2e-3, (k, rel)
    print(f"  oracle autograd vs reference: worst per-tensor relative gradient error {worst:.2e} over {len(grads)} tensors")
    for k, v in stats.items():
        report(k, sdo[k].detach(), v, 1e-5)
    torch.save({"lr": lr, "hr": hr, "imgs": {k: v.half() for k, v in imgs.items()}, "feats": {k: v.half() for k, v in feats.items()},
                "sr": sr.detach(), "loss": loss.detach(), "grads": grads, "stats": stats},
               os.path.join(GOLDEN, "fusion_train.pt"))
    print(f"  fusion_train.pt: {os.path.getsize(os.path.join(GOLDEN, 'fusion_train.pt')) / 1e6:.2f} MB, loss {loss.item():.6f}")


FLAG_ARGS = {"dynamic_expert_selection": "enable_dynamic_selection", "cross_band_attention": "enable_cross_band_attn",
             "adaptive_frequency_bands": "enable_adaptive_bands", "multi_resolution_fusion": "enable_multi_resolution",
             "collaborative_learning": "enable_collaborative", "edge_enhancement": "enable_edge_enhance"}


def golden_flags():
    """VERDICT r2 missing #5: the network the reference builds when configs/train_config.yaml switches an improvement off
    (io.py:186-193 -> CompleteEnhancedFusionSR(enable_*=False)).  For each single improvement off, and all off, the REFERENCE's
    forward_with_precomputed on the 32x32 case -> tests/golden/fusion_flags.pt.  Weights = fusion_full.pt's state_dict for the
    modules that exist in the variant (not stored again) + the variant-only `simple_fusion` 1x1 convolution (stored, 39 values)."""
    ref = load_reference()
    full = load_sd(os.path.join(GOLDEN, "fusion_full.pt"))
    g = torch.Generator().manual_seed(90)
    simple = {"simple_fusion.weight": (torch.randn(3, 12, 1, 1, generator=g) * 0.3).half().float(),
              "simple_fusion.bias": (torch.randn(3, generator=g) * 0.05).half().float()}
    lr = lr_input(41, 1, 32, 32)
    bic = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    imgs = {n: (bic + 0.03 * torch.randn(bic.shape, generator=g)).clamp(0, 1).half().float() for n in fusion.EXPERTS}
    feats = {n: torch.randn(1, 64 if n == "nafnet" else 180, 32, 32, generator=g).half().float() for n in fusion.EXPERTS}
    variants = [{k: (k != off) for k in FLAG_ARGS} for off in FLAG_ARGS] + [{k: False for k in FLAG_ARGS}]
    outs = []
    for flags in variants:
        m = ref.CompleteEnhancedFusionSR(expert_ensemble=None, **{FLAG_ARGS[k]: v for k, v in flags.items()}).eval()
        sd = dict(full)
        sd.update(simple)
        missing = m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()}, strict=False)
        assert all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
        with torch.no_grad():
            want = m.forward_with_precomputed(lr, imgs, feats)
            got = fusion.fusion_forward(sd, lr, imgs, feats, flags=flags)
        off = [k for k, v in flags.items() if not v]
        report(f"fusion with {off} off", got, want, 2e-5)
        outs.append({"flags": flags, "out": want, "keys": sorted(k for k in m.state_dict() if not k.endswith("num_batches_tracked"))})
    torch.save({"lr": lr, "imgs": {k: v.half() for k, v in imgs.items()}, "feats": {k: v.half() for k, v in feats.items()},
                "simple": simple, "variants": outs}, os.path.join(GOLDEN, "fusion_flags.pt"))


if __name__ == "__main__":
    what = sys.argv[1:] or ["inference", "metrics", "train", "flags"]
    if "flags" in what:
        golden_flags()
    if "inference" in what:
        main()
    if "metrics" in what:
        golden_metrics()
    if "train" in what:
        golden_train()
