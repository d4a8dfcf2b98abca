"""Live re-check of the oracle against the REFERENCE itself, where the reference tree is present (the build container;
skipped on the GPU box, which never has /root/reference).  The committed fixtures under tests/golden/ were produced by the
same comparison (oracle/make_golden.py); this test repeats a slice of it on fresh seeded inputs so that an edit of the
oracle cannot drift away from the reference unnoticed between fixture regenerations."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import ref_harness  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not present")


def _fusion_model():
    from conftest import load_golden
    ref = ref_harness.load_reference()
    m = ref.CompleteEnhancedFusionSR(expert_ensemble=None)
    sd = load_golden("fusion_full.pt")["sd"]
    m.load_state_dict(sd, strict=False)
    return m, sd


def test_fusion_eval_forward_fresh_input():
    from ffsr_oracle import fusion
    from make_golden import train_case
    m, sd = _fusion_model()
    lr, imgs, feats, _ = train_case(123, 1, 24, 40)
    with torch.no_grad():
        want = m.eval().forward_with_precomputed(lr, imgs, feats)
        got = fusion.fusion_forward(sd, lr, imgs, feats)
    assert (got - want).abs().max().item() < 2e-5


def test_fusion_train_forward_and_gradients_fresh_input():
    import torch.nn.functional as F
    from ffsr_oracle import fusion
    from make_golden import train_case
    m, sd = _fusion_model()
    m.train()
    m.cross_band.band_attention.dropout = 0.0
    m.collaborative.cross_attn.dropout = 0.0
    lr, imgs, feats, hr = train_case(124, 2, 16, 16)
    F.l1_loss(m.forward_with_precomputed(lr, imgs, feats).clamp(0, 1), hr).backward()
    names = {k for k, _ in m.named_parameters()}
    sdo = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in sd.items()}
    F.l1_loss(fusion.fusion_forward(sdo, lr, imgs, feats, train=True).clamp(0, 1), hr).backward()
    for k, p in m.named_parameters():
        err = (sdo[k].grad - p.grad).abs().max().item()
        assert err <= 2e-3 * p.grad.abs().max().item() + 1e-12, (k, err)


def test_metrics_fresh_pair():
    from ffsr_oracle import metrics as om
    from make_golden import _load_reference_file, lr_input
    ref = _load_reference_file("src/utils/metrics.py", "ref_src_utils_metrics_live")
    a = lr_input(55, 1, 40, 56)
    b = (a + 0.04 * torch.randn(a.shape, generator=torch.Generator().manual_seed(5))).clamp(-0.1, 1.1)
    for crop, y in ((0, False), (4, True)):
        assert abs(om.psnr(a, b, crop, y) - ref.calculate_psnr(a, b, crop_border=crop, test_y_channel=y)) < 1e-4
        assert abs(om.ssim(a, b, crop, y) - ref.calculate_ssim(a, b, crop_border=crop, test_y_channel=y)) < 1e-6
