"""SURVEY 8 f2, optimiser side: the HIP loss / clip / AdamW / EMA kernels against torch's own autograd, AdamW and
clip_grad_norm_ on the CPU (oracle/ffsr_oracle/train.py); cache formats.  The backward pass itself: tests/test_gpu_backward.py."""
import importlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mod(name):
    return importlib.import_module("image-super-resolution_amd." + name)


@pytest.mark.parametrize("B,H,W,acc", [(2, 64, 64, 1), (32, 256, 256, 4), (1, 7, 13, 2)])
def test_l1_clamp_loss_and_gradient(B, H, W, acc):
    """config 5 geometry: 32 x 3 x 256 x 256 HR patches; values outside [0, 1] and exact ties exercise clamp / sign"""
    from ffsr_oracle import train as otrain
    T, E = mod("train"), mod("engine")
    g = torch.Generator().manual_seed(1)
    sr = torch.rand(B, 3, H, W, generator=g) * 1.4 - 0.2
    hr = torch.rand(B, 3, H, W, generator=g)
    sr[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.5, 0.25])
    hr[0, 0, 0, :4] = torch.tensor([0.3, 0.7, 0.5, 0.25])                 # boundaries of the clamp, sign(0) = 0
    want_loss, want_grad = otrain.l1_clamp_loss_and_grad(sr, hr, acc)
    loss, grad = T.l1_clamp_loss(E.nchw_to_map(sr, DEV), E.nchw_to_map(hr, DEV), accumulation_steps=acc)
    assert abs(loss.item() - want_loss.item()) < 2e-7 * max(1.0, abs(want_loss.item())) + 1e-7
    assert torch.equal(E.map_to_nchw(grad), want_grad)
    only_loss, none = T.l1_clamp_loss(E.nchw_to_map(sr, DEV), E.nchw_to_map(hr, DEV), accumulation_steps=acc, need_grad=False)
    assert none is None and only_loss.item() == loss.item()


@pytest.mark.parametrize("max_norm,grad_scale", [(1.0, 5.0), (1.0, 1e-3), (0.0, 1.0)])
def test_adamw_clip_ema_steps_match_torch(max_norm, grad_scale):
    """five steps on the real fusion-net tensors (1.4 M parameters): clipping active / inactive / disabled"""
    from ffsr_oracle import train as otrain
    T, W = mod("train"), mod("weights")
    sd = {k: v for k, v in W.fusion_state_dict(seed=5).items() if v.is_floating_point() and v.numel() > 0}
    opt = T.FusionOptimizer(sd, DEV, max_norm=max_norm)
    ref = otrain.Trainer(sd, max_norm=max_norm)
    g = torch.Generator().manual_seed(2)
    for it in range(5):
        grads = {k: torch.randn(v.shape, generator=g) * grad_scale / (1 + it) for k, v in sd.items()}
        for k, view in opt.views(opt.grad).items():
            view.copy_(grads[k])
        norm = opt.grad_norm().item()
        opt.step()
        want_norm = ref.step(grads)
        if want_norm is not None:
            assert abs(norm - want_norm.item()) <= 1e-5 * want_norm.item()
    torch.cuda.synchronize()
    got_p, got_e = opt.views(), opt.views(opt.ema)
    got_m, got_v = opt.views(opt.exp_avg), opt.views(opt.exp_avg_sq)
    for i, (k, p) in enumerate(ref.params.items()):
        st = ref.opt.state[p]
        for name, got, want in (("param", got_p[k], p.data), ("ema", got_e[k], ref.shadow[k]),
                                ("exp_avg", got_m[k], st["exp_avg"]), ("exp_avg_sq", got_v[k], st["exp_avg_sq"])):
            err = (got.cpu() - want).abs().max().item()
            assert err <= 2e-6 * max(1.0, want.abs().max().item()), (k, name, err)


def test_train_entry_points_reject_bad_arguments():
    hip = mod("hip")
    x = torch.zeros(64, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    with pytest.raises(hip.FfsrError, match="invalid argument"):
        hip.call("ffsr_sumsq_f32", x.data_ptr(), 64, x.data_ptr(), 0, x.data_ptr(), st)             # no scratch
    with pytest.raises(hip.FfsrError, match="invalid argument"):
        hip.call("ffsr_adamw_ema_f32", x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, 64, None, 0.0, 1e-3,
                 0.9, 0.999, 1e-8, 0.0, 0, 0.0, st)                                                  # step 0
    with pytest.raises(hip.FfsrError, match="invalid argument"):
        hip.call("ffsr_l1_clamp_loss_f32", x.data_ptr(), 2, x.data_ptr(), 3, None, 0, x.data_ptr(), 1024, x.data_ptr(), 8, 3,
                 1.0, st)                                                                            # row stride < C


def test_cache_extract_writes_what_the_oracle_experts_produce(tmp_path):
    """the cache producer: four experts on the HIP engine -> the reference's 3-part files -> CachedSRDataset's view,
    against the oracle's expert outputs / hooked features (expert half of io._process_image; 40x56 needs pad16 + crops)"""
    from ffsr_oracle import pipeline
    from ffsr_oracle.scan_c import selective_scan_c
    from test_gpu_models import lr_image
    W, E, C = mod("weights"), mod("engine"), mod("cache")
    weights = W.random_weights(seed=40, small=True)
    eng = E.Engine(weights, DEV)
    lr = lr_image(31, 1, 40, 56)
    hr = torch.rand(3, 160, 224, generator=torch.Generator().manual_seed(3))
    outs, feats = C.extract(eng, lr)
    imgs_o, feats_o, _ = pipeline.run_experts(weights, lr, naf_cfg=dict(enc_blks=(1, 1, 1, 1), mid_blks=1, dec_blks=(1, 1, 1, 1)),
                                              scan_fn=selective_scan_c)
    for k in ("drct", "grl", "nafnet", "mamba"):
        assert tuple(outs[k].shape) == (1, 3, 160, 224) and tuple(feats[k].shape) == tuple(feats_o[k].shape)
        assert (outs[k] - imgs_o[k]).abs().max().item() < 1e-3 and (feats[k] - feats_o[k]).abs().max().item() < 1e-3, k
    C.save_entry(tmp_path, "0801x4", lr[0], hr, outs, feats)
    e = C.load_entry(tmp_path, C.list_stems(tmp_path)[0])
    assert torch.equal(e["expert_imgs"]["grl"], outs["grl"][0]) and torch.equal(e["expert_feats"]["nafnet"], feats["nafnet"][0])
    assert (e["expert_imgs"]["mamba"] - imgs_o["mamba"][0]).abs().max().item() < 2e-3          # fp16 storage of the reference
    # the loaded entry drives the fusion net exactly like the live path
    lrm = E.nchw_to_map(lr, DEV)
    imgs_m = {k: E.nchw_to_map(e["expert_imgs"][k][None], DEV) for k in ("drct", "grl", "nafnet")}
    imgs_m["mamba"] = E.nchw_to_map(outs["mamba"], DEV)
    feats_m = {k: E.nchw_to_map(feats[k], DEV) for k in feats}
    live = E.map_to_nchw(eng.process(lrm))
    cached = E.map_to_nchw(eng.fusion(lrm, imgs_m, feats_m))
    assert (live - cached).abs().max().item() < 1e-6


def test_tta_cache_roundtrip_matches_on_the_fly_tta(tmp_path):
    """SURVEY 8 f3, cached route: extract_tta writes the 8 x 3 part files of scripts/extract_test_tta_cache.py (fp16, with
    tta_info); fuse_tta = scripts/generate_fast_submission.py's per-image loop.  The result must agree with the on-the-fly
    Engine.process_tta up to the fp16 storage of the cached expert outputs, and generate_submission writes '{stem}x4.png'."""
    from test_gpu_models import lr_image
    W, E, C = mod("weights"), mod("engine"), mod("cache")
    eng = E.Engine(W.random_weights(seed=41, small=True), DEV)
    lr = lr_image(33, 1, 24, 40)                                              # non-square: rot90 variants swap H and W
    stem = C.tta_stem("0901x4.png")
    assert stem == "0901" and C.extract_tta(eng, lr, tmp_path, stem) == 8
    assert C.extract_tta(eng, lr, tmp_path, stem, resume=True) == 0 and C.list_tta_stems(tmp_path) == ["0901"]
    d5 = torch.load(tmp_path / "0901_t5_drct_part.pt", weights_only=True)
    assert d5["tta_info"] == {"hflip": True, "rot": 1, "t_idx": 5} and tuple(d5["lr"].shape) == (3, 40, 24)
    assert d5["lr"].dtype == torch.float16 and tuple(d5["outputs"]["drct"].shape) == (1, 3, 160, 96)
    assert torch.equal(d5["lr"].float(), torch.rot90(torch.flip(lr, [3]), 1, [2, 3])[0].half().float())
    cached = E.map_to_nchw(C.fuse_tta(eng, tmp_path, stem))
    live = E.map_to_nchw(eng.process_tta(E.nchw_to_map(lr, DEV)))
    assert tuple(cached.shape) == (1, 3, 96, 160) and (cached - live).abs().max().item() < 5e-3      # fp16 cache
    one = E.map_to_nchw(C.fuse_tta(eng, tmp_path, stem, num_variants=1))
    assert (one - E.map_to_nchw(eng.process(E.nchw_to_map(lr, DEV)))).abs().max().item() < 5e-3
    names = C.generate_submission(eng, tmp_path, tmp_path / "res")
    assert names == ["0901x4.png"]
    from PIL import Image
    with Image.open(tmp_path / "res" / "0901x4.png") as im:
        assert im.size == (160, 96)


def test_cached_dataset_entries_drive_the_training_step(tmp_path):
    """SURVEY 8 f2 end to end, the reference's data flow: frozen experts on the HIP engine -> 3-part cache files
    (scripts/extract_features_balanced.py format) -> CachedSRDataset-style load + default_collate (cache.load_entry / collate)
    -> train_epoch_cached's step (train.FusionTrainer.step) on the fusion network, with gradient accumulation over two
    micro-batches.  Checks the plumbing: shapes, finite loss, parameters move once per accumulation window."""
    from test_gpu_models import lr_image
    W, E, C, T = mod("weights"), mod("engine"), mod("cache"), mod("train")
    weights = W.random_weights(seed=43, small=True)
    eng = E.Engine(weights, DEV)
    g = torch.Generator().manual_seed(5)
    for i in range(4):
        lr = lr_image(50 + i, 1, 32, 32)
        hr = torch.rand(3, 128, 128, generator=g)
        outs, feats = C.extract(eng, lr)
        C.save_entry(tmp_path, f"{i:04d}x4", lr[0], hr, outs, feats)
    stems = C.list_stems(tmp_path)
    assert len(stems) == 4
    tr = T.FusionTrainer(weights["fusion"], DEV, accumulation_steps=2, attn_dropout=0.0)
    p0 = tr.opt.param.clone()
    losses = []
    for b0 in (0, 2):
        batch = C.collate([C.load_entry(tmp_path, s) for s in stems[b0:b0 + 2]])
        assert tuple(batch["lr"].shape) == (2, 3, 32, 32) and tuple(batch["expert_feats"]["nafnet"].shape) == (2, 64, 32, 32)
        to_map = lambda t: E.nchw_to_map(t.float(), DEV)
        losses.append(tr.step(to_map(batch["lr"]), to_map(batch["hr"]), {k: to_map(v) for k, v in batch["expert_imgs"].items()},
                              {k: to_map(v) for k, v in batch["expert_feats"].items()}).item())
        assert (tr.opt.step_count == 1) == (b0 == 2)
    assert all(0.0 < v < 1.0 for v in losses) and not torch.equal(tr.opt.param, p0)
    sd = tr.state_dict(ema=True)
    assert set(sd) >= {"refine.0.weight", "cross_band.lka_block.norm1.running_mean"}
    # the key set is the reference's CompleteEnhancedFusionSR.state_dict() (manifest = its keys minus the six int64
    # BatchNorm counters, oracle/make_manifest.py) so that load_state_dict(strict=True) accepts a checkpoint written from it;
    # the counters advance once per train-mode application (9 bands share cross_band's block, 4 experts lka_global)
    import json
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["fusion"]
    bns = [q + n for q in ("cross_band.lka_block.", "collaborative.lka_global.") for n in ("norm1", "norm2", "lka.bn")]
    assert set(sd) == set(man) | {b + ".num_batches_tracked" for b in bns}
    assert all(sd[b + ".num_batches_tracked"].dtype == torch.long for b in bns)
    assert int(sd["cross_band.lka_block.norm1.num_batches_tracked"]) == 9 * 2
    assert int(sd["collaborative.lka_global.lka.bn.num_batches_tracked"]) == 4 * 2
