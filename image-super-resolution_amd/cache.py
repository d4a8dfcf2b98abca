"""The reference's cached expert-feature format (the data format either side of the hot path): written by its extraction
scripts after running the four frozen experts, read back by ``CachedSRDataset`` for the cached-feature training step.

  {stem}_drct_part.pt   {'outputs': {'drct': [1,3,4h,4w]}, 'features': {'drct': [1,180,h,w]}, 'lr': [3,h,w], 'hr': [3,4h,4w],
                         'filename': stem}                                  scripts/extract_features_balanced.py:162-169
  {stem}_rest_part.pt   {'outputs': {'grl', 'nafnet'}, 'features': {'grl': [1,180,h,w], 'nafnet': [1,64,h,w]}, 'filename'}
                                                                            scripts/extract_features_balanced.py:172-183
  {stem}_mamba_part.pt  {'outputs': {'mamba': fp16}, 'features': {'mamba': fp16 [1,180,h,w]}, 'filename'}
                                                                            scripts/extract_mamba_features.py:226-237

``extract`` produces one entry with the HIP engine (Engine.run_experts = the expert half of io._process_image);
``load_entry`` follows ``CachedSRDataset.__getitem__`` (src/data/cached_dataset.py:135-226, augmentation excluded) but
reads with ``weights_only=True``: nothing from a cache file is executed.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

PARTS = ("drct", "rest", "mamba")


def _path(cache_dir, stem, part):
    return os.path.join(str(cache_dir), f"{stem}_{part}_part.pt")


def extract(engine, lr: torch.Tensor):
    """lr [1,3,h,w] float in [0,1] (host) -> (outputs, features) dicts of CPU fp32 NCHW tensors with the reference's
    shapes: outputs[name] [1,3,4h,4w] (clamped), features drct/grl/mamba [1,180,h,w], nafnet [1,64,h,w]."""
    from . import engine as E
    imgs, feats = engine.run_experts(E.nchw_to_map(lr, engine.device))
    torch.cuda.current_stream(engine.device).synchronize()
    return ({k: E.map_to_nchw(v[..., :3]) for k, v in imgs.items()}, {k: E.map_to_nchw(v) for k, v in feats.items()})


def save_entry(cache_dir, stem: str, lr: torch.Tensor, hr: torch.Tensor, outputs: Dict[str, torch.Tensor],
               features: Dict[str, torch.Tensor], parts=PARTS):
    """lr [3,h,w], hr [3,4h,4w]; outputs / features as returned by extract().  Writes the parts listed in `parts`."""
    os.makedirs(str(cache_dir), exist_ok=True)
    cpu = lambda t: t.detach().to("cpu", torch.float32).contiguous()
    if "drct" in parts:
        torch.save({"outputs": {"drct": cpu(outputs["drct"])}, "features": {"drct": cpu(features["drct"])},
                    "lr": cpu(lr), "hr": cpu(hr), "filename": stem}, _path(cache_dir, stem, "drct"))
    if "rest" in parts:
        torch.save({"outputs": {k: cpu(outputs[k]) for k in ("grl", "nafnet")},
                    "features": {k: cpu(features[k]) for k in ("grl", "nafnet")}, "filename": stem},
                   _path(cache_dir, stem, "rest"))
    if "mamba" in parts:
        torch.save({"outputs": {"mamba": cpu(outputs["mamba"]).half()}, "features": {"mamba": cpu(features["mamba"]).half()},
                    "filename": stem}, _path(cache_dir, stem, "mamba"))


def list_stems(cache_dir) -> List[str]:
    """stems with a drct part AND a rest part, sorted (cached_dataset.py:84-110)"""
    names = sorted(f[:-len("_drct_part.pt")] for f in os.listdir(str(cache_dir)) if f.endswith("_drct_part.pt"))
    return [s for s in names if os.path.exists(_path(cache_dir, s, "rest"))]


def load_entry(cache_dir, stem: str, load_features: bool = True) -> Dict[str, object]:
    """-> {'lr' [3,h,w], 'hr' [3,4h,4w], 'expert_imgs' {name: [3,4h,4w]}, 'expert_feats' {name: [C,h,w]}, 'filename'};
    a missing mamba part degrades to zeros exactly as the reference does (cached_dataset.py:176-186, 203-206)."""
    load = lambda part: torch.load(_path(cache_dir, stem, part), map_location="cpu", weights_only=True)
    d, r = load("drct"), load("rest")
    lr, hr = d["lr"], d["hr"]
    imgs = dict(d["outputs"])
    imgs.update(r["outputs"])
    m: Optional[dict] = load("mamba") if os.path.exists(_path(cache_dir, stem, "mamba")) else None
    if m is not None:
        for k, v in m["outputs"].items():
            imgs[k] = v.float()
    else:
        imgs["mamba"] = torch.zeros(next(iter(imgs.values())).shape)
    imgs = {k: (v.squeeze(0) if v.dim() == 4 else v) for k, v in imgs.items()}
    out = {"lr": lr, "hr": hr, "expert_imgs": imgs, "filename": stem}
    if load_features:
        feats = dict(d.get("features", {}))
        feats.update(r.get("features", {}))
        if m is not None:
            for k, v in m.get("features", {}).items():
                feats[k] = v.float()
        else:
            feats["mamba"] = torch.zeros(1, 180, lr.shape[-2], lr.shape[-1])
        out["expert_feats"] = {k: (v.squeeze(0) if v.dim() == 4 else v) for k, v in feats.items()}
    return out


def collate(entries: List[Dict[str, object]]) -> Dict[str, object]:
    """default_collate of the DataLoader for equally sized patches: stacks along a new batch axis"""
    out = {"lr": torch.stack([e["lr"] for e in entries]), "hr": torch.stack([e["hr"] for e in entries]),
           "expert_imgs": {k: torch.stack([e["expert_imgs"][k] for e in entries]) for k in entries[0]["expert_imgs"]},
           "filename": [e["filename"] for e in entries]}
    if "expert_feats" in entries[0]:
        out["expert_feats"] = {k: torch.stack([e["expert_feats"][k] for e in entries]) for k in entries[0]["expert_feats"]}
    return out


# ---------------------------------------------------------------------------------------------------------------------
# 8x geometric-TTA test cache + cached submission: scripts/extract_test_tta_cache.py:248-330 writes, per
# test image and per variant t0..t7 = {hflip} x {rot90^k}, the same three part files -- every tensor fp16, the drct part
# also carries the transformed LR, the original size and the transform; scripts/generate_fast_submission.py:188-256 reads
# them back, runs ONLY the fusion network per variant, undoes the transform, averages and clamps.
TTA_CONFIGS = tuple((hflip, rot) for hflip in (False, True) for rot in (0, 1, 2, 3))      # extract_test_tta_cache.py:253-256


def tta_stem(lr_path_or_stem: str) -> str:
    """'0901x4.png' -> '0901' (extract_test_tta_cache.py:264-265)"""
    raw = os.path.splitext(os.path.basename(str(lr_path_or_stem)))[0]
    return raw.replace("x4", "").rstrip("_")


def extract_tta(engine, lr: torch.Tensor, cache_dir, stem: str, resume: bool = False) -> int:
    """lr [1,3,h,w] float (host) -> the 24 part files {stem}_t{0..7}_{drct,rest,mamba}_part.pt.  Each variant is the
    transformed LR image through Engine.run_experts (pad16 -> four experts -> crops / clamps / NAFNet feature resample),
    stored as fp16 like the reference (run_expert_sequential :108-177).  Returns the number of variants written."""
    from . import engine as E, ops
    os.makedirs(str(cache_dir), exist_ok=True)
    half = lambda t: t.detach().to("cpu").half().contiguous()
    base = E.nchw_to_map(lr, engine.device)
    written = 0
    for t_idx, (hflip, rot) in enumerate(TTA_CONFIGS):
        t_stem = f"{stem}_t{t_idx}"
        if resume and os.path.exists(_path(cache_dir, t_stem, "drct")):
            continue
        variant = ops.dihedral(base, hflip, rot)
        imgs, feats = engine.run_experts(variant)
        torch.cuda.current_stream(engine.device).synchronize()
        out = {k: half(E.map_to_nchw(v[..., :3])) for k, v in imgs.items()}
        ft = {k: half(E.map_to_nchw(v)) for k, v in feats.items()}
        lr_v = E.map_to_nchw(variant)
        torch.save({"outputs": {"drct": out["drct"]}, "features": {"drct": ft["drct"]}, "lr": half(lr_v[0]), "filename": t_stem,
                    "original_stem": stem, "original_size": (int(lr_v.shape[2]), int(lr_v.shape[3])),
                    "tta_info": {"hflip": bool(hflip), "rot": int(rot), "t_idx": t_idx}}, _path(cache_dir, t_stem, "drct"))
        torch.save({"outputs": {k: out[k] for k in ("grl", "nafnet")}, "features": {k: ft[k] for k in ("grl", "nafnet")},
                    "filename": t_stem}, _path(cache_dir, t_stem, "rest"))
        torch.save({"outputs": {"mamba": out["mamba"]}, "features": {"mamba": ft["mamba"]}, "filename": t_stem},
                   _path(cache_dir, t_stem, "mamba"))
        written += 1
    return written


def list_tta_stems(cache_dir) -> List[str]:
    """unique image stems of a TTA cache (generate_fast_submission.py:174-176)"""
    return sorted(f[:-len("_t0_drct_part.pt")] for f in os.listdir(str(cache_dir)) if f.endswith("_t0_drct_part.pt"))


def fuse_tta(engine, cache_dir, stem: str, num_variants: int = 8) -> torch.Tensor:
    """generate_fast_submission.py:200-250 for one image: every cached variant through the fusion network only
    (Engine.fusion = forward_with_precomputed), inverse transform, mean over the variants found, clamp.
    -> SR map [1,4h,4w,3] on the device.  Files are read with weights_only=True."""
    from . import engine as E, ops
    acc, n = None, 0
    maps = []
    for t_idx in range(num_variants):
        t_stem = f"{stem}_t{t_idx}"
        if not os.path.exists(_path(cache_dir, t_stem, "drct")):
            continue                                                  # the reference warns and skips (:212-214)
        d, r, m = (torch.load(_path(cache_dir, t_stem, p), map_location="cpu", weights_only=True) for p in PARTS)
        info = d.get("tta_info", {"hflip": False, "rot": 0})
        dev_map = lambda t: E.nchw_to_map(t.float() if t.dim() == 4 else t.float().unsqueeze(0), engine.device)
        imgs = {"drct": dev_map(d["outputs"]["drct"]), "grl": dev_map(r["outputs"]["grl"]),
                "nafnet": dev_map(r["outputs"]["nafnet"]), "mamba": dev_map(m["outputs"]["mamba"])}
        feats = {"drct": dev_map(d["features"]["drct"]), "grl": dev_map(r["features"]["grl"]),
                 "nafnet": dev_map(r["features"]["nafnet"]), "mamba": dev_map(m["features"]["mamba"])}
        sr = engine.fusion(dev_map(d["lr"]), imgs, feats)
        maps.append((sr, bool(info["hflip"]), int(info["rot"])))
    if not maps:
        raise FileNotFoundError(f"no cached TTA variants of {stem!r} in {cache_dir}")
    for sr, hflip, rot in maps:                                       # mean of the de-transformed outputs, then clamp (:245)
        acc = ops.dihedral(sr, hflip, rot, inverse=True, out=acc, scale=1.0 / len(maps), accumulate=acc is not None)
    return ops.unary(acc, clamp=(0.0, 1.0), out=acc)


def generate_submission(engine, cache_dir, output_dir, no_tta: bool = False) -> List[str]:
    """the image loop of generate_fast_submission.py:188-256: one '{stem}x4.png' per cached test image"""
    from PIL import Image
    os.makedirs(str(output_dir), exist_ok=True)
    names = []
    for stem in list_tta_stems(cache_dir):
        sr = fuse_tta(engine, cache_dir, stem, 1 if no_tta else 8)
        name = f"{stem}x4.png"                                        # NTIRE naming (:248-251)
        Image.fromarray(engine.download(sr)).save(os.path.join(str(output_dir), name), compress_level=1)
        names.append(name)
    return names
