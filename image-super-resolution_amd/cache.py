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
