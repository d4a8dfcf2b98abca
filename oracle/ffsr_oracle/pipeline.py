"""Per-image host logic of the NTIRE entry point -- CPU oracle (test infrastructure).

Follows models/team29_FreqFusionSR/io.py: _pad16 :71, _process_image :222-289 (expert order,
crop / clamp / float, the NAFNet feature bilinear from the PADDED HR map straight to (h, w)),
_uint2tensor4 :100, _tensor2uint :107, and expert_loader.py forward_drct/grl/nafnet :427-471.
"""
import numpy as np
import torch
import torch.nn.functional as F
from . import drct, grl, nafnet, mambair, fusion


def pad16(t):
    _, _, h, w = t.shape
    ph, pw = (16 - h % 16) % 16, (16 - w % 16) % 16
    if ph or pw:
        t = F.pad(t, (0, pw, 0, ph), mode="reflect")
    return t, (h, w)


def uint2tensor4(img):
    return torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255.0).unsqueeze(0)


def tensor2uint(t):
    a = t.detach().squeeze().float().clamp(0, 1).cpu().numpy()
    if a.ndim == 3:
        a = np.transpose(a, (1, 2, 0))
    return np.uint8((a * 255.0).round())


def run_experts(weights, lr, scale=4, naf_cfg=None, scan_fn=None):
    """weights: dict drct/grl/nafnet/mamba -> state_dict.  Returns (imgs, feats, lr_in) like io._process_image."""
    lp, (h, w) = pad16(lr)
    imgs, feats = {}, {}
    sr, f = drct.drct_forward(weights["drct"], lp)
    imgs["drct"], feats["drct"] = sr.clamp(0, 1)[:, :, :h * scale, :w * scale], f[:, :, :h, :w]
    sr, f = grl.grl_forward(weights["grl"], lp)
    imgs["grl"], feats["grl"] = sr.clamp(0, 1)[:, :, :h * scale, :w * scale], f[:, :, :h, :w]
    sr, f = nafnet.nafnet_sr(weights["nafnet"], lp, scale, **(naf_cfg or {}))
    imgs["nafnet"] = sr[:, :, :h * scale, :w * scale]
    feats["nafnet"] = F.interpolate(f, size=(h, w), mode="bilinear", align_corners=False)
    sr, f = mambair.mambair_forward(weights["mamba"], lp, **({"scan_fn": scan_fn} if scan_fn else {}))
    imgs["mamba"], feats["mamba"] = sr.clamp(0, 1)[:, :, :h * scale, :w * scale], f[:, :, :h, :w]
    return imgs, feats, lp[:, :, :h, :w]


def process_image(weights, lr, scale=4, naf_cfg=None, scan_fn=None):
    """lr float [1,3,h,w] in [0,1] -> SR [1,3,4h,4w] in [0,1]."""
    imgs, feats, lr_in = run_experts(weights, lr, scale, naf_cfg, scan_fn)
    return fusion.fusion_forward(weights["fusion"], lr_in, imgs, feats, scale)
