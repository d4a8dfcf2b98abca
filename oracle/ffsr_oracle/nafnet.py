"""NAFNet-SIDD-width64 wrapped as an x4 SR expert -- CPU oracle (test infrastructure).

Follows src/models/nafnet/nafnet_arch.py (LayerNorm2d :26, SimpleGate :47, NAFBlock.forward :110,
NAFNet.forward :195, check_image_size :219) and src/models/nafnet/__init__.py (NAFNetSR.forward
:117: bicubic x4 -> U-Net at HR -> clamp).  ``sd`` uses the keys of the inner ``NAFNet`` module
(== the official checkpoint: ``intro.weight``, ``encoders.0.0.conv1.weight`` ...).
"""
import torch
import torch.nn.functional as F

ENC_BLKS = (2, 2, 4, 8)
MID_BLKS = 12
DEC_BLKS = (2, 2, 2, 2)


def layernorm2d(x, w, b, eps=1e-6):
    mu = x.mean(1, keepdim=True)
    var = (x - mu).pow(2).mean(1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w[None, :, None, None] + b[None, :, None, None]


def simple_gate(x):
    a, b = x.chunk(2, dim=1)
    return a * b


def naf_block(sd, p, x):
    c = x.shape[1]
    t = layernorm2d(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    t = F.conv2d(t, sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    t = F.conv2d(t, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1, groups=2 * c)
    t = simple_gate(t)
    pooled = t.mean((2, 3), keepdim=True)
    t = t * F.conv2d(pooled, sd[p + "sca.1.weight"], sd[p + "sca.1.bias"])
    t = F.conv2d(t, sd[p + "conv3.weight"], sd[p + "conv3.bias"])
    y = x + t * sd[p + "beta"]
    t = F.conv2d(layernorm2d(y, sd[p + "norm2.weight"], sd[p + "norm2.bias"]),
                 sd[p + "conv4.weight"], sd[p + "conv4.bias"])
    t = simple_gate(t)
    t = F.conv2d(t, sd[p + "conv5.weight"], sd[p + "conv5.bias"])
    return y + t * sd[p + "gamma"]


def nafnet_unet(sd, inp, enc_blks=ENC_BLKS, mid_blks=MID_BLKS, dec_blks=DEC_BLKS):
    """Returns (output, feature entering ``ending``) -- the latter is what the reference hook on
    ``ending`` captures as INPUT (expert_loader.py:553-558)."""
    _, _, H, W = inp.shape
    mult = 2 ** len(enc_blks)
    inp = F.pad(inp, (0, (mult - W % mult) % mult, 0, (mult - H % mult) % mult))
    x = F.conv2d(inp, sd["intro.weight"], sd["intro.bias"], padding=1)
    skips = []
    for i, n in enumerate(enc_blks):
        for j in range(n):
            x = naf_block(sd, f"encoders.{i}.{j}.", x)
        skips.append(x)
        x = F.conv2d(x, sd[f"downs.{i}.weight"], sd[f"downs.{i}.bias"], stride=2)
    for j in range(mid_blks):
        x = naf_block(sd, f"middle_blks.{j}.", x)
    for i, n in enumerate(dec_blks):
        x = F.pixel_shuffle(F.conv2d(x, sd[f"ups.{i}.0.weight"]), 2)
        x = x + skips[-1 - i]
        for j in range(n):
            x = naf_block(sd, f"decoders.{i}.{j}.", x)
    feat = x
    x = F.conv2d(x, sd["ending.weight"], sd["ending.bias"], padding=1) + inp
    return x[:, :, :H, :W], feat


def nafnet_sr(sd, lr, scale=4, **kw):
    """lr [B,3,h,w] -> (sr clamped to [0,1] [B,3,4h,4w], feat [B,width,4h(+pad),4w(+pad)])."""
    up = F.interpolate(lr, scale_factor=scale, mode="bicubic", align_corners=False)
    out, feat = nafnet_unet(sd, up, **kw)
    return out.clamp(0, 1), feat
