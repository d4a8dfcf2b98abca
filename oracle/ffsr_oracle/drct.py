"""DRCT-L x4 -- CPU oracle (test infrastructure).  Follows src/models/drct/drct_arch.py:
WindowAttention.forward :175, SwinTransformerBlock.forward :376 (+calculate_mask :353),
RDG.forward :292, DRCT.forward_features :761, DRCT.forward :777; factory drct/__init__.py:86
(embed 180, 12 RDG, heads 6, window 16, mlp_ratio 2, gc 32).  ``sd`` = reference state_dict keys.
"""
import torch
import torch.nn.functional as F
from .common import (rgb_mean, to_tokens, to_image, win_split, win_merge, shift_mask, sr_tail)


def rel_pos_index(ws):
    c = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0) + (ws - 1)
    return rel[..., 0] * (2 * ws - 1) + rel[..., 1]                      # [ws*ws, ws*ws]


def swin_block(sd, p, x, H, W, ws, shift, heads):
    """x [B, HW, C].  LN -> (shifted) window MSA with relative position bias -> +res; LN -> MLP -> +res."""
    B, L, C = x.shape
    hd = C // heads
    t = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"]).reshape(B, H, W, C)
    if shift:
        t = torch.roll(t, (-shift, -shift), (1, 2))
    w = win_split(t, ws, ws)
    qkv = F.linear(w, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(-1, ws * ws, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    a = q @ k.transpose(-2, -1)
    bias = sd[p + "attn.relative_position_bias_table"][rel_pos_index(ws).reshape(-1)]
    a = a + bias.reshape(ws * ws, ws * ws, heads).permute(2, 0, 1)[None]
    if shift:
        m = shift_mask(H, W, ws, ws, shift, shift)                      # [nW, N, N]
        a = (a.reshape(B, -1, heads, ws * ws, ws * ws) + m[None, :, None]).reshape(-1, heads, ws * ws, ws * ws)
    o = (a.softmax(-1) @ v).transpose(1, 2).reshape(-1, ws * ws, C)
    o = F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    t = win_merge(o, ws, ws, H, W)
    if shift:
        t = torch.roll(t, (shift, shift), (1, 2))
    x = x + t.reshape(B, L, C)
    t = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    t = F.linear(F.gelu(F.linear(t, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                 sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + t


def rdg(sd, p, x, H, W, ws, base_heads, gc=32):
    """Residual dense group: 5 Swin blocks on a growing channel concat (dims C, C+gc, ...);
    heads_j = base - (dim_j % base); shifts 0, ws/2, 0, ws/2, 0; 1x1 'adjust' convs."""
    C = x.shape[-1]
    feats = [x]
    for j in range(5):
        cat = torch.cat(feats, -1)
        dim = cat.shape[-1]
        heads = base_heads if j == 0 else base_heads - (dim % base_heads)
        shift = ws // 2 if j % 2 == 1 else 0
        t = swin_block(sd, f"{p}swin{j + 1}.", cat, H, W, ws, shift, heads)
        t = F.conv2d(to_image(t, H, W), sd[f"{p}adjust{j + 1}.weight"], sd[f"{p}adjust{j + 1}.bias"])
        if j < 4:
            t = F.leaky_relu(t, 0.2)
        feats.append(to_tokens(t))
    return feats[5] * 0.2 + x


def drct_forward(sd, lr, ws=16, heads=6):
    """lr [B,3,H,W] with H,W multiples of ws -> (sr [B,3,4H,4W] un-clamped, feat = conv_after_body output)."""
    n_groups = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    mean = rgb_mean(lr)
    x = F.conv2d(lr - mean, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    B, C, H, W = x.shape
    t = F.layer_norm(to_tokens(x), (C,), sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"])
    for i in range(n_groups):
        t = rdg(sd, f"layers.{i}.", t, H, W, ws, heads)
    t = F.layer_norm(t, (C,), sd["norm.weight"], sd["norm.bias"])
    feat = F.conv2d(to_image(t, H, W), sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1)
    return sr_tail(sd, feat + x) + mean, feat
