"""GRL-B x4 -- CPU oracle (test infrastructure).  Follows src/models/grl/grl_arch.py
(GRL.forward :506, forward_features :494, set_table_index_mask :386, TransformerStage.forward :168)
and grl/common/mixed_attn_block_efficient.py (AffineTransform :23, Attention.attn :77,
WindowAttention :97, AnchorStripeAttention :177, MixedAttention.forward :348, block forward :537),
mixed_attn_block.py (CPB_MLP :24, AnchorLinear :714, CAB :970), ops.py (tables/indices :170-375).
Configuration of create_grl_model (grl/__init__.py:74): dim 180, window 8, stripe [8,8] with no
groups and no stripe shift, anchor down-factor 2 (avg-pool), 3+3 heads, mlp_ratio 2, local CAB.
"""
import math
import torch
import torch.nn.functional as F
from .common import (rgb_mean, to_tokens, to_image, win_split, win_merge, shift_mask, sr_tail, cab)

WS = 8      # square window and stripe edge
DF = 2      # anchor down factor
HEADS = 3


def coords_table(ws, aws):
    """Log-spaced relative coordinate table (ops.py:214-252) for window edge ws vs anchor edge aws."""
    hi = ws - 1 - (ws - aws) // 2
    lo = -(aws - 1) - (ws - aws) // 2
    c = torch.arange(lo, hi + 1, dtype=torch.float32)
    t = torch.stack(torch.meshgrid(c, c, indexing="ij"), -1) / hi * 8
    return torch.sign(t) * torch.log2(t.abs() + 1.0) / math.log2(8)     # [n, n, 2]


def rel_index(ws, aws, window_to_anchor):
    """ops.py:352-375 (anchor grid starts at (0,0))."""
    def grid(n):
        return torch.stack(torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")).flatten(1)
    cw, ca = grid(ws), grid(aws)
    span = aws + ws - 1
    if window_to_anchor:
        d = (cw[:, :, None] - ca[:, None, :]) + (aws - 1)
    else:
        d = (ca[:, :, None] - cw[:, None, :]) + (ws - 1)
    return d[0] * span + d[1]


def cpb_bias(sd, p, table, index):
    """16*sigmoid(MLP(table))[index] -> [heads, N1, N2]  (AffineTransform, input independent)."""
    h = F.relu(F.linear(table.reshape(-1, 2), sd[p + "cpb_mlp.0.weight"], sd[p + "cpb_mlp.0.bias"]))
    tab = F.linear(h, sd[p + "cpb_mlp.2.weight"])                        # [n*n, heads]
    b = tab[index.reshape(-1)].reshape(index.shape[0], index.shape[1], -1).permute(2, 0, 1)
    return 16 * torch.sigmoid(b)


def cos_attn(sd, p, q, k, v, table, index, mask=None):
    """q [B_,h,N1,d], k [B_,h,N2,d], v [B_,h,N2,d] -> [B_,h,N1,d]."""
    a = F.normalize(q, dim=-1) @ F.normalize(k, dim=-1).transpose(-2, -1)
    a = a * torch.clamp(sd[p + "logit_scale"], max=math.log(100.0)).exp()
    a = a + cpb_bias(sd, p, table, index)[None]
    if mask is not None:
        nW = mask.shape[0]
        a = (a.reshape(-1, nW, *a.shape[1:]) + mask[None, :, None]).reshape(-1, *a.shape[1:])
    return a.softmax(-1) @ v


def mixed_attention(sd, p, x, H, W, window_shift):
    B, L, C = x.shape
    hd = C // 2 // HEADS
    qkv = F.linear(x, sd[p + "qkv.body.weight"], sd[p + "qkv.body.bias"])
    qkv_w, qkv_s = qkv.split(C * 3 // 2, dim=-1)
    # anchors: 2x2 average pool then Linear C -> C/2 (AnchorLinear)
    pooled = F.avg_pool2d(to_image(x, H, W), DF, DF)
    anchor = F.linear(to_tokens(pooled), sd[p + "anchor.body.0.reduction.weight"],
                      sd[p + "anchor.body.0.reduction.bias"]).reshape(B, H // DF, W // DF, C // 2)
    # --- square (shifted) window branch
    t = qkv_w.reshape(B, H, W, -1)
    sh = WS // 2 if window_shift else 0
    if sh:
        t = torch.roll(t, (-sh, -sh), (1, 2))
    w = win_split(t, WS, WS).reshape(-1, WS * WS, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
    mask = shift_mask(H, W, WS, WS, sh, sh) if sh else None
    o = cos_attn(sd, p + "window_attn.attn_transform.", w[0], w[1], w[2],
                 coords_table(WS, WS), rel_index(WS, WS, True), mask)
    o = win_merge(o.transpose(1, 2).reshape(-1, WS * WS, C // 2), WS, WS, H, W)
    if sh:
        o = torch.roll(o, (sh, sh), (1, 2))
    x_win = o.reshape(B, L, C // 2)
    # --- anchored stripe branch (two hops: anchors gather from the stripe, tokens read anchors)
    AW = WS // DF
    s = win_split(qkv_s.reshape(B, H, W, -1), WS, WS).reshape(-1, WS * WS, 3, HEADS, hd).permute(2, 0, 3, 1, 4)
    a = win_split(anchor, AW, AW).reshape(-1, AW * AW, HEADS, hd).permute(0, 2, 1, 3)
    tab = coords_table(WS, AW)
    g = cos_attn(sd, p + "stripe_attn.attn_transform1.", a, s[1], s[2], tab, rel_index(WS, AW, False))
    o = cos_attn(sd, p + "stripe_attn.attn_transform2.", s[0], a, g, tab, rel_index(WS, AW, True))
    x_str = win_merge(o.transpose(1, 2).reshape(-1, WS * WS, C // 2), WS, WS, H, W).reshape(B, L, C // 2)
    return F.linear(torch.cat([x_win, x_str], -1), sd[p + "proj.weight"], sd[p + "proj.bias"])


def grl_block(sd, p, x, H, W, window_shift):
    C = x.shape[-1]
    a = mixed_attention(sd, p + "attn.", x, H, W, window_shift)
    local = to_tokens(cab(sd, p + "conv.", to_image(x, H, W)))
    x = x + F.layer_norm(a, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"]) + local
    m = F.linear(F.gelu(F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])),
                 sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + F.layer_norm(m, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"])


def grl_forward(sd, lr):
    """lr [B,3,H,W] (H,W multiples of 8) -> (sr un-clamped, feat = conv_after_body output)."""
    depths = {}
    for k in sd:
        if k.startswith("layers.") and ".blocks." in k:
            _, i, _, j = k.split(".")[:4]
            depths[int(i)] = max(depths.get(int(i), 0), int(j) + 1)
    mean = rgb_mean(lr)
    x = F.conv2d(lr - mean, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    B, C, H, W = x.shape
    t = F.layer_norm(to_tokens(x), (C,), sd["norm_start.weight"], sd["norm_start.bias"])
    for i in sorted(depths):
        r = t
        for j in range(depths[i]):
            r = grl_block(sd, f"layers.{i}.blocks.{j}.", r, H, W, window_shift=(j % 2 == 0))
        r = F.conv2d(to_image(r, H, W), sd[f"layers.{i}.conv.weight"], sd[f"layers.{i}.conv.bias"], padding=1)
        t = to_tokens(r) + t
    t = F.layer_norm(t, (C,), sd["norm_end.weight"], sd["norm_end.bias"])
    feat = F.conv2d(to_image(t, H, W), sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1)
    return sr_tail(sd, feat + x) + mean, feat
