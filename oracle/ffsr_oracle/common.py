"""Shared pieces of the Swin-family experts -- CPU oracle (test infrastructure)."""
import torch
import torch.nn.functional as F

RGB_MEAN = (0.4488, 0.4371, 0.4040)  # drct_arch.py:668, grl_arch.py:266, mambair_arch.py:548


def rgb_mean(x):
    return torch.tensor(RGB_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)


def to_tokens(x):          # [B,C,H,W] -> [B,HW,C]
    return x.flatten(2).transpose(1, 2)


def to_image(t, H, W):     # [B,HW,C] -> [B,C,H,W]
    return t.transpose(1, 2).reshape(t.shape[0], -1, H, W)


def win_split(x, wh, ww):  # [B,H,W,C] -> [B*nW, wh*ww, C]
    B, H, W, C = x.shape
    x = x.reshape(B, H // wh, wh, W // ww, ww, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, wh * ww, C)


def win_merge(w, wh, ww, H, W):  # inverse of win_split -> [B,H,W,C]
    C = w.shape[-1]
    B = w.shape[0] // ((H // wh) * (W // ww))
    x = w.reshape(B, H // wh, W // ww, wh, ww, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, C)


def shift_region_ids(H, W, wh, ww, sh, sw):
    """Region labels of the shifted-window mask (drct_arch.py:353-371, grl ops.py:76-99): the image
    is cut into 3x3 regions by the python slices [0,-w), [-w,-s), [-s,end)."""
    ids = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -wh), slice(-wh, -sh), slice(-sh, None)):
        for ws in (slice(0, -ww), slice(-ww, -sw), slice(-sw, None)):
            ids[hs, ws] = cnt
            cnt += 1
    return ids


def shift_mask(H, W, wh, ww, sh, sw):
    """[nW, wh*ww, wh*ww] additive mask: 0 where the two tokens share a region, -100 otherwise."""
    ids = win_split(shift_region_ids(H, W, wh, ww, sh, sw)[None, :, :, None], wh, ww)[..., 0]
    diff = ids[:, None, :] - ids[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def sr_tail(sd, x, slope=0.01):
    """conv_before_upsample(+LeakyReLU 0.01) -> 2x(conv 64->256, PixelShuffle 2) -> conv_last
    (drct_arch.py:738-743,786-788).  Key names differ between experts -> resolved by probing."""
    x = F.leaky_relu(F.conv2d(x, sd["conv_before_upsample.0.weight"],
                              sd["conv_before_upsample.0.bias"], padding=1), slope)
    up = "upsample.up." if "upsample.up.0.weight" in sd else "upsample."   # GRL wraps in .up
    for i in (0, 2):
        x = F.pixel_shuffle(F.conv2d(x, sd[f"{up}{i}.weight"], sd[f"{up}{i}.bias"], padding=1), 2)
    return F.conv2d(x, sd["conv_last.weight"], sd["conv_last.bias"], padding=1)


def channel_attention(sd, p, x):
    """RCAN channel attention: x * sigmoid(W2 relu(W1 avgpool(x)))  (mambair_arch.py:20-38)."""
    y = x.mean((2, 3), keepdim=True)
    y = F.relu(F.conv2d(y, sd[p + "attention.1.weight"], sd[p + "attention.1.bias"]))
    y = torch.sigmoid(F.conv2d(y, sd[p + "attention.3.weight"], sd[p + "attention.3.bias"]))
    return x * y


def cab(sd, p, x):
    """conv3x3 -> GELU -> conv3x3 -> channel attention (mambair_arch.py:41-66, mixed_attn_block.py:970)."""
    x = F.gelu(F.conv2d(x, sd[p + "cab.0.weight"], sd[p + "cab.0.bias"], padding=1))
    x = F.conv2d(x, sd[p + "cab.2.weight"], sd[p + "cab.2.bias"], padding=1)
    return channel_attention(sd, p + "cab.3.", x)
