"""Pieces shared by the Swin-family experts (host side): packed SR tail, CAB with channel attention."""
from __future__ import annotations

import torch

from . import ops
from .ops import ACT_GELU, ACT_LRELU, ACT_RELU, ACT_SIGMOID

RGB_MEAN = (0.4488, 0.4371, 0.4040)   # drct_arch.py:668, grl_arch.py:266, mambair_arch.py:548


def dev(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().float().contiguous().to(device)


def tokens(x: torch.Tensor) -> torch.Tensor:
    """[B,H,W,C] map -> [B*H*W, C] matrix view (same memory)."""
    s = ops.ld(x)
    return x.as_strided((ops.rows(x), x.shape[3]), (s, 1), x.storage_offset())


def to_map(t: torch.Tensor, B, H, W) -> torch.Tensor:
    s = t.stride(0)
    return t.as_strided((B, H, W, t.shape[1]), (H * W * s, W * s, s, 1), t.storage_offset())


class SRTail:
    """conv_after_body (+ hook output) -> +skip -> conv_before_upsample + LeakyReLU(0.01) -> 2x(conv, PixelShuffle 2)
    -> conv_last (+ RGB mean folded into its bias).  drct_arch.py:779-789, grl_arch.py:512-517, mambair_arch.py:664-669."""

    def __init__(self, sd, device, up_prefix="upsample."):
        C = sd["conv_after_body.weight"].shape[0]
        self.after_body = ops.pack_conv(sd["conv_after_body.weight"], sd["conv_after_body.bias"], device)
        self.before_up = ops.pack_conv(sd["conv_before_upsample.0.weight"], sd["conv_before_upsample.0.bias"], device)
        self.up0 = ops.pack_conv(sd[up_prefix + "0.weight"], sd[up_prefix + "0.bias"], device)
        self.up2 = ops.pack_conv(sd[up_prefix + "2.weight"], sd[up_prefix + "2.bias"], device)
        mean = torch.tensor(RGB_MEAN)
        self.last = ops.pack_conv(sd["conv_last.weight"], sd["conv_last.bias"].float() + mean.to(sd["conv_last.bias"].device), device)
        self.neg_mean = dev(-mean, device)

    def center(self, lr):
        """lr [B,H,W,3] -> lr - mean (pad channel stays 0)."""
        return ops.unary(lr, cbias=self.neg_mean)

    def __call__(self, body_tokens_map, x0):
        """body map [B,H,W,C] (after the final norm), x0 = conv_first output -> (sr [B,4H,4W,3], feat [B,H,W,C])."""
        feat = ops.conv2d(body_tokens_map, self.after_body)
        x = ops.scale_add(feat, x0)
        x = ops.conv2d(x, self.before_up, act=ACT_LRELU, slope=0.01)
        x = ops.conv2d(x, self.up0, shuffle=2)
        x = ops.conv2d(x, self.up2, shuffle=2)
        return ops.conv2d(x, self.last), feat


class CAB:
    """conv3x3 -> GELU -> conv3x3 -> RCAN channel attention (mambair_arch.py:41-66; grl mixed_attn_block.py:942-983).
    ``__call__`` returns (conv output c2, attention vector [B, C]); the caller fuses ``c2 * att`` into its residual."""

    def __init__(self, sd, p, device):
        self.c0 = ops.pack_conv(sd[p + "cab.0.weight"], sd[p + "cab.0.bias"], device)
        mid = self.c0.N
        self.c2 = ops.pack_conv(sd[p + "cab.2.weight"], sd[p + "cab.2.bias"], device, cin_pad=ops.pad4(mid))
        self.a1 = ops.pack_conv(sd[p + "cab.3.attention.1.weight"], sd[p + "cab.3.attention.1.bias"], device)
        sq = self.a1.N
        self.a3 = ops.pack_conv(sd[p + "cab.3.attention.3.weight"], sd[p + "cab.3.attention.3.bias"], device,
                                cin_pad=ops.pad4(sq))

    def __call__(self, x):
        if isinstance(x, ops.Planes):      # conv -> GELU -> conv without an fp32 round trip
            c2 = ops.conv2d(ops.conv2d(x, self.c0, act=ACT_GELU, out_planes=True, want_f32=False), self.c2)
        else:
            c1 = ops.conv2d(x, self.c0, act=ACT_GELU)
            c2 = ops.conv2d(ops.widen(c1, self.c2.Cin), self.c2)
        if c2.shape[3] <= 1024 and self.a1.N <= 64:
            att = ops.channel_attention(c2, self.a1, self.a3)                 # pool + FC + ReLU + FC + sigmoid, 2 launches
        else:
            pooled = ops.colmean(c2)                                           # [B, C]
            s = ops.linear(pooled, self.a1, act=ACT_RELU)                      # [B, C/r] (zero padded to x4)
            s = s.as_strided((s.shape[0], self.a3.Cin), (s.stride(0), 1), s.storage_offset())
            att = ops.linear(s, self.a3, act=ACT_SIGMOID)                      # [B, C]
        return c2, att
