"""NAFNet-SIDD-width64 as an x4 SR expert on the HIP kernels (host side).

Mirrors src/models/nafnet/__init__.py:NAFNetSR.forward :117 (bicubic x4 -> U-Net at HR -> clamp) and
nafnet_arch.py (NAFBlock.forward :110, NAFNet.forward :195).  ``sd`` = the inner NAFNet state_dict
(official checkpoint keys).  Returns the SR image and the feature entering ``ending`` (the tensor the
reference's forward hook captures, expert_loader.py:553-558).
"""
from __future__ import annotations

import os

import torch

from . import ops
from .common import dev, to_map, tokens


CONV1_TOK_MAXC = int(os.environ.get("FFSR_NAF_CONV1_TOK_MAXC", "256"))   # norm1 + conv1 on the token GEMM up to this width


class _Block:
    def __init__(self, sd, p, device):
        c = sd[p + "conv3.weight"].shape[0]
        self.c = c
        self.n1 = (dev(sd[p + "norm1.weight"], device), dev(sd[p + "norm1.bias"], device))
        self.n2 = (dev(sd[p + "norm2.weight"], device), dev(sd[p + "norm2.bias"], device))
        self.conv1 = ops.pack_conv(sd[p + "conv1.weight"], sd[p + "conv1.bias"], device)
        self.dw = ops.pack_dwconv(sd[p + "conv2.weight"], sd[p + "conv2.bias"], device)
        self.conv3 = ops.pack_conv(sd[p + "conv3.weight"], sd[p + "conv3.bias"], device)
        self.sca = ops.pack_conv(sd[p + "sca.1.weight"], sd[p + "sca.1.bias"], device)
        self.conv4 = ops.pack_conv(sd[p + "conv4.weight"], sd[p + "conv4.bias"], device)
        # shallow levels: SimpleGate rides in conv4's store (weight rows packed as interleaved 32-channel pairs)
        self.conv4g = ops.pack_conv(sd[p + "conv4.weight"], sd[p + "conv4.bias"], device, gate_pairs=True) if (c < 256 and c % 32 == 0) else None
        self.conv5 = ops.pack_conv(sd[p + "conv5.weight"], sd[p + "conv5.bias"], device)
        self.beta = dev(sd[p + "beta"].reshape(-1), device)
        self.gamma = dev(sd[p + "gamma"].reshape(-1), device)
        # second half of the block, y + gamma * conv5(SimpleGate(conv4(norm2(y)))) (nafnet_arch.py:125-131), as ONE kernel at the
        # shallow levels (c = 64, 128: ffsr_tok_chain_f32 mode 1).  gamma is folded into conv5 (diag(gamma) W5, gamma * b5), so the
        # residual is the input row the kernel holds anyway: one read and one write of the map instead of seven passes.
        # first half: norm1 + conv1 as one kernel (nafnet_arch.py:113-115)
        self.conv1_t = ops.pack_tok_gemm(sd[p + "conv1.weight"], sd[p + "conv1.bias"], device,
                                         ln=(sd[p + "norm1.weight"], sd[p + "norm1.bias"]), eps=1e-6) \
            if (c <= CONV1_TOK_MAXC and ops.tok_gemm_ok(c, 2 * c)) else None
        # norm2 + conv4 likewise where the gated half is not one kernel (c = 256: LayerNorm 61 us + tile GEMM 237 us -> 204 us)
        self.conv4_t = ops.pack_tok_gemm(sd[p + "conv4.weight"], sd[p + "conv4.bias"], device,
                                         ln=(sd[p + "norm2.weight"], sd[p + "norm2.bias"]), eps=1e-6) \
            if (128 < c <= CONV1_TOK_MAXC and ops.tok_gemm_ok(c, 2 * c)) else None
        self.ffn = None
        if ops.tok_chain_ok(c, c, 1):
            gam = sd[p + "gamma"].detach().float().reshape(-1)
            w5 = sd[p + "conv5.weight"].detach().float().reshape(c, c)
            self.ffn = ops.pack_tok_chain(sd[p + "conv4.weight"], sd[p + "conv4.bias"], w5 * gam[:, None].to(w5.device),
                                          sd[p + "conv5.bias"].detach().float() * gam.to(w5.device), device, mode=1,
                                          ln=(sd[p + "norm2.weight"], sd[p + "norm2.bias"]), eps=1e-6)

        # ... and conv3 (with the channel attention on its input and beta on its output) as the HEAD of that kernel:
        # y = x + beta * conv3(gated * sca) stays in registers (nafnet_arch.py:122-125)
        self.conv3_t = None
        # (c = 128 only: at c = 64 the pass is pure streaming -- measured 854 us fused against 413 + 420 us for conv3 and the
        #  gated half as two launches at 1408 x 2048, and slower still at 1024 x 1024; at c = 128 546 against 629 us)
        if self.ffn is not None and c == 128 and ops.tok_gemm_ok(c, c) and ops.tok_head_ok(c, 1):
            bet = sd[p + "beta"].detach().float().reshape(-1)
            w3 = sd[p + "conv3.weight"].detach().float().reshape(c, c)
            self.conv3_t = ops.pack_tok_gemm(w3 * bet[:, None].to(w3.device), sd[p + "conv3.bias"].detach().float() * bet.to(w3.device),
                                             device)

    def __call__(self, x):
        c = self.c
        # deep levels (c >= 256: long K, few pixels) are MFMA-heavy: their LayerNorm outputs go to the GEMM as bf16
        # hi/lo planes; the shallow levels are HBM-bound either way and keep the fp32 path
        pl = ops.PLANES_AUTO and ops.GEMM_MODE == "bf16x3" and c >= 256
        if self.conv1_t is not None and ops.tok_enabled() and ops.rows(x) > 64 * 24:
            Bx, Hx, Wx, _ = x.shape
            t = to_map(ops.tok_gemm(tokens(x), self.conv1_t), Bx, Hx, Wx)
        else:
            t = ops.layernorm(x, *self.n1, eps=1e-6, out_planes=True if pl else None, want_f32=not pl)
            t = ops.conv2d(t, self.conv1)
        g, pooled = ops.dw3x3_gate_pool(t, self.dw)              # SimpleGate + global average pool
        sca = ops.linear(pooled, self.sca)                       # [B, c] channel attention
        if self.conv3_t is not None and ops.tok_enabled() and ops.HEAD_FUSED and ops.rows(x) > 64 * 24:
            B, H, W, _ = x.shape
            return to_map(ops.tok_head_chain(tokens(g), self.conv3_t, self.ffn, in_scale=sca.contiguous(), hres=tokens(x),
                                             rows_per_batch=H * W), B, H, W)
        y = ops.conv2d(g, self.conv3, akscale=sca.contiguous(), res=x, cvec=self.beta)   # x + conv3(g*sca)*beta
        B, H, W, _ = y.shape
        if self.ffn is not None and ops.tok_enabled() and B * H * W > 64 * 24:
            yt = tokens(y)
            return to_map(ops.tok_chain(yt, self.ffn, res=yt), B, H, W)
        if self.conv4_t is not None and ops.tok_enabled() and B * H * W > 64 * 24:
            t = to_map(ops.tok_gemm(tokens(y), self.conv4_t), B, H, W)
            g = ops.mul_add(t[..., :c], t[..., c:])              # SimpleGate
            return ops.conv2d(g, self.conv5, res=y, cvec=self.gamma)
        t = ops.layernorm(y, *self.n2, eps=1e-6, out_planes=True if pl else None, want_f32=not pl)
        if self.conv4g is not None and ops.GEMM_MODE == "bf16x3" and ops.GATE_FUSED and B * H * W > 64 * 24:
            g = ops.conv2d(t, self.conv4g, gate=True)            # conv4 + SimpleGate in one kernel
        else:
            t = ops.conv2d(t, self.conv4)
            g = ops.mul_add(t[..., :c], t[..., c:])              # SimpleGate
        return ops.conv2d(g, self.conv5, res=y, cvec=self.gamma)


class NAFNetSR:
    def __init__(self, sd, device, scale=4):
        self.scale = scale
        self.device = device
        n_enc = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("encoders."))
        n_dec = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("decoders."))

        def blocks(prefix):
            idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)})
            return [_Block(sd, f"{prefix}{j}.", device) for j in idx]

        self.intro = ops.pack_conv(sd["intro.weight"], sd["intro.bias"], device)
        self.ending = ops.pack_conv(sd["ending.weight"], sd["ending.bias"], device)
        self.enc = [blocks(f"encoders.{i}.") for i in range(n_enc)]
        self.downs = [ops.pack_conv(sd[f"downs.{i}.weight"], sd[f"downs.{i}.bias"], device, stride=2, pad=0)
                      for i in range(n_enc)]
        self.mid = blocks("middle_blks.")
        self.ups = [ops.pack_conv(sd[f"ups.{i}.0.weight"], None, device) for i in range(n_dec)]
        self.dec = [blocks(f"decoders.{i}.") for i in range(n_dec)]
        self.mult = 2 ** n_enc

    def __call__(self, lr):
        """lr [B,h,w,3] (ld 4) -> (sr [B,4h,4w,3] clamped to [0,1], feat [B,4h,4w,width])."""
        B, h, w, _ = lr.shape
        H, W = h * self.scale, w * self.scale
        if H % self.mult or W % self.mult:
            raise ValueError(f"NAFNet input {H}x{W} must be a multiple of {self.mult} (io.main pads LR to 16)")
        up = ops.bicubic_up(lr, self.scale)
        x = ops.conv2d(up, self.intro)
        skips = []
        for blks, down in zip(self.enc, self.downs):
            for blk in blks:
                x = blk(x)
            skips.append(x)
            x = ops.conv2d(x, down)
        for blk in self.mid:
            x = blk(x)
        for blks, upc, skip in zip(self.dec, self.ups, reversed(skips)):
            x = ops.conv2d(x, upc, shuffle=2, res=skip)           # 1x1 conv + PixelShuffle(2) + skip
            for blk in blks:
                x = blk(x)
        feat = x
        out = ops.conv2d(x, self.ending, res=up)
        return ops.unary(out, clamp=(0.0, 1.0), out=out), feat
