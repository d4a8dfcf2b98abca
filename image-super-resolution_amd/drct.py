"""DRCT-L x4 on the HIP kernels (host side).  Mirrors src/models/drct/drct_arch.py: DRCT.forward :777,
forward_features :761, RDG.forward :292 (dense concat kept in ONE [pixels, 308] buffer, every block writes its
32 new channels in place), SwinTransformerBlock.forward :376 (roll / partition / mask folded into the
attention kernel's addressing), WindowAttention.forward :175.
"""
from __future__ import annotations

import torch

from . import ops
from .common import SRTail, dev, to_map, tokens
from .ops import ACT_GELU, ACT_LRELU

import os

TAIL_FUSED = os.environ.get("FFSR_TOK_TAIL", "1") != "0"   # FFSR_TOK_TAIL=0: adjust convolutions as their own launches (A/B runs)


class _Swin:
    def __init__(self, sd, p, device, dim, heads, ws, shift):
        self.dim, self.heads, self.ws, self.shift = dim, heads, ws, shift
        self.n1 = (dev(sd[p + "norm1.weight"], device), dev(sd[p + "norm1.bias"], device))
        self.n2 = (dev(sd[p + "norm2.weight"], device), dev(sd[p + "norm2.bias"], device))
        self.qkv = ops.pack_conv(sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], device)
        self.proj = ops.pack_conv(sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], device)
        self.fc1 = ops.pack_conv(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], device)
        self.fc2 = ops.pack_conv(sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], device)
        self.bias = dev(sd[p + "attn.relative_position_bias_table"], device)     # [(2ws-1)^2, heads], gathered in-kernel
        self.scale = (dim // heads) ** -0.5
        # x + mlp(norm2(x)) as ONE kernel (ffsr_tok_chain_f32: LayerNorm, fc1, GELU, fc2 and the residual; the hidden layer
        # stays in registers).  drct_arch.py:77-95, :405-407
        # norm1 + qkv as one kernel (LayerNorm in the GEMM's prologue, drct_arch.py:385-388 + :166)
        self.qkv_t = ops.pack_tok_gemm(sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], device,
                                       ln=(sd[p + "norm1.weight"], sd[p + "norm1.bias"])) if ops.tok_gemm_ok(dim, 3 * dim) else None
        self.mlp = ops.pack_tok_chain(sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], sd[p + "mlp.fc2.weight"],
                                      sd[p + "mlp.fc2.bias"], device, mode=0, ln=(sd[p + "norm2.weight"], sd[p + "norm2.bias"])) \
            if ops.tok_chain_ok(dim, dim, 0) else None
        # ... and the attention's output projection + residual in front of it (drct_arch.py:400-404): x1 = x + proj(attn) only
        # ever exists in the kernel's registers
        self.proj_t = ops.pack_tok_gemm(sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], device) \
            if self.mlp is not None and ops.tok_gemm_ok(dim, dim) and ops.tok_head_ok(dim, 0) else None

    def __call__(self, x, B, H, W, tail=None):
        """x [P, dim] (row stride may be wider) -> [P, dim]; tail (see ops.tok_chain): the dense block's adjust convolution rides
        in the MLP kernel and the block's own output is not stored -> returns None"""
        pl = ops.PLANES_AUTO and ops.GEMM_MODE == "bf16x3"   # LN / fc1 results go to the GEMMs as bf16 hi/lo planes
        if self.qkv_t is not None and ops.tok_enabled():
            qkv = ops.tok_gemm(x, self.qkv_t)
        else:
            n = ops.layernorm(x, *self.n1, out_planes=True, want_f32=False) if pl else ops.layernorm(x, *self.n1)
            qkv = ops.linear(n, self.qkv)
        a = ops.window_attn(qkv, self.bias, B, H, W, self.dim, self.heads, self.ws, self.shift, self.scale)
        if self.proj_t is not None and ops.tok_enabled() and ops.HEAD_FUSED:
            if tail is not None:
                ops.tok_head_chain(a, self.proj_t, self.mlp, hres=x, tail=tail, want_f32=False)
                return None
            return ops.tok_head_chain(a, self.proj_t, self.mlp, hres=x)
        y = ops.linear(a, self.proj, res=x)
        if self.mlp is not None and ops.tok_enabled():
            if tail is not None:
                ops.tok_chain(y, self.mlp, res=y, tail=tail, want_f32=False)
                return None
            return ops.tok_chain(y, self.mlp, res=y)
        if not pl:
            h = ops.linear(ops.layernorm(y, *self.n2), self.fc1, act=ACT_GELU)
            return ops.linear(h, self.fc2, res=y)
        h = ops.linear(ops.layernorm(y, *self.n2, out_planes=True, want_f32=False), self.fc1, act=ACT_GELU,
                       out_planes=True, want_f32=False)
        return ops.linear(h, self.fc2, res=y)


class DRCT:
    def __init__(self, sd, device, ws=16, heads=6, gc=32):
        self.device, self.ws, self.gc = device, ws, gc
        self.embed = sd["conv_first.weight"].shape[0]
        n_groups = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
        self.conv_first = ops.pack_conv(sd["conv_first.weight"], sd["conv_first.bias"], device)
        self.pe_norm = (dev(sd["patch_embed.norm.weight"], device), dev(sd["patch_embed.norm.bias"], device))
        self.norm = (dev(sd["norm.weight"], device), dev(sd["norm.bias"], device))
        self.groups = []
        for i in range(n_groups):
            blocks = []
            for j in range(5):
                dim = self.embed + gc * j
                h = heads if j == 0 else heads - (dim % heads)
                sw = _Swin(sd, f"layers.{i}.swin{j + 1}.", device, dim, h, ws, ws // 2 if j % 2 else 0)
                adj = ops.pack_conv(sd[f"layers.{i}.adjust{j + 1}.weight"], sd[f"layers.{i}.adjust{j + 1}.bias"], device)
                adj_t = ops.pack_tok_gemm(sd[f"layers.{i}.adjust{j + 1}.weight"], sd[f"layers.{i}.adjust{j + 1}.bias"], device) \
                    if ops.tok_gemm_ok(dim, gc if j < 4 else self.embed) else None
                blocks.append((sw, adj, adj_t))
            self.groups.append(blocks)
        self.tail = SRTail(sd, device)

    def __call__(self, lr):
        """lr [B,H,W,3] (H, W multiples of the window) -> (sr [B,4H,4W,3] un-clamped, feat [B,H,W,embed])."""
        B, H, W, _ = lr.shape
        if H % self.ws or W % self.ws:
            raise ValueError(f"DRCT input {H}x{W} must be a multiple of the window size {self.ws}")
        E, gc, P = self.embed, self.gc, B * H * W
        x0 = ops.conv2d(self.tail.center(lr), self.conv_first)
        wide = E + 4 * gc
        cat = [torch.empty(P, wide, device=lr.device), torch.empty(P, wide, device=lr.device)]
        ops.layernorm(tokens(x0), *self.pe_norm, out=cat[0][:, :E])
        cur = 0
        for blocks in self.groups:
            buf = cat[cur]
            for j, (sw, adj, adj_t) in enumerate(blocks):
                dim = E + gc * j
                # the adjust convolution of the dense block as the tail of the Swin block's MLP kernel: the block output itself
                # (only consumed by adjust) never reaches HBM.  adjust1-4: LeakyReLU(0.2) -> the 32 new channels of the
                # concatenation; adjust5: x5 * 0.2 + x -> the next group's buffer  (drct_arch.py:292-301)
                if adj_t is not None and sw.mlp is not None and ops.tok_enabled() and TAIL_FUSED:
                    if j < 4:
                        sw(buf[:, :dim], B, H, W, tail=dict(tg=adj_t, out=buf[:, dim:dim + gc], act=ACT_LRELU, slope=0.2))
                    else:
                        sw(buf[:, :dim], B, H, W, tail=dict(tg=adj_t, out=cat[1 - cur][:, :E], cscale=0.2, res=buf[:, :E]))
                    continue
                y = sw(buf[:, :dim], B, H, W)
                if j < 4 and adj_t is not None and ops.tok_enabled():
                    ops.tok_gemm(y, adj_t, act=ACT_LRELU, slope=0.2, out=buf[:, dim:dim + gc])
                elif j < 4:
                    ops.linear(y, adj, act=ACT_LRELU, slope=0.2, out=buf[:, dim:dim + gc])
                else:
                    ops.linear(y, adj, cscale=0.2, res=buf[:, :E], out=cat[1 - cur][:, :E])     # x5 * 0.2 + x
            cur = 1 - cur
        return self.tail(ops.final_norm(cat[cur][:, :E], self.norm, B, H, W), x0)
