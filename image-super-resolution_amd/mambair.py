"""MambaIR x4 on the HIP kernels (host side).  Mirrors src/models/mambair/mambair_arch.py: MambaIR.forward :662,
forward_features :651, ResidualGroup.forward :765, VSSBlock.forward :414, SS2D.forward :371 / forward_core :339.

SS2D re-design for the GPU: the x_proj einsum of all four directions is ONE token GEMM (the projection is
pointwise, the directions differ only in traversal order), the dt projection + softplus are fused into the
scan kernel, the four traversals are index maps inside the kernel, and y1+y2+y3+y4 -> out_norm -> * SiLU(z)
is one fused row kernel.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .common import CAB, SRTail, dev, to_map, tokens
from .ops import ACT_SILU


class _VSS:
    def __init__(self, sd, p, device):
        s = p + "self_attention."
        self.ln1 = (dev(sd[p + "ln_1.weight"], device), dev(sd[p + "ln_1.bias"], device))
        self.ln2 = (dev(sd[p + "ln_2.weight"], device), dev(sd[p + "ln_2.bias"], device))
        self.skip1 = dev(sd[p + "skip_scale"], device)
        self.skip2 = dev(sd[p + "skip_scale2"], device)
        self.in_proj = ops.pack_conv(sd[s + "in_proj.weight"], None, device)
        self.Dm = self.in_proj.N // 2
        # ln_1 + in_proj as one kernel (mambair_arch.py:417 + :238)
        C_ = sd[s + "in_proj.weight"].shape[1]
        self.in_proj_t = ops.pack_tok_gemm(sd[s + "in_proj.weight"], None, device, ln=(sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])) \
            if ops.tok_gemm_ok(C_, 2 * self.Dm) else None
        self.dw = ops.pack_dwconv(sd[s + "conv2d.weight"], sd[s + "conv2d.bias"], device)
        xw = sd[s + "x_proj_weight"].float()                                  # [4, R+2N, Dm]
        self.R = xw.shape[1] - 32
        self.x_proj = ops.pack_conv(xw.reshape(-1, xw.shape[2]), None, device)  # [4*(R+2N), Dm]
        # (FFSR_TOK_XPROJ=1: x_proj on the token GEMM -- measured 156 us against 158 us for the f32-input tile kernel: not adopted)
        xw2 = xw.reshape(-1, xw.shape[2])
        self.x_proj_t = ops.pack_tok_gemm(xw2, None, device) \
            if os.environ.get("FFSR_TOK_XPROJ", "0") == "1" and ops.tok_gemm_ok(xw2.shape[1], xw2.shape[0]) else None
        self.dtw = dev(sd[s + "dt_projs_weight"], device)                     # [4, Dm, R]
        self.dtb = dev(sd[s + "dt_projs_bias"], device)                       # [4, Dm]
        self.A = dev(-torch.exp(sd[s + "A_logs"].float()), device)            # [4*Dm, 16]
        self.Ds = dev(sd[s + "Ds"], device)
        self.out_norm = (dev(sd[s + "out_norm.weight"], device), dev(sd[s + "out_norm.bias"], device))
        self.out_proj = ops.pack_conv(sd[s + "out_proj.weight"], None, device)
        # out_norm + gate + out_proj + skip + ln_2 as one kernel (mambair_arch.py:381-386, :417-419)
        ow = sd[s + "out_proj.weight"]
        self.out_proj_t = ops.pack_tok_gemm(ow, None, device, check=False) if ops.tok_proj_ok(ow.shape[1], ow.shape[0]) else None
        self.cab = CAB(sd, p + "conv_blk.", device)

    def __call__(self, x, B, H, W):
        """x [P, C] tokens -> [P, C]"""
        Dm = self.Dm
        pl = ops.PLANES_AUTO and ops.GEMM_MODE == "bf16x3"
        if self.in_proj_t is not None and ops.tok_enabled():
            xz = ops.tok_gemm(x, self.in_proj_t)                              # [P, 2*Dm] = x | z
        else:
            n1 = ops.layernorm(x, *self.ln1, out_planes=True, want_f32=False) if pl else ops.layernorm(x, *self.ln1)
            xz = ops.linear(n1, self.in_proj)
        u = ops.dwconv2d(to_map(xz[:, :Dm], B, H, W), self.dw, act=ACT_SILU)  # [B,H,W,Dm]
        ut = tokens(u)
        if self.x_proj_t is not None and ops.tok_enabled():
            xdbl = ops.tok_gemm(ut, self.x_proj_t)                            # [P, 4*(R+32)]
        else:
            xdbl = ops.linear(ut, self.x_proj)
        fused_proj = pl and self.out_proj_t is not None and ops.tok_enabled() and ops.PROJ_FUSED
        y4 = ops.selective_scan4(ut, xdbl, self.dtw, self.dtb, self.A, self.Ds, B, H, W, Dm, self.R,
                                 pairs=False if fused_proj else None)
        if fused_proj:
            # (4-wave workgroups: the 12-k-step rows need ~440 registers, two waves per SIMD would spill)
            y, n2 = ops.tok_proj(y4[0], self.out_proj_t, xdirs=4, xstride=y4.stride(0), z=xz[:, Dm:], pro_ln=self.out_norm,
                                 res=x, rvec=self.skip1, post_ln=self.ln2, out_pre_ln=True, out_planes=True,
                                 waves=ops.TOK_WAVES or 4)
            c2, att = self.cab(n2.reshape_map(B, H, W))
            return ops.scale_add(y, tokens(c2), avec=self.skip2, bvec=att, rows_per_batch=H * W)
        g = ops.mamba_norm_gate(y4, xz[:, Dm:], *self.out_norm, out_planes=True if pl else None, want_f32=not pl)
        y = ops.linear(g, self.out_proj, res=x, rvec=self.skip1)              # x * skip_scale + out_proj(LN(sum y) * silu(z))
        if pl:
            c2, att = self.cab(ops.layernorm(y, *self.ln2, out_planes=True, want_f32=False).reshape_map(B, H, W))
        else:
            c2, att = self.cab(to_map(ops.layernorm(y, *self.ln2), B, H, W))
        return ops.scale_add(y, tokens(c2), avec=self.skip2, bvec=att, rows_per_batch=H * W)


class MambaIR:
    def __init__(self, sd, device):
        self.device = device
        depths = {}
        for k in sd:
            if k.startswith("layers.") and ".blocks." in k:
                parts = k.split(".")
                depths[int(parts[1])] = max(depths.get(int(parts[1]), 0), int(parts[4]) + 1)
        self.conv_first = ops.pack_conv(sd["conv_first.weight"], sd["conv_first.bias"], device)
        self.pe_norm = (dev(sd["patch_embed.norm.weight"], device), dev(sd["patch_embed.norm.bias"], device))
        self.norm = (dev(sd["norm.weight"], device), dev(sd["norm.bias"], device))
        self.groups = []
        for i in sorted(depths):
            blocks = [_VSS(sd, f"layers.{i}.residual_group.blocks.{j}.", device) for j in range(depths[i])]
            conv = ops.pack_conv(sd[f"layers.{i}.conv.weight"], sd[f"layers.{i}.conv.bias"], device)
            self.groups.append((blocks, conv))
        self.tail = SRTail(sd, device)

    def __call__(self, lr):
        """lr [B,H,W,3] -> (sr un-clamped [B,4H,4W,3], feat [B,H,W,embed])."""
        B, H, W, _ = lr.shape
        x0 = ops.conv2d(self.tail.center(lr), self.conv_first)
        t = ops.layernorm(tokens(x0), *self.pe_norm)
        for blocks, conv in self.groups:
            r = t
            for blk in blocks:
                r = blk(r, B, H, W)
            t = tokens(ops.conv2d(to_map(r, B, H, W), conv, res=to_map(t, B, H, W)))
        return self.tail(ops.final_norm(t, self.norm, B, H, W), x0)
