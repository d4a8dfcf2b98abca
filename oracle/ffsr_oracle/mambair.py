"""MambaIR x4 -- CPU oracle (test infrastructure).  Follows src/models/mambair/mambair_arch.py:
SS2D.forward_core :339, SS2D.forward :371, VSSBlock.forward :414, ResidualGroup.forward :765,
MambaIR.forward_features :651, MambaIR.forward :662, with io.py:156-163's kwargs (dim 180, 6x6
blocks, d_state 16, expand 2, dt_rank 12).  The scan itself is scan.selective_scan_ref
(parity unpinned -- see that file).
"""
import torch
import torch.nn.functional as F
from .common import rgb_mean, to_tokens, to_image, sr_tail, cab
from .scan import selective_scan_ref


def ss2d(sd, p, x, scan_fn=selective_scan_ref):
    """x [B,H,W,C] -> [B,H,W,C]: in_proj -> (dw3x3+SiLU -> 4-direction selective scan -> LN) * SiLU(z) -> out_proj."""
    B, H, W, C = x.shape
    L = H * W
    xz = F.linear(x, sd[p + "in_proj.weight"])
    u, z = xz.chunk(2, dim=-1)
    Dm = u.shape[-1]
    u = F.silu(F.conv2d(u.permute(0, 3, 1, 2), sd[p + "conv2d.weight"], sd[p + "conv2d.bias"],
                        padding=1, groups=Dm))                                        # [B,Dm,H,W]
    xw = sd[p + "x_proj_weight"]                 # [4, R+2N, Dm]
    dtw = sd[p + "dt_projs_weight"]              # [4, Dm, R]
    N = sd[p + "A_logs"].shape[1]
    R = dtw.shape[2]
    # four traversals of the same map: row-major, column-major, and both reversed
    rows = u.reshape(B, Dm, L)
    cols = u.transpose(2, 3).reshape(B, Dm, L)
    xs = torch.stack([rows, cols, rows.flip(-1), cols.flip(-1)], 1)                   # [B,4,Dm,L]
    proj = torch.einsum("bkdl,kcd->bkcl", xs, xw)
    dts, Bs, Cs = proj.split([R, N, N], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dtw)
    y = scan_fn(xs.reshape(B, 4 * Dm, L), dts.reshape(B, 4 * Dm, L).contiguous(),
                -torch.exp(sd[p + "A_logs"].float()), Bs.contiguous(), Cs.contiguous(),
                sd[p + "Ds"].float(), z=None, delta_bias=sd[p + "dt_projs_bias"].reshape(-1).float(),
                delta_softplus=True, return_last_state=False).reshape(B, 4, Dm, L)
    y_rows = y[:, 0] + y[:, 2].flip(-1)
    y_cols = (y[:, 1] + y[:, 3].flip(-1)).reshape(B, Dm, W, H).transpose(2, 3).reshape(B, Dm, L)
    y = (y_rows + y_cols).transpose(1, 2).reshape(B, H, W, Dm)
    y = F.layer_norm(y, (Dm,), sd[p + "out_norm.weight"], sd[p + "out_norm.bias"])
    return F.linear(y * F.silu(z), sd[p + "out_proj.weight"])


def vss_block(sd, p, t, H, W, scan_fn=selective_scan_ref):
    B, L, C = t.shape
    x = t.reshape(B, H, W, C)
    y = x * sd[p + "skip_scale"] + ss2d(sd, p + "self_attention.",
                                        F.layer_norm(x, (C,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]), scan_fn)
    n = F.layer_norm(y, (C,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]).permute(0, 3, 1, 2)
    y = y * sd[p + "skip_scale2"] + cab(sd, p + "conv_blk.", n).permute(0, 2, 3, 1)
    return y.reshape(B, L, C)


def mambair_forward(sd, lr, scan_fn=selective_scan_ref):
    """lr [B,3,H,W] -> (sr un-clamped, feat = conv_after_body output)."""
    depths = {}
    for k in sd:
        if k.startswith("layers.") and ".blocks." in k:
            parts = k.split(".")
            depths[int(parts[1])] = max(depths.get(int(parts[1]), 0), int(parts[4]) + 1)
    mean = rgb_mean(lr)
    x = F.conv2d(lr - mean, sd["conv_first.weight"], sd["conv_first.bias"], padding=1)
    B, C, H, W = x.shape
    t = F.layer_norm(to_tokens(x), (C,), sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"])
    for i in sorted(depths):
        r = t
        for j in range(depths[i]):
            r = vss_block(sd, f"layers.{i}.residual_group.blocks.{j}.", r, H, W, scan_fn)
        r = F.conv2d(to_image(r, H, W), sd[f"layers.{i}.conv.weight"], sd[f"layers.{i}.conv.bias"], padding=1)
        t = to_tokens(r) + t
    t = F.layer_norm(t, (C,), sd["norm.weight"], sd["norm.bias"])
    feat = F.conv2d(to_image(t, H, W), sd["conv_after_body.weight"], sd["conv_after_body.bias"], padding=1)
    return sr_tail(sd, feat + x) + mean, feat
