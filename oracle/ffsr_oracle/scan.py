"""Selective scan (S6) recurrence -- restatement of the published mamba-ssm algorithm.

PARITY UNPINNED: the reference calls ``mamba_ssm.ops.selective_scan_interface.selective_scan_fn``
(mambair_arch.py:11, call site :356-362), an un-vendored CUDA wheel (mamba-ssm 2.3.0 per
scripts/kaggle_inference_fixed.py:19) that is absent from /root/reference and from this image.
No reference test or fixture pins its output.  This file follows the published definition:

    delta_t = softplus(dt_t + delta_bias)                 (PyTorch softplus, threshold 20)
    h_t     = exp(delta_t * A) * h_{t-1} + delta_t * B_t * u_t
    y_t     = <C_t, h_t> + D * u_t

with u, delta: [B, Dm, L]; A: [Dm, N]; B, C: [B, G, N, L] shared by the Dm/G channels of each
group; D, delta_bias: [Dm]; fp32 throughout, h_0 = 0.
"""
import torch
import torch.nn.functional as F


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                       delta_softplus=False, return_last_state=False):
    u = u.float()
    delta = delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias.float()[None, :, None]
    if delta_softplus:
        delta = F.softplus(delta)
    Bsz, Dm, L = u.shape
    N = A.shape[1]
    G = B.shape[1]
    rep = Dm // G
    A = A.float()
    Bg = B.float().repeat_interleave(rep, dim=1)   # [B, Dm, N, L]
    Cg = C.float().repeat_interleave(rep, dim=1)
    h = torch.zeros(Bsz, Dm, N, dtype=torch.float32, device=u.device)
    ys = torch.empty(Bsz, Dm, L, dtype=torch.float32, device=u.device)
    du = delta * u
    for t in range(L):
        dA = torch.exp(delta[:, :, t, None] * A[None])
        h = dA * h + du[:, :, t, None] * Bg[:, :, :, t]
        ys[:, :, t] = (h * Cg[:, :, :, t]).sum(-1)
    if D is not None:
        ys = ys + u * D.float()[None, :, None]
    if z is not None:
        ys = ys * F.silu(z.float())
    if return_last_state:
        return ys, h
    return ys
