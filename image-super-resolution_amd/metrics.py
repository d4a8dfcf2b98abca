"""PSNR / SSIM on the device, composed from the C-ABI kernels.

Mirrors src/utils/metrics.py: rgb_to_y :30 (ITU-R BT.601, MATLAB style), calculate_psnr :76 (clamp, crop_border,
Y channel, 10 log10(1/MSE)), calculate_ssim_torch :129 (11x11 Gaussian sigma 1.5, zero padding, C1 = 0.01^2,
C2 = 0.03^2, mean of the map).  The reference prefers scikit-image's SSIM when it is importable (metrics.py:229);
scikit-image is absent from this image, so the torch formulation is the one restated here and in the oracle.
Inputs are channels-last maps [B, H, W, 3] in [0, 1] on the device.
"""
from __future__ import annotations

import math

import torch

from . import ops

_CACHE = {}


def _consts(device):
    key = str(device)
    if key not in _CACHE:
        w = torch.tensor([65.481, 128.553, 24.966]).reshape(1, 3, 1, 1) / 255.0
        to_y = ops.pack_conv(w, torch.tensor([16.0 / 255.0]), device, cin_pad=4)
        g = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)])
        g = g / g.sum()
        gauss = ops.pack_dwconv((g[:, None] @ g[None, :]).float().reshape(1, 1, 11, 11), None, device)
        _CACHE[key] = (to_y, gauss)
    return _CACHE[key]


def _prep(img, crop_border, y_channel):
    B, H, W, _ = img.shape
    c = crop_border
    x = ops.crop(img, H - 2 * c, W - 2 * c, clamp=True, y0=c, x0=c)                         # clamp(0,1) + crop
    if y_channel:
        x = ops.conv2d(x, _consts(img.device)[0], tile_hint=13)      # exact f32 1x1 conv 3 -> 1
    return x


def psnr(img1, img2, crop_border=0, test_y_channel=False) -> float:
    a, b = _prep(img1, crop_border, test_y_channel), _prep(img2, crop_border, test_y_channel)
    C = a.shape[3]
    d = ops.scale_add(a[..., :C], b[..., :C], beta=-1.0)
    sq = ops.mul_add(d, d)
    mse = float(ops.colmean(sq).double().mean().item())
    return float("inf") if mse < 1e-10 else 10.0 * math.log10(1.0 / mse)


def ssim(img1, img2, crop_border=0, test_y_channel=True) -> float:
    """calculate_ssim_torch on the (cropped) Y channel (test_y_channel=True) or averaged over RGB."""
    a, b = _prep(img1, crop_border, test_y_channel), _prep(img2, crop_border, test_y_channel)
    gauss = _consts(img1.device)[1]
    vals = []
    for c in range(a.shape[3] if not test_y_channel else 1):
        x, y = a[..., c:c + 1], b[..., c:c + 1]
        mu1, mu2 = ops.dwconv2d(x, gauss), ops.dwconv2d(y, gauss)
        e11, e22, e12 = (ops.dwconv2d(ops.mul_add(p, q), gauss) for p, q in ((x, x), (y, y), (x, y)))
        out = ops.new_map(*x.shape[:3], 1, x.device)
        ops.hip.call("ffsr_ssim_map_f32", mu1.data_ptr(), mu2.data_ptr(), e11.data_ptr(), e22.data_ptr(), e12.data_ptr(),
                     ops.ld(mu1), out.data_ptr(), ops.ld(out), ops.rows(x), torch.cuda.current_stream().cuda_stream)
        vals.append(float(ops.colmean(out).double().mean().item()))
    return sum(vals) / len(vals)
