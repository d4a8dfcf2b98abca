"""PSNR / SSIM -- CPU oracle (test infrastructure).  Follows src/utils/metrics.py: rgb_to_y :30-52,
calculate_psnr :76-126, calculate_ssim_torch :129-186 (the branch taken when scikit-image is not installed)."""
import math

import torch
import torch.nn.functional as F


def rgb_to_y(img):
    return (65.481 * img[:, 0:1] + 128.553 * img[:, 1:2] + 24.966 * img[:, 2:3] + 16.0) / 255.0


def _prep(img, crop_border, y):
    img = img.clamp(0, 1)
    if crop_border > 0:
        img = img[:, :, crop_border:-crop_border, crop_border:-crop_border]
    return rgb_to_y(img) if y else img


def psnr(a, b, crop_border=0, test_y_channel=False):
    a, b = _prep(a, crop_border, test_y_channel), _prep(b, crop_border, test_y_channel)
    mse = torch.mean((a - b) ** 2).item()
    return float("inf") if mse < 1e-10 else 10 * math.log10(1.0 / mse)


def ssim(a, b, crop_border=0, test_y_channel=True, window_size=11, sigma=1.5):
    a, b = _prep(a, crop_border, test_y_channel), _prep(b, crop_border, test_y_channel)
    C = a.shape[1]
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = g / g.sum()
    win = (g[:, None] @ g[None, :]).float().expand(C, 1, window_size, window_size).contiguous()
    pad = window_size // 2
    mu1, mu2 = F.conv2d(a, win, padding=pad, groups=C), F.conv2d(b, win, padding=pad, groups=C)
    s1 = F.conv2d(a * a, win, padding=pad, groups=C) - mu1 ** 2
    s2 = F.conv2d(b * b, win, padding=pad, groups=C) - mu2 ** 2
    s12 = F.conv2d(a * b, win, padding=pad, groups=C) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))
    return m.mean().item()
