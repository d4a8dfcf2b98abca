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


# ---------------------------------------------------------------------------------------------------------------------
# The evaluation script's variant (utils/utils_image.py:148-189, what eval.py reports): uint8 in, OpenCV's 8-bit Y,
# float64 PSNR, scikit-image's SSIM -- integer kernels of csrc/ffsr_metrics.hip.
def psnr_ssim_u8(output_img, target_img, crop_border=4, test_y_channel=True, device="cuda"):
    """cal_psnr_ssim on two uint8 HxWx3 RGB images (numpy arrays or tensors) -> (psnr dB, ssim), both Python floats.
    Mirrors :154-187: common size, border crop, Y = cv2.COLOR_RGB2YCrCb luma (or the three RGB planes), mse -> PSNR,
    structural_similarity(data_range=255)."""
    from .engine import require_gpu
    device = require_gpu(device)
    imgs = []
    for im in (output_img, target_img):
        t = torch.as_tensor(im)
        if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
            raise ValueError(f"expected uint8 HxWx3 images, got {t.dtype} {tuple(t.shape)}")
        imgs.append(t)
    h, w = min(imgs[0].shape[0], imgs[1].shape[0]), min(imgs[0].shape[1], imgs[1].shape[1])
    c = int(crop_border)
    Hc, Wc = h - 2 * c, w - 2 * c
    P = 1 if test_y_channel else 3
    planes = []
    with torch.cuda.device(device):
        st = torch.cuda.current_stream(device).cuda_stream
        for t in imgs:
            t = t[:h, :w].contiguous().to(device)
            pl = torch.empty(P, Hc, Wc, dtype=torch.uint8, device=device)
            ops.hip.call("ffsr_u8_planes", t.data_ptr(), h, w, c, int(test_y_channel), pl.data_ptr(), st)
            planes.append(pl)
        n_partial = 1024
        partial = torch.empty(2 * n_partial, dtype=torch.float64, device=device)
        out = torch.empty(2, dtype=torch.float64, device=device)
        ops.hip.call("ffsr_psnr_ssim_u8", planes[0].data_ptr(), planes[1].data_ptr(), P, Hc, Wc, partial.data_ptr(),
                     n_partial, out.data_ptr(), st)
        mse, ssim_v = out.tolist()
    return (float("inf") if mse == 0 else 20.0 * math.log10(255.0 / math.sqrt(mse))), ssim_v


def cal_psnr_ssim(output_image_path, target_image_path, crop_border=4, test_y_channel=True):
    """utils/utils_image.py:148 with its own signature: two image files -> (psnr, ssim)."""
    import numpy as np
    from PIL import Image

    def read(p):
        with Image.open(p) as im:
            return np.array(im.convert("RGB"), dtype=np.uint8)
    return psnr_ssim_u8(read(output_image_path), read(target_image_path), crop_border, test_y_channel)
