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


# ---------------------------------------------------------------------------------------------------------------------
# The evaluation script's variant, utils/utils_image.py:148-189 (cal_psnr_ssim): uint8 images, border crop, OpenCV's
# Y of COLOR_RGB2YCrCb, PSNR = 20 log10(255 / sqrt(mse)) in float64, SSIM = skimage.metrics.structural_similarity.
# PARITY UNPINNED: cv2 and scikit-image are absent from this image (and utils_image.py imports cv2 at :10), so the
# two library calls are restated from their published algorithms (numpy float64 + scipy's uniform_filter) and the
# reference holds no fixture for them.
def rgb2y_opencv_u8(img):
    """cv2.cvtColor(img, cv2.COLOR_RGB2YCrCb)[:, :, 0] for uint8 HxWx3: OpenCV's 8-bit path is fixed point,
    Y = (4899 R + 9617 G + 1868 B + 2^13) >> 14 (0.299 / 0.587 / 0.114 scaled by 2^14, round to nearest)."""
    import numpy as np
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def _ssim_skimage(x, y, data_range=255.0, win=7):
    """skimage.metrics.structural_similarity defaults for 2-D float64 inputs: uniform win x win filter, K1 0.01, K2 0.03,
    sample covariance (NP / (NP - 1)), mean of the map cropped by (win - 1) // 2."""
    import numpy as np
    from scipy.ndimage import uniform_filter
    NP = win * win
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
    uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    p = (win - 1) // 2
    return float(np.mean(S[p:S.shape[0] - p, p:S.shape[1] - p], dtype=np.float64))


def psnr_ssim_u8(output_img, target_img, crop_border=4, test_y_channel=True):
    """cal_psnr_ssim on two uint8 HxWx3 RGB arrays (the image-file reads of :150-151 are the caller's)."""
    import numpy as np
    h, w = min(output_img.shape[0], target_img.shape[0]), min(output_img.shape[1], target_img.shape[1])
    a, b = output_img[:h, :w, :], target_img[:h, :w, :]
    if crop_border > 0:
        a, b = a[crop_border:-crop_border, crop_border:-crop_border, :], b[crop_border:-crop_border, crop_border:-crop_border, :]
    if test_y_channel:
        a, b = rgb2y_opencv_u8(a).astype(np.float64), rgb2y_opencv_u8(b).astype(np.float64)
    else:
        a, b = a.astype(np.float64), b.astype(np.float64)
    mse = np.mean((a - b) ** 2)
    psnr_v = float("inf") if mse == 0 else 20 * math.log10(255.0 / math.sqrt(mse))
    if a.ndim == 2:
        ssim_v = _ssim_skimage(a, b)
    else:
        ssim_v = sum(_ssim_skimage(a[..., c], b[..., c]) for c in range(3)) / 3.0
    return psnr_v, ssim_v
