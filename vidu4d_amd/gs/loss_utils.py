"""Image losses imported by the hot path (reference: gs/utils/loss_utils.py:17-67)."""
import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def _gaussian_window(size, sigma, channels, device, dtype):
    x = torch.arange(size, dtype=dtype, device=device) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = (g / g.sum())[:, None]
    w = (g @ g.T)[None, None]
    return w.expand(channels, 1, size, size).contiguous()


def ssim(img1, img2, window_size=11, size_average=True):
    if img1.dim() == 3:
        img1, img2 = img1[None], img2[None]
    c = img1.shape[1]
    w = _gaussian_window(window_size, 1.5, c, img1.device, img1.dtype)
    p = window_size // 2
    mu1, mu2 = F.conv2d(img1, w, padding=p, groups=c), F.conv2d(img2, w, padding=p, groups=c)
    s1 = F.conv2d(img1 * img1, w, padding=p, groups=c) - mu1 * mu1
    s2 = F.conv2d(img2 * img2, w, padding=p, groups=c) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, w, padding=p, groups=c) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)
