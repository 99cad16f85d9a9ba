"""Image-distance metrics of the reference's evaluation step on the GPU (HIP kernels, no CPU fallback).

Mirrors /root/reference/src/python/utils.py:420-491 — `calc_img_dists(x, y)` returns
(PSNR, RMSE, SSIM, mean-L2 * 255, mean-L_inf * 255, mean dE2000) as Python floats — and the single metrics
`psnr`, `rmse`, `ssim` (pytorch_ssim/__init__.py:98-107), `l2_norm`, `linf_norm`, `deltaE`
(perc_al/differential_color_functions.py:183-190).  x, y: [3,H,W] or [B,3,H,W] float tensors in [0,1] on any device
(moved to the GPU as the reference does).  Two launches per call pair: one fused pass for the per-pixel metrics
(`spaa_img_dists`) and one tiled 11x11 pass for SSIM (`spaa_ssim`); block partials are added in fixed order.
"""
import math

import torch

from . import _lib
from .models import to_nhwc4


def _window(window_size=11, sigma=1.5):
    """pytorch_ssim/__init__.py:9-21: normalised 1-D Gaussian, outer product, float32."""
    g = torch.Tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float().reshape(-1).contiguous()


def _prep(x, y):
    if not torch.cuda.is_available():
        raise RuntimeError('spaa_amd.metrics needs the GPU (no CPU fallback)')
    x, y = (t if t.ndim == 4 else t[None] for t in (x, y))
    if x.shape != y.shape or x.shape[1] != 3:
        raise ValueError(f'expected two [B,3,H,W] / [3,H,W] images of the same shape, got {tuple(x.shape)} {tuple(y.shape)}')
    dev = x.device if x.is_cuda else (y.device if y.is_cuda else torch.device('cuda', torch.cuda.current_device()))
    return to_nhwc4(x.detach().float().to(dev).contiguous()), to_nhwc4(y.detach().float().to(dev).contiguous())


def _pixel_sums(x4, y4):
    npix = x4.shape[0] * x4.shape[1] * x4.shape[2]
    partial = torch.zeros((npix + 255) // 256, 4, device=x4.device)
    _lib.call('spaa_img_dists', _lib.ptr(x4), _lib.ptr(y4), _lib.ptr(partial), npix)
    return partial.double().sum(0).tolist(), npix


def _ssim_mean(x4, y4):
    b, h, w, _ = x4.shape
    partial = torch.zeros(b, (h + 15) // 16, (w + 15) // 16, device=x4.device)
    win = _window().to(x4.device)
    _lib.call('spaa_ssim', _lib.ptr(x4), _lib.ptr(y4), _lib.ptr(win), _lib.ptr(partial), b, h, w)
    return partial.double().sum().item() / (b * 3 * h * w)


def calc_img_dists(x, y):
    """utils.py:420-423."""
    x4, y4 = _prep(x, y)
    (sq, l2, li, de), npix = _pixel_sums(x4, y4)
    mse = sq / (3 * npix)
    return (10 * math.log10(1 / mse), math.sqrt(mse * 3), _ssim_mean(x4, y4), l2 / npix * 255, li / npix * 255, de / npix)


def psnr(x, y):
    (sq, *_), npix = _pixel_sums(*_prep(x, y))
    return 10 * math.log10(1 / (sq / (3 * npix)))


def rmse(x, y):
    (sq, *_), npix = _pixel_sums(*_prep(x, y))
    return math.sqrt(sq / (3 * npix) * 3)


def ssim(x, y):
    return _ssim_mean(*_prep(x, y))


def l2_norm(x, y):
    (_, l2, _, _), npix = _pixel_sums(*_prep(x, y))
    return l2 / npix * 255


def linf_norm(x, y):
    (_, _, li, _), npix = _pixel_sums(*_prep(x, y))
    return li / npix * 255


def deltaE(x, y):
    (_, _, _, de), npix = _pixel_sums(*_prep(x, y))
    return de / npix
