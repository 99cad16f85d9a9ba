"""HIP versions of perc_al/differential_color_functions.py (same names, NCHW tensors in / out).

`rgb2lab_diff(rgb, device)` (:39-64) and `ciede2000_diff(lab1, lab2, device)` (:109-180) reproduce the reference's
constants exactly (0.0405 threshold, `aHP - 39`, f(0)=0, the 1e-4 guards).  `deltaE_map_with_grad` exposes the
fused loss kernel's analytic gradient (what the attack loop uses instead of autograd over ~300 ATen ops).
"""
import torch

from . import _lib
from .models import to_nhwc4, to_nchw


def rgb2lab_diff(rgb_image, device=None):
    x4 = to_nhwc4(rgb_image if device is None else rgb_image.to(device))
    lab = torch.zeros_like(x4)
    _lib.call('spaa_rgb2lab', _lib.ptr(x4), _lib.ptr(lab), x4.shape[0] * x4.shape[1] * x4.shape[2])
    return to_nchw(lab)


def ciede2000_diff(lab1, lab2, device=None):
    a = to_nhwc4(lab1 if device is None else lab1.to(device))
    b = to_nhwc4(lab2 if device is None else lab2.to(device))
    n, h, w, _ = a.shape
    de = torch.zeros(n, h, w, device=a.device)
    _lib.call('spaa_ciede2000', _lib.ptr(a), _lib.ptr(b), _lib.ptr(de), n * h * w)
    return de


def deltaE(x, y):
    """:183-190 — mean CIEDE2000 over all pixels."""
    while x.ndim < 4:
        x, y = x[None], y[None]
    return ciede2000_diff(rgb2lab_diff(x), rgb2lab_diff(y)).mean().item()


def stealth_loss_with_grad(cam_infer, cam_scene, caml2_w=1.0, camdE_w=1.0):
    """Fused forward+backward of the camera-side stealth terms (projector_based_attack.py:279-287).
    Returns (caml2 [B], camdE [B], grad [B,3,H,W]) with grad = d/d cam_infer of
    sum_b (caml2_w * caml2_b + camdE_w * camdE_b)."""
    y4, s4 = to_nhwc4(cam_infer), to_nhwc4(cam_scene)
    b, h, w, _ = y4.shape
    lab = torch.zeros_like(s4)
    _lib.call('spaa_rgb2lab', _lib.ptr(s4), _lib.ptr(lab), b * h * w)
    nblk = (h * w + 255) // 256
    part = torch.zeros(b, nblk, 3, device=y4.device)
    g = torch.zeros_like(y4)
    _lib.call('spaa_stealth_loss_fwd_bwd', _lib.ptr(y4), _lib.ptr(s4), _lib.ptr(lab), float(caml2_w), float(camdE_w),
              1.0 / (h * w), _lib.ptr(g), None, _lib.ptr(part), b, h * w)
    sums = part.sum(dim=1) / (h * w)
    return sums[:, 0], sums[:, 1], to_nchw(g)
