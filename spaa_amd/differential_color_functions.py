"""HIP versions of perc_al/differential_color_functions.py (same names, NCHW tensors in / out).

`rgb2lab_diff(rgb, device)` (:39-64) and `ciede2000_diff(lab1, lab2, device)` (:109-180) reproduce the reference's
constants exactly (0.0405 threshold, `aHP - 39`, f(0)=0, the 1e-4 guards) and are differentiable like the
reference's (its `_diff` suffix): `torch.autograd.Function`s around the forward kernels and their hand-derived
adjoints (`spaa_rgb2lab_bwd`, `spaa_ciede2000_bwd`).  `stealth_loss_with_grad` exposes the fused loss kernel's
analytic gradient (what the attack loop uses instead of autograd over ~300 ATen ops).
"""
import torch

from . import _lib
from .models import to_nhwc4, to_nchw


class _Rgb2LabFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb):
        with _lib.on_device(rgb.device):
            x4 = to_nhwc4(rgb)
            lab = torch.zeros_like(x4)
            _lib.call('spaa_rgb2lab', _lib.ptr(x4), _lib.ptr(lab), x4.shape[0] * x4.shape[1] * x4.shape[2])
            ctx.x4 = x4
            return to_nchw(lab)

    @staticmethod
    def backward(ctx, g_lab):
        with _lib.on_device(g_lab.device):
            g4 = to_nhwc4(g_lab)
            gx = torch.zeros_like(g4)
            _lib.call('spaa_rgb2lab_bwd', _lib.ptr(ctx.x4), _lib.ptr(g4), _lib.ptr(gx), g4.shape[0] * g4.shape[1] * g4.shape[2])
            return to_nchw(gx)


class _Ciede2000Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lab1, lab2):
        with _lib.on_device(lab1.device):
            a, b = to_nhwc4(lab1), to_nhwc4(lab2)
            n, h, w, _ = a.shape
            de = torch.zeros(n, h, w, device=a.device)
            _lib.call('spaa_ciede2000', _lib.ptr(a), _lib.ptr(b), _lib.ptr(de), n * h * w)
            ctx.a, ctx.b = a, b
            return de

    @staticmethod
    def backward(ctx, g_de):
        with _lib.on_device(g_de.device):
            a, b = ctx.a, ctx.b
            g = g_de.detach().float().contiguous()
            need1, need2 = ctx.needs_input_grad
            g1 = torch.zeros_like(a) if need1 else None
            g2 = torch.zeros_like(b) if need2 else None
            _lib.call('spaa_ciede2000_bwd', _lib.ptr(a), _lib.ptr(b), _lib.ptr(g), _lib.ptr(g1), _lib.ptr(g2), g.numel())
            return (to_nchw(g1) if need1 else None), (to_nchw(g2) if need2 else None)


def _on(x, device):
    x = x if device is None else x.to(device)
    if not x.is_cuda:
        raise RuntimeError('spaa_amd needs CUDA(HIP) tensors; there is no CPU fallback')
    return x


def rgb2lab_diff(rgb_image, device=None):
    """:39-64.  [B,3,H,W] sRGB in [0,1] -> Lab, differentiable."""
    return _Rgb2LabFn.apply(_on(rgb_image, device))


def ciede2000_diff(lab1, lab2, device=None):
    """:109-180.  Two [B,3,H,W] Lab images -> [B,H,W] per-pixel CIEDE2000, differentiable w.r.t. both."""
    lab1, lab2 = _on(lab1, device), _on(lab2, device)
    if lab1.shape != lab2.shape:
        lab1, lab2 = torch.broadcast_tensors(lab1, lab2)
    return _Ciede2000Fn.apply(lab1, lab2)


def deltaE(x, y):
    """:183-190 — mean CIEDE2000 over all pixels."""
    while x.ndim < 4:
        x, y = x[None], y[None]
    with torch.no_grad():
        return ciede2000_diff(rgb2lab_diff(x), rgb2lab_diff(y)).mean().item()


def stealth_loss_with_grad(cam_infer, cam_scene, caml2_w=1.0, camdE_w=1.0):
    """Fused forward+backward of the camera-side stealth terms (projector_based_attack.py:279-287).
    Returns (caml2 [B], camdE [B], grad [B,3,H,W]) with grad = d/d cam_infer of
    sum_b (caml2_w * caml2_b + camdE_w * camdE_b)."""
    with _lib.on_device(cam_infer.device):
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
