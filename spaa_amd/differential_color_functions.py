"""HIP versions of perc_al/differential_color_functions.py (same names, NCHW tensors in / out).

`rgb2lab_diff(rgb, device)` (:39-64) and `ciede2000_diff(lab1, lab2, device)` (:109-180) reproduce the reference's
constants exactly (0.0405 threshold, `aHP - 39`, f(0)=0, the 1e-4 guards) and are differentiable like the
reference's (its `_diff` suffix): they are the registered custom ops `spaa::rgb2lab` / `spaa::ciede2000`
(spaa_amd/ops.py) with hand-derived adjoint kernels (`spaa_rgb2lab_bwd`, `spaa_ciede2000_bwd`).
`stealth_loss_with_grad` exposes the fused loss kernel's analytic gradient (what the attack loop uses instead of
autograd over ~300 ATen ops).
"""
import torch

from . import ops


def _on(x, device):
    x = x if device is None else x.to(device)
    if not x.is_cuda:
        raise RuntimeError('spaa_amd needs CUDA(HIP) tensors; there is no CPU fallback')
    return x


def rgb2lab_diff(rgb_image, device=None):
    """:39-64.  [B,3,H,W] sRGB in [0,1] -> Lab, differentiable."""
    t = torch.ops.spaa
    return t.nhwc4_to_nchw(t.rgb2lab(t.nchw_to_nhwc4(_on(rgb_image, device))))


def ciede2000_diff(lab1, lab2, device=None):
    """:109-180.  Two [B,3,H,W] Lab images -> [B,H,W] per-pixel CIEDE2000, differentiable w.r.t. both."""
    lab1, lab2 = _on(lab1, device), _on(lab2, device)
    if lab1.shape != lab2.shape:
        lab1, lab2 = torch.broadcast_tensors(lab1, lab2)
    t = torch.ops.spaa
    return t.ciede2000(t.nchw_to_nhwc4(lab1), t.nchw_to_nhwc4(lab2))


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
    t = torch.ops.spaa
    with torch.no_grad():
        y4, s4 = t.nchw_to_nhwc4(cam_infer), t.nchw_to_nhwc4(cam_scene)
        hw = y4.shape[1] * y4.shape[2]
        sums, g = t.stealth_loss(y4, s4, t.rgb2lab(s4), float(caml2_w), float(camdE_w), 1.0 / hw)
        return sums[:, 0] / hw, sums[:, 1] / hw, t.nhwc4_to_nchw(g)
