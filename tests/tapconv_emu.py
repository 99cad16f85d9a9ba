"""Test helper: a plain-PyTorch (CPU) emulation of the tap-list convolution *semantics*
(include/spaa_hip.h, spaa_tapconv_t), used to check spaa_amd/convplan.py's tap/weight packing without a GPU."""
import torch
import torch.nn.functional as F


def emulate(plan, inp, hout, wout):
    """inp [B,Hin,Win,Cs] (NHWC) -> out [B,hout,wout,cout] using plan.classes_host."""
    b, hin, win, _ = inp.shape
    cin = plan.cin
    out = torch.zeros(b, hout, wout, plan.cout)
    if getattr(plan, 'nfold', 1) > 1:
        # spaa_tapconv_t.nfold: GEMM row c*Cout + n -> output pixel (2y + c//2, 2x + c%2), channel n
        # (k2/s2: the one tap (0, 0); k3/s2: the 2x2 neighbourhood, zero weights where a class has no tap)
        assert plan.s_in == 1 and plan.s_out == 2
        hm, wm = (hout + 1) // 2, (wout + 1) // 2
        acc = torch.zeros(b, hm, wm, 4 * plan.cout)
        for dy, dx, w in plan.classes_host[0].taps:
            iy, ix = torch.arange(hm) + dy, torch.arange(wm) + dx
            vy, vx = (iy >= 0) & (iy < hin), (ix >= 0) & (ix < win)
            g = inp[:, iy.clamp(0, hin - 1)][:, :, ix.clamp(0, win - 1)][..., :cin]
            acc += (g * (vy.view(1, -1, 1, 1) & vx.view(1, 1, -1, 1))) @ w.t()
        for c in range(4):
            oy, ox = 2 * torch.arange(hm) + c // 2, 2 * torch.arange(wm) + c % 2
            ky, kx = oy < hout, ox < wout
            out[:, oy[ky][:, None], ox[kx][None, :]] = acc[:, ky][:, :, kx][..., c * plan.cout:(c + 1) * plan.cout]
        return out + plan.bias.cpu() if plan.bias is not None else out
    for c in plan.classes_host:
        if plan.s_out == 1:
            hm, wm = hout, wout
        else:
            hm, wm = (hout + 1) // 2, (wout + 1) // 2
        acc = torch.zeros(b, hm, wm, plan.cout)
        ys = torch.arange(hm) * plan.s_in
        xs = torch.arange(wm) * plan.s_in
        for dy, dx, w in c.taps:
            iy, ix = ys + dy, xs + dx
            vy = (iy >= 0) & (iy < hin)
            vx = (ix >= 0) & (ix < win)
            g = inp[:, iy.clamp(0, hin - 1)][:, :, ix.clamp(0, win - 1)][..., :cin]
            g = g * (vy.view(1, -1, 1, 1) & vx.view(1, 1, -1, 1))
            acc += g @ w.t()
        oy = c.oy0 + plan.s_out * torch.arange(hm)
        ox = c.ox0 + plan.s_out * torch.arange(wm)
        ky, kx = oy < hout, ox < wout
        out[:, oy[ky][:, None], ox[kx][None, :]] = acc[:, ky][:, :, kx]
    if plan.bias is not None:
        out = out + plan.bias.cpu()
    return out


def nhwc(x, cs=None):
    """NCHW -> NHWC with optional zero channel padding."""
    y = x.permute(0, 2, 3, 1).contiguous()
    if cs is not None and cs > y.shape[-1]:
        y = F.pad(y, (0, cs - y.shape[-1]))
    return y


def nchw(x, c=None):
    y = x.permute(0, 3, 1, 2)
    return (y[:, :c] if c is not None else y).contiguous()
