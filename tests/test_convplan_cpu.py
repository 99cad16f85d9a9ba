"""CPU checks of the conv -> tap-list lowering (spaa_amd/convplan.py) against torch's own conv ops and autograd."""
import pytest
import torch
import torch.nn.functional as F

from spaa_amd import convplan as cp
from tapconv_emu import emulate, nhwc, nchw

torch.manual_seed(0)


@pytest.mark.parametrize('ci,co,k,s,p,h,w', [(3, 8, 3, 1, 1, 9, 11), (6, 5, 3, 2, 1, 12, 10), (4, 7, 1, 1, 0, 6, 5),
                                             (3, 4, 7, 2, 3, 20, 18), (5, 6, 1, 2, 0, 8, 8), (2, 3, 3, 2, 1, 9, 7)])
def test_conv_fwd_and_dgrad(ci, co, k, s, p, h, w):
    x = torch.randn(2, ci, h, w, requires_grad=True)
    wt = torch.randn(co, ci, k, k)
    b = torch.randn(co)
    y = F.conv2d(x, wt, b, s, p)
    plan = cp.conv_fwd_plan(wt, b, s, p, device='cpu')
    y2 = emulate(plan, nhwc(x.detach(), plan.cin_p), y.shape[2], y.shape[3])
    assert torch.allclose(nchw(y2), y, atol=1e-4)
    gy = torch.randn_like(y)
    y.backward(gy)
    dplan = cp.conv_dgrad_plan(wt, s, p, device='cpu')
    gx = emulate(dplan, nhwc(gy, dplan.cin_p), h, w)
    assert torch.allclose(nchw(gx), x.grad, atol=1e-4)


@pytest.mark.parametrize('ci,co,h,w', [(8, 32, 12, 10), (4, 64, 9, 7)])
def test_k3s2_classes_folded_into_gemm_rows(ci, co, h, w):
    """3x3 / stride-2 input gradient and transposed convolution with the four parity classes folded into the GEMM rows (zero
    weights where a class has no tap): the same result as the four separate classes."""
    x = torch.randn(2, ci, h, w, requires_grad=True)
    wt = torch.randn(co, ci, 3, 3)
    y = F.conv2d(x, wt, None, 2, 1)
    gy = torch.randn_like(y)
    y.backward(gy)
    dplan = cp.conv_dgrad_plan(wt, 2, 1, device='cpu', fold=True)
    assert dplan.nfold == 4 and dplan.ntaps_total == 4
    gx = emulate(dplan, nhwc(gy, dplan.cin_p), h, w)
    assert torch.allclose(nchw(gx), x.grad, atol=1e-4)
    xt = torch.randn(2, co, h, w)
    wtt = torch.randn(co, ci, 3, 3)
    yt = F.conv_transpose2d(xt, wtt, None, 2, 1, 1)
    tplan = cp.deconv_fwd_plan(wtt, None, 2, 1, device='cpu', fold=True)
    assert tplan.nfold == 4
    assert torch.allclose(nchw(emulate(tplan, nhwc(xt), yt.shape[2], yt.shape[3])), yt, atol=1e-4)


def test_conv_dgrad_channel_subset():
    x = torch.randn(1, 6, 8, 8, requires_grad=True)
    wt = torch.randn(5, 6, 3, 3)
    y = F.conv2d(x, wt, None, 2, 1)
    gy = torch.randn_like(y)
    y.backward(gy)
    dplan = cp.conv_dgrad_plan(wt, 2, 1, device='cpu', in_ch=(3, 6))
    gx = emulate(dplan, nhwc(gy, dplan.cin_p), 8, 8)
    assert torch.allclose(nchw(gx), x.grad[:, 3:6], atol=1e-4)


@pytest.mark.parametrize('ci,co,k,p,op,h,w', [(6, 4, 3, 1, 1, 5, 7), (5, 3, 2, 0, 0, 6, 4), (32, 8, 2, 0, 0, 5, 3)])
def test_deconv_fwd_and_dgrad(ci, co, k, p, op, h, w):
    x = torch.randn(2, ci, h, w, requires_grad=True)
    wt = torch.randn(ci, co, k, k)
    b = torch.randn(co)
    y = F.conv_transpose2d(x, wt, b, 2, p, op)
    plan = cp.deconv_fwd_plan(wt, b, 2, p, device='cpu')
    assert plan.nfold == (4 if (k == 2 and ci % 32 == 0) else 1)
    y2 = emulate(plan, nhwc(x.detach(), plan.cin_p), y.shape[2], y.shape[3])
    assert torch.allclose(nchw(y2), y, atol=1e-4)
    gy = torch.randn_like(y)
    y.backward(gy)
    dplan = cp.deconv_dgrad_plan(wt, 2, p, device='cpu')
    gx = emulate(dplan, nhwc(gy, dplan.cin_p), h, w)
    assert torch.allclose(nchw(gx), x.grad, atol=1e-4)


def test_linear_and_bn_fold():
    x = torch.randn(3, 10, requires_grad=True)
    w, b = torch.randn(7, 10), torch.randn(7)
    y = F.linear(x, w, b)
    plan = cp.linear_fwd_plan(w, b, device='cpu')
    y2 = emulate(plan, F.pad(x.detach(), (0, 2)).view(3, 1, 1, 12), 1, 1)
    assert torch.allclose(y2.view(3, 7), y, atol=1e-5)
    gy = torch.randn_like(y)
    y.backward(gy)
    dplan = cp.linear_dgrad_plan(w, device='cpu')
    gx = emulate(dplan, F.pad(gy, (0, 1)).view(3, 1, 1, 8), 1, 1)
    assert torch.allclose(gx.view(3, 10), x.grad, atol=1e-5)
    cw = torch.randn(4, 3, 3, 3)
    g, bb, m, v = torch.rand(4) + 0.5, torch.randn(4), torch.randn(4), torch.rand(4) + 0.5
    xi = torch.randn(1, 3, 6, 6)
    ref = F.batch_norm(F.conv2d(xi, cw, None, 1, 1), m, v, g, bb, False, 0.0, 1e-5)
    fw, fb = cp.fold_bn(cw, g, bb, m, v)
    assert torch.allclose(F.conv2d(xi, fw, fb, 1, 1), ref, atol=1e-5)


@pytest.mark.parametrize('ci,co,k,s,p,h,w', [(3, 64, 7, 2, 3, 14, 18), (3, 32, 3, 2, 1, 10, 12), (3, 64, 3, 1, 1, 7, 9), (3, 32, 3, 2, 0, 11, 9)])
def test_thin_fold_layout(ci, co, k, s, p, h, w):
    """The folded weight layout of the thin-output matrix-core kernel (ConvPlan.thin_fold, csrc/tapconv_thinmf.hip): emulated on the
    CPU exactly as the kernel reads it -- GEMM row = class * 4 + channel, [channel block][tap column][plane][tap row][16][32], the
    chunk swap of rows 8-15, output pixel (S y + class / S, S x + class % S) -- it gives the layer's input gradient; the three bf16
    planes sum to the fp32 weights exactly."""
    x = torch.randn(2, ci, h, w, requires_grad=True)
    wt = torch.randn(co, ci, k, k)
    y = F.conv2d(x, wt, None, s, p)
    gy = torch.randn_like(y)
    y.backward(gy)
    dplan = cp.conv_dgrad_plan(wt, s, p, device='cpu')
    assert dplan.thin_ok()
    S = dplan.s_out
    dy0, dy1, dx0, dx1 = dplan.tap_range
    tbh, tbw, nkb = max(dy1 - dy0 + 1, 2), dx1 - dx0 + 1, dplan.cin_p // 32
    planes = dplan.thin_fold(False).view(torch.bfloat16).view(nkb, tbw, 3, tbh, 16, 32).float()
    wsum = planes.sum(dim=2)                                           # h + m + l
    half = dplan.thin_fold(True).view(torch.float16).view(nkb, tbw, 1, tbh, 16, 32).float()[:, :, 0]
    assert (half - wsum).abs().max() <= 1e-3 * wsum.abs().max()
    un = wsum.view(nkb, tbw, tbh, 16, 4, 8).clone()                    # undo the chunk swap of rows 8-15
    un[..., 8:, 0, :], un[..., 8:, 2, :] = wsum.view(nkb, tbw, tbh, 16, 4, 8)[..., 8:, 2, :], wsum.view(nkb, tbw, tbh, 16, 4, 8)[..., 8:, 0, :]
    un[..., 8:, 1, :], un[..., 8:, 3, :] = wsum.view(nkb, tbw, tbh, 16, 4, 8)[..., 8:, 3, :], wsum.view(nkb, tbw, tbh, 16, 4, 8)[..., 8:, 1, :]
    wf = un.view(nkb, tbw, tbh, 16, 32).permute(3, 2, 1, 0, 4).reshape(16, tbh, tbw, dplan.cin_p)   # [row][dyi][dxi][c]
    g = nhwc(gy, dplan.cin_p)                                          # [B, Hm, Wm, cin_p] (class grid = its input grid)
    hm, wm = (h + S - 1) // S, (w + S - 1) // S                        # class grid; beyond gy: zeros
    gp = F.pad(g, (0, 0, max(-dx0, 0) + 4, dx1 + 4 + wm - g.shape[2], max(-dy0, 0) + 4, dy1 + 4 + hm - g.shape[1]))
    oy_, ox_ = max(-dy0, 0) + 4, max(-dx0, 0) + 4
    out = torch.zeros(2, h, w, 4)
    for cls in range(S * S):
        acc = torch.zeros(2, hm, wm, 4)
        for dyi in range(tbh):
            for dxi in range(tbw):
                patch = gp[:, oy_ + dy0 + dyi:oy_ + dy0 + dyi + hm, ox_ + dx0 + dxi:ox_ + dx0 + dxi + wm]
                acc += torch.einsum('bhwc,nc->bhwn', patch, wf[4 * cls:4 * cls + 4, dyi, dxi])
        cy, cx = cls // S, cls % S
        out[:, cy::S, cx::S] = acc[:, :(h - cy + S - 1) // S, :(w - cx + S - 1) // S]
    assert torch.allclose(nchw(out)[:, :ci], x.grad, atol=1e-3), float((nchw(out)[:, :ci] - x.grad).abs().max())
    assert (out[..., ci:] == 0).all()


@pytest.mark.parametrize('pad', [1, 0, 2])
def test_winograd_form_all_paddings(pad):
    """The Winograd F(2x2,3x3) form of a 3x3 / s1 plan (convplan.attach_winograd: U = G g G^T as a 16-'tap' matrix W[n][pos * Cin + c],
    run by csrc/tapconv_wino.hip) restated on the CPU: V = B^T d B of the 4 x 4 input patch whose origin is (2 ty - pad, 2 tx - pad),
    M[pos] = sum_c U[pos] V[pos], Y = A^T M A -- for the same-size layer (pad 1), the unpadded layer (pad 0: taps 0..2, output 2
    smaller) and its input gradient (pad 2: taps -2..0, output 2 larger), against torch's conv2d / autograd."""
    h, w = 9, 11
    torch.manual_seed(pad)
    if pad == 2:      # the unpadded layer 64 -> 32 read backwards: its output gradient [32, h, w] -> input gradient [64, h + 2, w + 2]
        wt = torch.randn(32, 64, 3, 3, dtype=torch.float64)
        src = torch.randn(2, 32, h, w, dtype=torch.float64)
        xin = torch.randn(2, 64, h + 2, w + 2, dtype=torch.float64, requires_grad=True)
        F.conv2d(xin, wt, None, 1, 0).backward(src)
        ref, plan = xin.grad, cp.conv_dgrad_plan(wt.float(), 1, 0, device='cpu')
    else:
        wt = torch.randn(64, 32, 3, 3, dtype=torch.float64)
        src = torch.randn(2, 32, h, w, dtype=torch.float64)
        ref, plan = F.conv2d(src, wt, None, 1, pad), cp.conv_fwd_plan(wt.float(), None, 1, pad, device='cpu')
    assert plan.wino is not None and plan.wino_pad == pad
    n_in, n_out = src.shape[1], ref.shape[1]
    kp = plan.wino.cls[0]['Kpad']
    U = plan.wino.weights[:plan.wino._npad * kp].view(plan.wino._npad, kp)[:n_out, :16 * n_in].double().view(n_out, 16, n_in)
    BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
    AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
    ho, wo = ref.shape[2], ref.shape[3]
    big = F.pad(src, (pad, pad + 3, pad, pad + 3))          # zeros around: patches past the border read zeros, as the DMA does
    out = torch.zeros(2, n_out, ho + 1, wo + 1, dtype=torch.float64)
    for ty in range((ho + 1) // 2):
        for tx in range((wo + 1) // 2):
            d = big[:, :, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                      # origin (2 ty - pad, 2 tx - pad) of the image
            V = torch.einsum('ij,bcjk,lk->bcil', BT, d, BT).reshape(2, n_in, 16)
            M = torch.einsum('npc,bcp->bnp', U, V).reshape(2, n_out, 4, 4)
            out[:, :, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = torch.einsum('ij,bnjk,lk->bnil', AT, M, AT)
    assert torch.allclose(out[:, :, :ho, :wo], ref.detach(), atol=1e-5)


def test_small_linear_plan_dispatch():
    """convplan.SmallLinearPlan (the classifiers' last layer at batch <= 64 on csrc/linear_small.hip): which calls take the dedicated
    kernel and which the wrapped 1 x 1 convolution plan; the wrapped plan's attributes stay reachable (flops, tune keys)."""
    w, b = torch.randn(12, 16), torch.randn(12)
    plan = cp.linear_fwd_plan(w, b, device='cpu')
    assert isinstance(plan, cp.SmallLinearPlan) and plan.n == 12 and plan.k == 16 and plan.cout == 12
    x, y = torch.zeros(3, 1, 1, 16), torch.zeros(3, 1, 1, 12)
    assert plan.applies(x, y, {})
    assert not plan.applies(x, y, {'act': 1})                                    # fused epilogue: the convolution tiles
    assert not plan.applies(x.half(), y, {})                                     # fp16 storage
    assert not plan.applies(torch.zeros(65, 1, 1, 16), torch.zeros(65, 1, 1, 12), {})   # more rows than the kernel is meant for
    assert not plan.applies(torch.zeros(3, 1, 1, 20), y, {})                     # a wider buffer (channel window)
    assert plan.flops(3, 1, 1) == plan.conv.flops(3, 1, 1)
    assert not isinstance(cp.linear_dgrad_plan(torch.randn(10, 16), device='cpu'), cp.SmallLinearPlan)   # 10 outputs: rows not 16-byte aligned
