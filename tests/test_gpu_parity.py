"""GPU (-m gpu): parity of the HIP path (through the C-ABI library) with the oracle and the golden fixtures.

Tolerances.  north_star asks for 1e-4 relative L-inf on the produced projection image.  The SPAA loop is a chaotic,
discontinuous map (normalised-gradient steps of length 1-2 through ReLU networks, hard masks): the REFERENCE ITSELF
changes its 50-iteration output by ~2e-1 relative L-inf when run with 1 instead of 8 CPU threads or when the
initial image is perturbed by one ulp (fixture tests/golden/sensitivity_64.npz, produced by the unmodified reference;
tests/test_oracle_golden.py::test_reference_sensitivity_envelope checks it and re-measures it with the oracle).
So parity is asserted where it is well defined:
  * every forward quantity                                        <= 1e-4 (typically 1e-6) relative L-inf
  * gradients                                                     <= 1e-4 relative L2 (ReLU-gate flips of
    activations within rounding of 0 give sparse O(1) element differences, bounded by count)
  * ONE iteration from identical state (teacher forcing along the oracle's trajectory)  <= 1e-4 relative L-inf
  * full runs: exact where the trajectory has no knife edge (Q7 gray output), otherwise within the reference's own
    measured sensitivity envelope.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import spaa_oracle as so
from spaa_amd import synthetic as syn
from tapconv_emu import nhwc, nchw

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_inf(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def outlier_fraction(a, b, tol):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs() > tol * b.abs().max()).float().mean().item()


@pytest.fixture(scope='module')
def hip():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from spaa_amd import _lib, convplan, models, classifier, projector_based_attack, differential_color_functions
    _lib.load()  # raises if the HIP library is missing: there is no fallback
    return dict(lib=_lib, cp=convplan, models=models, clf=classifier, attack=projector_based_attack,
                dcf=differential_color_functions)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def make_pcnet(hip, sd, cam_sz, use_rough=True):
    m = hip['models']
    pc = m.PCNet(sd['mask'], m.WarpingNet(out_size=tuple(cam_sz)), use_rough=use_rough)
    pc.load_state_dict(sd)
    return pc.to(DEV)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('ci,co,k,s,p,h,w,b', [(3, 32, 3, 2, 1, 32, 32, 3), (6, 32, 3, 2, 1, 16, 24, 2),
                                               (32, 64, 3, 2, 1, 31, 33, 2), (128, 256, 3, 1, 1, 16, 16, 3),
                                               (256, 128, 3, 1, 1, 9, 17, 2), (32, 64, 1, 1, 0, 24, 24, 2),
                                               (32, 3, 3, 1, 1, 32, 32, 2), (3, 64, 7, 2, 3, 56, 56, 2),
                                               (64, 128, 1, 2, 0, 28, 28, 2), (512, 512, 3, 1, 1, 7, 7, 5)])
def test_tapconv_conv_forward_and_dgrad(hip, ci, co, k, s, p, h, w, b):
    cp = hip['cp']
    torch.manual_seed(ci * 1000 + co)
    x = torch.randn(b, ci, h, w, requires_grad=True)
    wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
    bias = torch.randn(co)
    y = F.conv2d(x, wt, bias, s, p)
    plan = cp.conv_fwd_plan(wt, bias, s, p, DEV)
    out = torch.zeros(b, y.shape[2], y.shape[3], (co + 3) // 4 * 4, device=DEV)
    plan.run(nhwc(x.detach(), plan.cin_p).to(DEV), out)
    assert rel_inf(nchw(out.cpu(), co), y) < 1e-5
    gy = torch.randn_like(y)
    y.backward(gy)
    dplan = cp.conv_dgrad_plan(wt, s, p, DEV)
    gx = torch.zeros(b, h, w, (ci + 3) // 4 * 4, device=DEV)
    dplan.run(nhwc(gy, dplan.cin_p).to(DEV), gx)
    assert rel_inf(nchw(gx.cpu(), ci), x.grad) < 1e-5


@pytest.mark.parametrize('ci,co,k,p,op,h,w', [(128, 64, 3, 1, 1, 16, 16), (64, 32, 2, 0, 0, 16, 12), (32, 2, 2, 0, 0, 8, 8)])
def test_tapconv_deconv_forward_and_dgrad(hip, ci, co, k, p, op, h, w):
    cp = hip['cp']
    torch.manual_seed(7)
    x = torch.randn(2, ci, h, w, requires_grad=True)
    wt = torch.randn(ci, co, k, k) / (ci * k * k) ** 0.5
    bias = torch.randn(co)
    y = F.conv_transpose2d(x, wt, bias, 2, p, op)
    plan = cp.deconv_fwd_plan(wt, bias, 2, p, DEV)
    out = torch.zeros(2, y.shape[2], y.shape[3], (co + 3) // 4 * 4, device=DEV)
    plan.run(nhwc(x.detach(), plan.cin_p).to(DEV), out)
    assert rel_inf(nchw(out.cpu(), co), y) < 1e-5
    gy = torch.randn_like(y)
    y.backward(gy)
    dplan = cp.deconv_dgrad_plan(wt, 2, p, DEV)
    gx = torch.zeros(2, h, w, ci, device=DEV)
    dplan.run(nhwc(gy, dplan.cin_p).to(DEV), gx)
    assert rel_inf(nchw(gx.cpu(), ci), x.grad) < 1e-5


@pytest.mark.parametrize('tile', [9, 10, 11, 28, 29, 38, 47])
def test_directconv_thin_layers(hip, tile):
    """The VALU variants for thin layers (few output or few input channels) compute the same tap-list convolution."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(11)
    cases = [(32, 3, 3, 1, 1, 20, 24), (64, 3, 7, 2, 3, 28, 28), (6, 32, 3, 2, 1, 16, 16)] if tile == 9 else \
        [(3, 32, 3, 1, 1, 20, 24), (3, 32, 3, 2, 1, 16, 16), (6, 32, 3, 2, 1, 12, 20)]
    if tile in (11, 28, 29, 47):  # (the dgrad of ci=3 layers has 32/64 gradient channels in and 3 out: the thin-N case)
        cases = [(32, 3, 3, 1, 1, 21, 24), (64, 3, 7, 2, 3, 28, 30), (3, 32, 3, 2, 1, 16, 18), (3, 64, 7, 2, 3, 30, 28),
                 (32, 4, 3, 1, 1, 37, 70), (64, 2, 3, 1, 1, 9, 33)]
    try:
        for ci, co, k, s, p, h, w in cases:
            x = torch.randn(2, ci, h, w, requires_grad=True)
            wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
            bias = torch.randn(co)
            y = F.conv2d(x, wt, bias, s, p)
            gy = torch.randn_like(y)
            y.backward(gy)
            fplan, dplan = cp.conv_fwd_plan(wt, bias, s, p, DEV), cp.conv_dgrad_plan(wt, s, p, DEV)
            out = torch.zeros(2, y.shape[2], y.shape[3], (co + 3) // 4 * 4, device=DEV)
            gx = torch.zeros(2, h, w, (ci + 3) // 4 * 4, device=DEV)
            add = torch.randn(2, co, y.shape[2], y.shape[3])
            cp.FORCE_TILE = tile
            fplan.run(nhwc(x.detach(), fplan.cin_p).to(DEV), out, add=nhwc(add, out.shape[3]).to(DEV), act=lib.ACT_RELU)
            dplan.run(nhwc(gy, dplan.cin_p).to(DEV), gx)
            cp.FORCE_TILE = 0
            # (a tile that does not apply to a layer falls back to the MFMA kernel inside ConvPlan.run: still checked)
            assert rel_inf(nchw(out.cpu(), co), F.relu(y + add)) < 1e-5
            assert rel_inf(nchw(gx.cpu(), ci), x.grad) < 1e-5
    finally:
        cp.FORCE_TILE = 0


def test_c3conv_first_layers(hip):
    """csrc/tapconv_c3.hip (tile 76): the first convolution of the classifier bodies (classifier.py:21-33 of the reference: ResNet-18
    7 x 7 / stride 2, VGG-16 3 x 3, Inception-v3 3 x 3 / stride 2 without padding) from a 3-channel NHWC4 image -- K = 3 x taps
    products only, patch and weights staged once -- against conv2d in float64: fp32-accurate (bf16x6, exact operands), fp32 and fp16
    output, bias + ReLU + gate bytes, ragged tiles, 32 / 48 / 64 output channels."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(76)
    try:
        for co, k, s, p, h, w in [(64, 3, 1, 1, 20, 37), (64, 7, 2, 3, 40, 70), (32, 3, 2, 0, 35, 41), (48, 3, 1, 1, 9, 33), (64, 7, 2, 3, 64, 64),
                                  (64, 5, 1, 2, 17, 19)]:
            x = torch.randn(2, 3, h, w)
            wt = torch.randn(co, 3, k, k) / (3 * k * k) ** 0.5
            bias = torch.randn(co)
            ref = F.relu(F.conv2d(x.double(), wt.double(), bias.double(), s, p)).float()
            plan = cp.conv_fwd_plan(wt, bias, s, p, DEV)
            assert plan.c3_ok()
            # fp16 output, round 5: the operands rounded to fp16 as well (the image in registers, one fp16 weight plane: what every layer
            # of fp16-storage mode multiplies) -- against float64 on the SAME fp16 operands; `c3h` disabled: the exact-operand form
            ref_h = F.relu(F.conv2d(x.half().double(), wt.half().double(), bias.double(), s, p)).float()
            for dt, tol, form in ((torch.float32, 2e-6, None), (torch.float16, 1e-3, 'exact'), (torch.float16, 1e-3, 'half')):
                out = torch.zeros(2, ref.shape[2], ref.shape[3], co, device=DEV, dtype=dt)
                mask = torch.zeros(2, ref.shape[2], ref.shape[3], co // 4, device=DEV, dtype=torch.uint8)
                (cp.DEFAULT_DISABLE.add if form == 'exact' else cp.DEFAULT_DISABLE.discard)('c3h')
                cp.FORCE_TILE = 76
                plan.run(nhwc(x, 4).to(DEV), out, act=lib.ACT_RELU, mask_out=mask)
                cp.FORCE_TILE = 0
                assert plan.last_tile == 76, plan.last_tile
                want = ref_h if form == 'half' else ref
                assert rel_inf(nchw(out.float().cpu(), co), want) < tol, (co, k, s, dt, form, rel_inf(nchw(out.float().cpu(), co), want))
                if form == 'half':     # ... and within the operands' rounding of the unrounded layer
                    assert rel_inf(nchw(out.float().cpu(), co), ref) < 4e-3, (co, k, s, rel_inf(nchw(out.float().cpu(), co), ref))
                assert torch.equal(mask.cpu(), lib.pack_gate_mask(out.float().cpu()))
    finally:
        cp.FORCE_TILE = 0
        cp.DEFAULT_DISABLE.discard('c3h')


def test_smallcin_two_output_halves(hip):
    """csrc/smallcin.hip with 33..64 output channels (VGG-16's first layer, classifier.py:21-24): two 32-channel halves over one
    staged patch; fp32 and fp16 output, residual + ReLU + gate bytes, ragged tiles, a channel count that is not a multiple of 32."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(38)
    try:
        for ci, co, k, s, p, h, w in [(3, 64, 3, 1, 1, 20, 37), (3, 48, 3, 1, 1, 9, 33), (6, 64, 3, 2, 1, 16, 22), (3, 64, 3, 1, 0, 12, 40)]:
            x = torch.randn(2, ci, h, w)
            wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
            bias = torch.randn(co)
            y = F.conv2d(x, wt, bias, s, p)
            add = torch.randn_like(y)
            ref = F.relu(y + add)
            plan = cp.conv_fwd_plan(wt, bias, s, p, DEV)
            for dt, tol in ((torch.float32, 1e-5), (torch.float16, 2e-3)):
                out = torch.zeros(2, y.shape[2], y.shape[3], co, device=DEV, dtype=dt)
                mask = torch.zeros(2, y.shape[2], y.shape[3], co // 4, device=DEV, dtype=torch.uint8)
                cp.FORCE_TILE = 38
                plan.run(nhwc(x, plan.cin_p).to(DEV), out, add=nhwc(add, co).to(DEV).to(dt), act=lib.ACT_RELU, mask_out=mask)
                cp.FORCE_TILE = 0
                assert plan.last_tile == 38, plan.last_tile
                assert rel_inf(nchw(out.float().cpu(), co), ref) < tol
                assert torch.equal(mask.cpu(), hip['lib'].pack_gate_mask(out.float().cpu()))
    finally:
        cp.FORCE_TILE = 0


@pytest.mark.parametrize('tile', [12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 30, 31, 32, 33, 34, 35, 36, 37, 225, 431, 234, 436, 39, 40, 41, 42, 43, 44, 45, 46, 242, 48, 49, 50, 51, 52, 53, 54, 248, 450, 948, 950, 952, 954])
def test_tapconv_x6_is_fp32_accurate(hip, tile):
    """fp32 emulated on the bf16 matrix cores (exact 3-way operand split, 6 of 9 partial products, separate
    accumulators): the error against an fp64 reference must not exceed that of the exact-fp32 MFMA kernel."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(21)
    try:
        for ci, co, k, s, p, h, w, b in [(128, 256, 3, 1, 1, 16, 16, 3), (64, 96, 3, 2, 1, 17, 19, 2), (32, 64, 1, 1, 0, 24, 24, 2),
                                         (6, 32, 3, 2, 1, 16, 24, 2), (256, 128, 3, 1, 1, 9, 17, 2), (3, 64, 7, 2, 3, 40, 40, 2)]:
            x = torch.relu(torch.randn(b, ci, h, w)) * (1 + 3 * torch.rand(b, ci, 1, 1))   # activation-like, wide range
            wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
            bias = torch.randn(co)
            truth = F.conv2d(x.double(), wt.double(), bias.double(), s, p)
            gy = torch.randn(b, co, truth.shape[2], truth.shape[3])
            g_truth = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), s, p)
            fplan, dplan = cp.conv_fwd_plan(wt, bias, s, p, DEV), cp.conv_dgrad_plan(wt, s, p, DEV)
            errs = {}
            for t in (6, tile):
                out = torch.zeros(b, truth.shape[2], truth.shape[3], (co + 3) // 4 * 4, device=DEV)
                gx = torch.zeros(b, h, w, (ci + 3) // 4 * 4, device=DEV)
                cp.FORCE_TILE = t
                fplan.run(nhwc(x, fplan.cin_p).to(DEV), out)
                dplan.run(nhwc(gy, dplan.cin_p).to(DEV), gx)
                cp.FORCE_TILE = 0
                e_f = (nchw(out.cpu(), co).double() - truth).abs().max().item() / truth.abs().max().item()
                e_d = (nchw(gx.cpu(), ci).double() - g_truth).abs().max().item() / g_truth.abs().max().item()
                errs[t] = (e_f, e_d)
            print(f'ci={ci} co={co} k={k} s={s}: rel err vs fp64  fp32-MFMA fwd {errs[6][0]:.1e} dgrad {errs[6][1]:.1e} | '
                  f'bf16x6 fwd {errs[tile][0]:.1e} dgrad {errs[tile][1]:.1e}')
            assert errs[tile][0] < max(2 * errs[6][0], 3e-7) and errs[tile][1] < max(2 * errs[6][1], 3e-7)
    finally:
        cp.FORCE_TILE = 0


@pytest.mark.parametrize('ci,co,h,w,b', [(128, 256, 64, 64, 2), (256, 128, 30, 44, 3), (64, 128, 17, 35, 2), (64, 192, 16, 32, 1),
                                         (256, 256, 14, 14, 64), (512, 512, 7, 7, 64)])   # (the last two: ResNet-18 layer3 / layer4 at the benchmark's batch -- canvas layout + K ranges)
def test_winograd_is_fp32_accurate(hip, ci, co, h, w, b):
    """3x3 / s1 / p1 layers through Winograd F(2x2,3x3) on the bf16x6 arithmetic (csrc/tapconv_wino.hip): forward and input
    gradient against fp64.  The transforms add roundings: the bound is 3x the exact-fp32 MFMA kernel's error (measured ~2x)."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(ci + co)
    x = torch.relu(torch.randn(b, ci, h, w)) * (1 + 3 * torch.rand(b, ci, 1, 1))
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5
    bias = torch.randn(co)
    truth = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1)
    gy = torch.randn(b, co, h, w)
    g_truth = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), 1, 1)
    fplan, dplan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV), cp.conv_dgrad_plan(wt, 1, 1, DEV)
    assert fplan.wino is not None and dplan.wino is not None
    errs = {}
    try:
        for t in (6, 34, 70):
            out = torch.zeros(b, h, w, co, device=DEV)
            gx = torch.zeros(b, h, w, ci, device=DEV)
            cp.FORCE_TILE = t
            fplan.run(nhwc(x, ci).to(DEV), out)
            dplan.run(nhwc(gy, co).to(DEV), gx)
            cp.FORCE_TILE = 0
            errs[t] = ((nchw(out.cpu(), co).double() - truth).abs().max().item() / truth.abs().max().item(),
                       (nchw(gx.cpu(), ci).double() - g_truth).abs().max().item() / g_truth.abs().max().item())
    finally:
        cp.FORCE_TILE = 0
    print(f'ci={ci} co={co} {h}x{w}: rel err vs fp64 fp32-MFMA {errs[6][0]:.1e}/{errs[6][1]:.1e}  bf16x6 {errs[34][0]:.1e}/{errs[34][1]:.1e}  '
          f'winograd {errs[70][0]:.1e}/{errs[70][1]:.1e}')
    assert errs[70][0] < max(3 * errs[6][0], 6e-7) and errs[70][1] < max(3 * errs[6][1], 6e-7)


@pytest.mark.parametrize('ci,co,h,w,b', [(64, 96, 23, 37, 3), (80, 192, 33, 31, 2), (96, 192, 73, 73, 2)])   # (80 -> 192 = Conv2d_4a: its 80 input channels keep the forward off this kernel, the gradient's 192 do not)
@pytest.mark.parametrize('tile', [70, 73])
def test_winograd_unpadded_layers(hip, ci, co, h, w, b, tile):
    """An UNPADDED 3x3 / s1 convolution (Inception-v3's Conv2d_2a / Conv2d_4a: classifier.py:29-33 of the reference) and its
    input gradient through the Winograd kernel: the output is 2 smaller (pad 0) resp. 2 larger (pad 2) than the input, which only
    moves the patch origin (csrc/tapconv_wino.hip: `reserved0` bits 27-28).  Against fp64 at the padded layers' bound, bias + ReLU +
    byte mask on the forward, byte-mask gate on the gradient."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(ci + co + h)
    x = torch.relu(torch.randn(b, ci, h, w))
    wt, bias = torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5, torch.randn(co)
    truth = F.relu(F.conv2d(x.double(), wt.double(), bias.double(), 1, 0))
    gy = torch.randn(b, co, h - 2, w - 2)
    keep = torch.rand(b, ci, h, w) > 0.3
    g_truth = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), 1, 0) * keep
    fplan, dplan = cp.conv_fwd_plan(wt, bias, 1, 0, DEV), cp.conv_dgrad_plan(wt, 1, 0, DEV)
    assert (fplan.wino is not None and fplan.wino_pad == 0) == (ci % 32 == 0) and dplan.wino is not None and dplan.wino_pad == 2
    errs = {}
    try:
        for t in (34, tile):
            cp.FORCE_TILE = t
            out = torch.zeros(b, h - 2, w - 2, co, device=DEV)
            m_out = torch.zeros(b, h - 2, w - 2, co // 4, dtype=torch.uint8, device=DEV)
            fplan.run(nhwc(x, ci).to(DEV), out, act=lib.ACT_RELU, mask_out=m_out)
            gx = torch.zeros(b, h, w, ci, device=DEV)
            dplan.run(nhwc(gy, co).to(DEV), gx, gate_bits=lib.pack_gate_mask(nhwc(keep.float(), ci).to(DEV)))
            if t == tile:
                assert (fplan.wino is None or fplan.wino.last_tile == tile) and dplan.wino.last_tile == tile and torch.equal(m_out, lib.pack_gate_mask(out))
            errs[t] = (((nchw(out.cpu(), co).double() - truth).abs().max() / truth.abs().max()).item(),
                       ((nchw(gx.cpu(), ci).double() - g_truth).abs().max() / g_truth.abs().max()).item())
    finally:
        cp.FORCE_TILE = 0
    print(f'unpadded {ci}->{co} {h}x{w}: rel err vs fp64 direct bf16x6 {errs[34][0]:.1e}/{errs[34][1]:.1e}, winograd {errs[tile][0]:.1e}/{errs[tile][1]:.1e}')
    assert errs[tile][0] < max(3 * errs[34][0], 6e-7) and errs[tile][1] < max(3 * errs[34][1], 6e-7)


@pytest.mark.parametrize('ci,co,h,w,b,tile,ks', [(256, 256, 14, 14, 64, 70, 0), (512, 512, 7, 7, 64, 70, 0), (64, 96, 35, 35, 9, 70, 0),
                                                 (128, 128, 14, 14, 5, 71, 2), (96, 64, 17, 17, 7, 70, 3), (64, 128, 7, 9, 3, 70, 1),
                                                 (128, 256, 20, 38, 2, 70, 4), (64, 64, 15, 20, 10, 71, 2),
                                                 (256, 256, 14, 14, 64, 73, 0), (512, 512, 7, 7, 64, 73, 0), (128, 128, 14, 14, 5, 73, 2)])
def test_winograd_canvas_and_k_ranges(hip, ci, co, h, w, b, tile, ks):
    """csrc/tapconv_wino.hip, small images: the batch laid out on virtual canvases (workgroup regions tile the canvas, gap rows /
    columns are the zero padding) and K cut into ranges summed in fixed order by a second kernel.  Against the direct bf16x6
    kernel and torch on the CPU: every epilogue form (plain; bias + residual + ReLU + byte mask out; byte-mask gate; float gate =
    the generic path), channel windows of wider buffers, nothing written outside the window, bitwise run to run."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(ci + co + h)
    x, wt, bias = torch.randn(b, ci, h, w), torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5), torch.randn(co)
    add, gate = torch.randn(b, co, h, w), torch.randn(b, co, h, w)
    plan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV)
    ref = F.conv2d(x, wt, bias, 1, 1)
    old_nc = cp.DEBUG_WINO_NOCANVAS
    try:
        cp.DEBUG_WINO_NOCANVAS = 2     # canvas wherever it has fewer workgroup regions (the cost model may prefer image-aligned ones)
        cp.FORCE_TILE = tile + 100 * ks
        xin = nhwc(x).to(DEV)
        out = torch.zeros(b, h, w, co, device=DEV)
        plan.run(xin, out)
        wp = plan.wino.last_wino_plan
        print(f'{ci}->{co} {h}x{w} B={b}: N tile {wp[0]}, K ranges {wp[1]} x {wp[6]} blocks, canvas {wp[2]} ({wp[3]} x {wp[4]} images, {wp[7]} canvases), {wp[5]} workgroups')
        assert wp[2] == 1 and (ks == 0 or wp[1] == min(ks, ci // 32)) and plan.wino.last_tile == tile
        assert rel_inf(nchw(out.cpu()), ref) < 1e-5
        out2 = torch.zeros_like(out)
        plan.run(xin, out2)
        assert torch.equal(out, out2)
        # the image-aligned form of the same layer, no split: equal up to the summation order over K ranges
        cp.DEBUG_WINO_NOCANVAS = 1
        cp.FORCE_TILE = tile + 100
        out3 = torch.zeros_like(out)
        plan.run(xin, out3)
        assert plan.wino.last_wino_plan[2] == 0 and plan.wino.last_wino_plan[1] == 1
        assert rel_inf(out, out3) < 2e-6 and (wp[1] > 1 or torch.equal(out, out3))
        cp.DEBUG_WINO_NOCANVAS = 2
        cp.FORCE_TILE = tile + 100 * ks
        # bias + residual + ReLU + byte mask out (the branch-free epilogue; behind a K split: the second pass)
        mask = torch.zeros(b, h, w, co // 4, dtype=torch.uint8, device=DEV)
        plan.run(xin, out, add=nhwc(add).to(DEV), act=lib.ACT_RELU, mask_out=mask)
        assert rel_inf(nchw(out.cpu()), F.relu(ref + add)) < 1e-5 and torch.equal(mask, lib.pack_gate_mask(out))
        if wp[1] > 1:
            # round 6: the K ranges meeting INSIDE the kernel (SPAA_SPLITK_FIXUP=1: the last-arriving workgroup of a tile adds them in
            # fixed order, arrival counters at the head of the workspace; off by default -- measured slower, spaa_amd/convplan.py) is
            # bitwise the separate second pass, whichever workgroup arrives last, launch after launch on the same workspace (the
            # counters are left zero)
            keep = cp.WINO_SPLITK_FIXUP
            try:
                cp.WINO_SPLITK_FIXUP = True
                out_fx, mask_fx = torch.zeros_like(out), torch.zeros_like(mask)
                plan.run(xin, out_fx, add=nhwc(add).to(DEV), act=lib.ACT_RELU, mask_out=mask_fx)
                assert plan.wino._ws_fix is not None and plan.wino.last_wino_plan[1] == wp[1]
                assert torch.equal(out, out_fx) and torch.equal(mask, mask_fx)
                for _ in range(3):
                    plan.run(xin, out3)
                    assert torch.equal(out3, out2)
                assert int(plan.wino._ws_fix[:cp.SPLITK_HDR].view(torch.int32).abs().max()) == 0
            finally:
                cp.WINO_SPLITK_FIXUP = keep
        gbits = lib.pack_gate_mask(nhwc(gate).to(DEV))
        plan.run(xin, out, gate_bits=gbits)
        assert rel_inf(nchw(out.cpu()), ref * (gate > 0)) < 1e-5
        plan.run(xin, out, gate=nhwc(gate).to(DEV))          # float gate: the generic epilogue
        assert rel_inf(nchw(out.cpu()), ref * (gate > 0)) < 1e-5
        wide_in = torch.randn(b, h, w, ci + 32, device=DEV)
        wide_out = torch.full((b, h, w, co + 64), 7.0, device=DEV)
        plan.run(wide_in, wide_out, in_coff=32, out_coff=64)
        ref2 = F.conv2d(nchw(wide_in.cpu())[:, 32:], wt, bias, 1, 1)
        assert rel_inf(nchw(wide_out.cpu())[:, 64:], ref2) < 1e-5 and (wide_out[..., :64] == 7.0).all()
    finally:
        cp.FORCE_TILE = 0
        cp.DEBUG_WINO_NOCANVAS = old_nc


@pytest.mark.parametrize('kind,ci,co,h,w,b,ci2', [('deconv', 128, 64, 64, 64, 2, 0), ('deconv', 128, 64, 13, 21, 3, 32), ('deconv', 64, 32, 9, 40, 2, 0),
                                                  ('dgrad', 64, 32, 24, 24, 2, 64), ('dgrad', 64, 32, 17, 35, 3, 0), ('dgrad', 128, 64, 28, 28, 2, 0),
                                                  ('dgrad', 256, 128, 14, 14, 4, 0), ('deconv', 32, 128, 8, 8, 5, 32)])
def test_patch_staged_stride2_kernel(hip, kind, ci, co, h, w, b, ci2):
    """csrc/tapconv_x6p.hip (tile 74): ConvTranspose2d(k3, s2, p1, op1) forward and the input gradient of a k3 / s2 / p1 convolution as
    ONE launch over the four output-parity classes (input patch staged once, exactly the nine real (class, tap) products), with
    an optional fused 1 x 1 second source at output resolution (models.py:293,299: transConv1(x) + skipConv2(x1)).  Against
    fp64 at the exact-fp32 kernel's error level, every epilogue form, channel windows, bitwise run to run."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(ci + co + h + ci2)
    if kind == 'deconv':   # x [b, ci, h, w] -> [b, co, 2h, 2w]
        wt, bias = torch.randn(ci, co, 3, 3) / (ci * 2.25) ** 0.5, torch.randn(co)
        x = torch.relu(torch.randn(b, ci, h, w))
        plan = cp.deconv_fwd_plan(wt, bias, 2, 1, DEV, 'tc', fold=False)
        truth = F.conv_transpose2d(x.double(), wt.double(), bias.double(), 2, 1, 1)
        ho, wo = 2 * h, 2 * w
    else:                  # gradient of conv(co -> ci channels ... ) : consumes [b, ci, h, w], produces [b, co, ~2h, ~2w]
        ho, wo = 2 * h - (h % 2), 2 * w          # (an odd and an even output size)
        wt, bias = torch.randn(ci, co, 3, 3) / (ci * 2.25) ** 0.5, None
        x = torch.randn(b, ci, h, w)
        assert (ho + 2 - 3) // 2 + 1 == h and (wo + 2 - 3) // 2 + 1 == w
        plan = cp.conv_dgrad_plan(wt, 2, 1, DEV, 'dg', fold=False)
        truth = torch.nn.grad.conv2d_input((b, co, ho, wo), wt.double(), x.double(), 2, 1)
    assert plan.x6p_ok()
    x2 = w2 = None
    if ci2:
        w2, b2 = torch.randn(co, ci2, 1, 1) / ci2 ** 0.5, torch.randn(co)
        x2 = torch.randn(b, ci2 + 32, ho, wo)          # (a channel window of a wider tensor)
        plan.attach_second_source(w2, b2)
        truth = truth + F.conv2d(x2[:, 32:].double(), w2.double(), b2.double())
    kw = dict(inp2=nhwc(x2).to(DEV), in2_coff=32) if ci2 else {}
    xin = nhwc(x).to(DEV)
    errs = {}
    try:
        for t in ((74,) if ci2 else (6, 34, 74)):
            cp.FORCE_TILE = t
            out = torch.zeros(b, ho, wo, co, device=DEV)
            plan.run(xin, out, **kw)
            assert plan.last_tile == t
            errs[t] = (nchw(out.cpu()).double() - truth).abs().max().item() / truth.abs().max().item()
        print(f'{kind} {ci}->{co} {h}x{w} second source {ci2}: rel err vs fp64 ' + ' '.join(f'tile {t}: {e:.1e}' for t, e in errs.items()))
        assert errs[74] < (max(2 * errs[6], 3e-7) if not ci2 else 2e-6)
        cp.FORCE_TILE = 74
        ref = truth.float()
        out = torch.zeros(b, ho, wo, co, device=DEV)
        plan.run(xin, out, **kw)
        out2 = torch.zeros_like(out)
        plan.run(xin, out2, **kw)
        assert torch.equal(out, out2)
        # the compile-time schedule (STD: these layers have the canonical tap order) against the run-time one: the same products in
        # the same order, bit for bit
        assert plan.x6p_canonical() and cp.X6P_STD
        cp.X6P_STD = False
        try:
            out3 = torch.zeros_like(out)
            plan.run(xin, out3, **kw)
        finally:
            cp.X6P_STD = True
        assert torch.equal(out, out3)
        add, gate = torch.randn(b, co, ho, wo), torch.randn(b, co, ho, wo)
        mask = torch.zeros(b, ho, wo, co // 4, dtype=torch.uint8, device=DEV)
        plan.run(xin, out, add=nhwc(add).to(DEV), act=lib.ACT_RELU, mask_out=mask, **kw)
        assert rel_inf(nchw(out.cpu()), F.relu(ref + add)) < 1e-5 and torch.equal(mask, lib.pack_gate_mask(out))
        gbits = lib.pack_gate_mask(nhwc(gate).to(DEV))
        plan.run(xin, out, gate_bits=gbits, **kw)
        assert rel_inf(nchw(out.cpu()), ref * (gate > 0)) < 1e-5
        plan.run(xin, out, gate=nhwc(gate).to(DEV), **kw)          # float gate: the generic epilogue
        assert rel_inf(nchw(out.cpu()), ref * (gate > 0)) < 1e-5
        wide_in = torch.zeros(b, h, w, ci + 32, device=DEV)
        wide_in[..., 32:] = xin
        wide_in[..., :32] = 3.0
        wide_out = torch.full((b, ho, wo, co + 64), 7.0, device=DEV)
        plan.run(wide_in, wide_out, in_coff=32, out_coff=64, **kw)
        assert rel_inf(nchw(wide_out.cpu())[:, 64:], ref) < 1e-5 and (wide_out[..., :64] == 7.0).all()
    finally:
        cp.FORCE_TILE = 0


@pytest.mark.parametrize('ci,co,ci2', [(64, 32, 0), (64, 32, 64), (128, 64, 32)])
def test_patch_staged_stride2_full_size_is_reproducible(hip, ci, co, ci2):
    """Tile 74 at the benchmark's sizes (batch 64, 64 x 64 class grid: 2048 workgroups, two per compute unit, short K): every
    image equal to the implicit-GEMM tile's result and bitwise run to run.  (A counted `vmcnt` that let a combo's weights stay in
    flight corrupted one or two workgroups of 2048 per launch -- only at this size, only sometimes: caught by
    test_benchmarked_configuration_first_iteration, pinned here.)"""
    cp = hip['cp']
    torch.manual_seed(ci)
    b = 64
    wt = torch.randn(ci, co, 3, 3) / (ci * 2.25) ** 0.5
    plan = cp.conv_dgrad_plan(wt, 2, 1, DEV, 'dg', fold=False)
    x = torch.randn(b, 64, 64, ci, device=DEV)
    add = torch.randn(b, 128, 128, co, device=DEV)
    gb = (torch.rand(b, 128, 128, co // 4, device=DEV) * 16).to(torch.uint8)
    kw = {}
    if ci2:
        plan.attach_second_source(torch.randn(co, ci2) / ci2 ** 0.5, None)
        x2 = torch.randn(b, 128, 128, ci2, device=DEV)
        kw = dict(inp2=x2)
    try:
        outs = []
        for rep in range(4):
            cp.FORCE_TILE = 74
            o = torch.zeros(b, 128, 128, co, device=DEV)
            plan.run(x, o, add=add, gate_bits=gb, **kw)
            outs.append(o)
        assert all(torch.equal(outs[0], o) for o in outs[1:])
        if not ci2:
            cp.FORCE_TILE = 30
            ref = torch.zeros(b, 128, 128, co, device=DEV)
            plan.run(x, ref, add=add, gate_bits=gb)
            d = (ref - outs[0]).abs().flatten(1).max(dim=1).values / ref.abs().max()
            assert d.max() < 2e-6, [i for i, v in enumerate(d.tolist()) if v > 2e-6]
    finally:
        cp.FORCE_TILE = 0


@pytest.mark.parametrize('ca,cb,co,h,w,b', [(256, 64, 128, 64, 64, 2), (128, 128, 64, 40, 70, 3), (64, 32, 192, 17, 33, 2)])
def test_winograd_two_sources(hip, ca, cb, co, h, w, b):
    """conv(a, Wa) + conv(b, Wb) as ONE Winograd launch over the concatenated input channels read from two tensors
    (cp.conv_fwd_plan_2src: `conv5(x4) + skipConv3(x2)`, models.py:294,298), and the mirror image for input gradients
    (cp.conv_dgrad_plan_2src): against fp64, epilogues, channel windows of wider tensors."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(ca + cb)
    xa, xb = torch.randn(b, ca, h, w), torch.randn(b, cb, h, w)
    wa, wb, bias = torch.randn(co, ca, 3, 3) / (3 * (ca + cb) ** 0.5), torch.randn(co, cb, 3, 3) / (3 * (ca + cb) ** 0.5), torch.randn(co)
    truth = F.conv2d(xa.double(), wa.double(), bias.double(), 1, 1) + F.conv2d(xb.double(), wb.double(), None, 1, 1)
    plan = cp.conv_fwd_plan_2src(wa, wb, bias, DEV, 'two')
    assert plan is not None
    wide_a = torch.randn(b, h, w, ca + 32, device=DEV)
    wide_a[..., 32:] = nhwc(xa).to(DEV)
    wide_b = torch.randn(b, h, w, cb + 64, device=DEV)
    wide_b[..., 64:] = nhwc(xb).to(DEV)
    out = torch.zeros(b, h, w, co, device=DEV)
    plan.run(wide_a, out, inp2=wide_b, in_coff=32, in2_coff=64)
    assert plan.wino.last_tile in (70, 71)
    e = (nchw(out.cpu()).double() - truth).abs().max().item() / truth.abs().max().item()
    print(f'two sources {ca}+{cb}->{co} {h}x{w}: rel err vs fp64 {e:.1e}, plan {plan.wino.last_wino_plan}')
    assert e < 6e-7
    add = torch.randn(b, co, h, w)
    mask = torch.zeros(b, h, w, co // 4, dtype=torch.uint8, device=DEV)
    plan.run(wide_a, out, inp2=wide_b, in_coff=32, in2_coff=64, add=nhwc(add).to(DEV), act=lib.ACT_RELU, mask_out=mask)
    assert rel_inf(nchw(out.cpu()), F.relu(truth.float() + add)) < 1e-5 and torch.equal(mask, lib.pack_gate_mask(out))
    with pytest.raises(ValueError):
        plan.run(wide_a, out, in_coff=32)                    # the second tensor is not optional
    # input gradients of two convolutions that read the same tensor: ga [co], gb [cb2] -> [ci]
    ci, co_a, co_b = 64, ca, cb
    wa2, wb2 = torch.randn(co_a, ci, 3, 3) / (3 * (ca + cb) ** 0.5), torch.randn(co_b, ci, 3, 3) / (3 * (ca + cb) ** 0.5)
    ga, gb = torch.randn(b, co_a, h, w), torch.randn(b, co_b, h, w)
    gt = (torch.nn.grad.conv2d_input((b, ci, h, w), wa2.double(), ga.double(), 1, 1)
          + torch.nn.grad.conv2d_input((b, ci, h, w), wb2.double(), gb.double(), 1, 1))
    dplan = cp.conv_dgrad_plan_2src(wa2, wb2, DEV, 'two_d')
    gx = torch.zeros(b, h, w, ci, device=DEV)
    gbits = (torch.rand(b, h, w, ci // 4, device=DEV) * 16).to(torch.uint8)
    dplan.run(nhwc(ga).to(DEV), gx, inp2=nhwc(gb).to(DEV))
    assert (nchw(gx.cpu()).double() - gt).abs().max().item() / gt.abs().max().item() < 6e-7
    dplan.run(nhwc(ga).to(DEV), gx, inp2=nhwc(gb).to(DEV), gate_bits=gbits)
    keep = torch.stack([(gbits.cpu() >> e_) & 1 for e_ in range(4)], -1).reshape(b, h, w, ci).permute(0, 3, 1, 2).bool()
    assert rel_inf(nchw(gx.cpu()), gt.float() * keep) < 1e-5


@pytest.mark.parametrize('ci,co,h,w,b', [(64, 64, 56, 56, 3), (128, 128, 28, 28, 5), (128, 64, 21, 45, 2), (64, 192, 9, 70, 2)])
def test_winograd_four_wave_workgroups(hip, ci, co, h, w, b):
    """Tile 73: the Winograd kernel with four-wave workgroups (8 x 32 output pixels, two workgroups per compute unit).  The same
    per-tile arithmetic in the same order as the eight-wave form with the 64-wide N tile (tile 71): bitwise equal, every epilogue."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(ci + h)
    x, wt, bias = torch.randn(b, ci, h, w), torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5), torch.randn(co)
    add, gate = torch.randn(b, co, h, w), torch.randn(b, co, h, w)
    plan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV)
    xin, addn = nhwc(x).to(DEV), nhwc(add).to(DEV)
    gbits = lib.pack_gate_mask(nhwc(gate).to(DEV))
    res = {}
    old_nc = cp.DEBUG_WINO_NOCANVAS
    try:
        cp.DEBUG_WINO_NOCANVAS = 1
        for tile in (171, 173):          # (+ 100: one K range -- a split changes the summation order)
            cp.FORCE_TILE = tile
            o1, o2, o3 = (torch.zeros(b, h, w, co, device=DEV) for _ in range(3))
            mask = torch.zeros(b, h, w, co // 4, dtype=torch.uint8, device=DEV)
            plan.run(xin, o1)
            assert plan.wino.last_tile == tile % 100
            plan.run(xin, o2, add=addn, act=lib.ACT_RELU, mask_out=mask)
            plan.run(xin, o3, gate=nhwc(gate).to(DEV))
            o4 = torch.zeros_like(o1)
            plan.run(xin, o4, gate_bits=gbits)
            res[tile] = (o1, o2, o3, o4, mask)
    finally:
        cp.FORCE_TILE = 0
        cp.DEBUG_WINO_NOCANVAS = old_nc
    for a, c in zip(res[171], res[173]):
        assert torch.equal(a, c)
    assert rel_inf(nchw(res[173][0].cpu()), F.conv2d(x, wt, bias, 1, 1)) < 1e-5


def test_winograd_epilogues_and_masks(hip):
    """The Winograd kernel shares the epilogue of the other bf16x6 kernels: bias, residual, ReLU, byte gate masks."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(5)
    b, ci, co, h, w = 2, 64, 128, 20, 38
    x, wt, bias = torch.randn(b, ci, h, w), torch.randn(co, ci, 3, 3) / 24, torch.randn(co)
    add, gate = torch.randn(b, co, h, w), torch.randn(b, co, h, w)
    plan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV)
    try:
        cp.FORCE_TILE = 70
        out, aux = torch.zeros(b, h, w, co, device=DEV), torch.zeros(b, h, w, co, device=DEV)
        mask = torch.zeros(b, h, w, co // 4, dtype=torch.uint8, device=DEV)
        plan.run(nhwc(x).to(DEV), out, add=nhwc(add).to(DEV), act=lib.ACT_RELU, mask_out=mask)
        ref = F.relu(F.conv2d(x, wt, bias, 1, 1) + add)
        assert rel_inf(nchw(out.cpu()), ref) < 1e-5
        assert torch.equal(mask, lib.pack_gate_mask(out))
        gbits = lib.pack_gate_mask(nhwc(gate).to(DEV))
        plan.run(nhwc(x).to(DEV), out, gate_bits=gbits)
        assert rel_inf(nchw(out.cpu()), F.conv2d(x, wt, bias, 1, 1) * (gate > 0)) < 1e-5
        plan.run(nhwc(x).to(DEV), out, act=lib.ACT_RELU_CLAMP1, aux_out=aux)
        pre = F.relu(F.conv2d(x, wt, bias, 1, 1))
        assert rel_inf(nchw(out.cpu()), pre.clamp(max=1)) < 1e-5 and rel_inf(nchw(aux.cpu()), pre) < 1e-5
        # a sub-range of a wider buffer (channel offsets), as the engine's concatenated buffers use
        wide_in = torch.randn(b, h, w, ci + 32, device=DEV)
        wide_out = torch.zeros(b, h, w, co + 64, device=DEV)
        plan.run(wide_in, wide_out, in_coff=32, out_coff=64)
        ref2 = F.conv2d(nchw(wide_in.cpu())[:, 32:], wt, bias, 1, 1)
        assert rel_inf(nchw(wide_out.cpu())[:, 64:], ref2) < 1e-5 and (wide_out[..., :64] == 0).all()
    finally:
        cp.FORCE_TILE = 0


@pytest.mark.parametrize('tile', [0, 6, 18, 34, 36, 39, 40, 41, 42, 44, 45, 46, 48, 50, 51, 52])
def test_tapconv_epilogues(hip, tile):
    cp, lib = hip['cp'], hip['lib']
    cp.FORCE_TILE = tile
    torch.manual_seed(3)
    x, wt, bias = torch.randn(2, 32, 12, 12), torch.randn(64, 32, 3, 3) / 17, torch.randn(64)
    add, gate, gate2 = torch.randn(2, 64, 12, 12), torch.randn(2, 64, 12, 12), torch.randn(2, 64, 12, 12)
    plan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV)
    out, aux = torch.zeros(2, 12, 12, 64, device=DEV), torch.zeros(2, 12, 12, 64, device=DEV)
    plan.run(nhwc(x).to(DEV), out, add=nhwc(add).to(DEV), act=lib.ACT_RELU, gate=nhwc(gate).to(DEV), aux_out=aux,
             gate2=nhwc(gate2).to(DEV))
    ref = F.relu(F.conv2d(x, wt, bias, 1, 1) + add) * (gate > 0)
    assert rel_inf(nchw(out.cpu()), ref) < 1e-5
    assert rel_inf(nchw(aux.cpu()), ref * (gate2 > 0)) < 1e-5
    plan.run(nhwc(x).to(DEV), out, act=lib.ACT_RELU_CLAMP1, aux_out=aux)
    pre = F.relu(F.conv2d(x, wt, bias, 1, 1))
    assert rel_inf(nchw(out.cpu()), pre.clamp(max=1)) < 1e-5 and rel_inf(nchw(aux.cpu()), pre) < 1e-5
    plan.run(nhwc(x).to(DEV), out, act=lib.ACT_LEAKY01)
    assert rel_inf(nchw(out.cpu()), F.leaky_relu(F.conv2d(x, wt, bias, 1, 1), 0.1)) < 1e-5
    # strided output classes (3x3 stride-2 transposed conv) with a residual, non-square, ragged against the tile
    xt, wtt = torch.randn(2, 32, 9, 11), torch.randn(32, 64, 3, 3) / 17
    addt = torch.randn(2, 64, 18, 22)
    tplan = cp.deconv_fwd_plan(wtt, bias, 2, 1, DEV)
    outt = torch.zeros(2, 18, 22, 64, device=DEV)
    tplan.run(nhwc(xt).to(DEV), outt, add=nhwc(addt).to(DEV), act=lib.ACT_RELU)
    assert rel_inf(nchw(outt.cpu()), F.relu(F.conv_transpose2d(xt, wtt, bias, 2, 1, 1) + addt)) < 1e-5
    cp.FORCE_TILE = 0
    with pytest.raises(AssertionError):  # shape mismatch is caught on the host, before any launch
        plan.run(nhwc(x).to(DEV), torch.zeros(2, 12, 12, 32, device=DEV))


# ---------------------------------------------------------------------------------------------------------------
def test_color_kernels_vs_reference_golden(hip, golden_dir):
    dcf = hip['dcf']
    z = load(golden_dir, 'color_kat')
    a, b = torch.from_numpy(z['rgb_a']).to(DEV), torch.from_numpy(z['rgb_b']).to(DEV)
    lab_a, lab_b = dcf.rgb2lab_diff(a), dcf.rgb2lab_diff(b)
    assert rel_inf(lab_a, torch.from_numpy(z['lab_a'])) < 1e-5 and rel_inf(lab_b, torch.from_numpy(z['lab_b'])) < 1e-5
    de = dcf.ciede2000_diff(lab_a, lab_b)
    assert (de.cpu() - torch.from_numpy(z['de'])).abs().max() < 2e-4  # dE units (0..100)
    s1 = torch.tensor([50., 2.6772, -79.7751]).view(1, 3, 1, 1).to(DEV)
    s2 = torch.tensor([50., 0., -82.7485]).view(1, 3, 1, 1).to(DEV)
    assert abs(dcf.ciede2000_diff(s1, s2).item() - 2.0213) < 2e-4  # reference's `aHP - 39`, not the textbook 2.0425
    l2, dE, g = dcf.stealth_loss_with_grad(a, b, 0.0, 1.0)
    assert np.allclose(dE.cpu().numpy(), z['de'].mean(axis=(1, 2)), rtol=1e-5)
    g = g.cpu() * (a.shape[2] * a.shape[3])
    g_ref = torch.from_numpy(z['grad_a'])
    fin = torch.isfinite(g_ref)  # the reference yields NaN on near-grey, near-black pairs (aC^7 underflows)
    # Chroma below ~1 Lab unit means a*, b* are cancellation residues (grey pixels: a* ~ 2e-3 from the 4-decimal
    # matrix): the hue angle and with it the reference's own gradient carry percent-level rounding noise there.
    lab_a_ref, lab_b_ref = torch.from_numpy(z['lab_a']), torch.from_numpy(z['lab_b'])
    chroma = torch.minimum(lab_a_ref[:, 1:].norm(dim=1), lab_b_ref[:, 1:].norm(dim=1))
    well = (chroma > 1.0)[:, None].expand_as(g_ref) & fin
    scale = g_ref[fin].abs().max()
    assert ((g - g_ref)[well].abs().max() / scale) < 1e-4
    assert ((g - g_ref)[fin].abs().max() / scale) < 5e-2
    assert (g[0, :, 1, :] == 0).all()  # identical pixels: exactly zero gradient
    l2b, _, g2 = dcf.stealth_loss_with_grad(a, b, 1.0, 0.0)
    ar = a.cpu().clone().requires_grad_(True)
    n = torch.norm(b.cpu() - ar, dim=1)
    n.sum().backward()
    assert np.allclose(l2b.cpu().numpy(), n.detach().mean(dim=(1, 2)).numpy(), rtol=1e-5)
    assert rel_inf(g2.cpu() * (a.shape[2] * a.shape[3]), ar.grad) < 1e-5


@pytest.mark.parametrize('name', ['pcnet_64', 'pcnet_nonsq', 'pcnet_256', 'pcnet_norough_64'])
def test_pcnet_forward_and_input_gradient(hip, golden_dir, name):
    """PCNet.forward(x, s) through the reference interface; x is white noise, the worst case for the fp32 sampling
    coordinates (1 ulp of a pixel coordinate ~1.5e-5 px).  `pcnet_norough_64`: PCNet(use_rough=False) (models.py:344-345),
    fixture from the reference's own module."""
    z = load(golden_dir, name)
    cam_sz = tuple(int(v) for v in z['cam_sz'])
    sd = syn.pcnet_state_dict(int(z['seed']), cam_sz=cam_sz, mask=str(z['mask']))
    rough = bool(z['use_rough']) if 'use_rough' in z.files else True
    if not rough:
        sd['shading_net.conv1_s.weight'] = sd['shading_net.conv1_s.weight'][:, :3].contiguous()
    pc = make_pcnet(hip, sd, cam_sz, rough)
    x = torch.from_numpy(z['x']).to(DEV).requires_grad_(True)
    y = pc(x, torch.from_numpy(z['s']).to(DEV))
    assert rel_inf(y, torch.from_numpy(z['y'])) < 1e-4
    eng = pc.engine(x.shape[0], x.shape[-2:])
    fg = torch.from_numpy(z['fine_grid'])[0]
    assert (eng.grid[..., :2].cpu() - fg).abs().max() < 1e-5
    (y * torch.from_numpy(z['r']).to(DEV)).sum().backward()
    g_ref = torch.from_numpy(z['grad_x'])
    if not rough:
        # gate-aware: with the reference's (== oracle's) ReLU / clamp gates in the engine, the input gradient agrees to rounding
        import gates
        xc, sc = torch.from_numpy(z['x']), torch.from_numpy(z['s'])
        with torch.no_grad():
            xw = so.warp(sd, xc, cam_sz) * sd['mask']
            _, acts = so.shading_net(sd, xw, (sc,), return_all=True)
        M = hip['models']
        eng.set_scene(M.to_nhwc4(sc.to(DEV)))
        eng.forward(M.to_nhwc4(xc.to(DEV)), clamp01=False)
        pairs = gates.pcnet_pairs(eng, acts)
        flips, per_layer = gates.count_flips(pairs)
        gates.inject(pairs, (eng,))
        r4 = M.to_nhwc4(torch.from_numpy(z['r']).to(DEV))
        gP = torch.zeros_like(r4)
        state = torch.ones(eng.B, 4, dtype=torch.int32, device=DEV)
        hip['lib'].call('spaa_select_grad', hip['lib'].ptr(r4), hip['lib'].ptr(r4), hip['lib'].ptr(state), hip['lib'].ptr(eng.a['Ypre']),
                        hip['lib'].ptr(gP), eng.B, eng.Hc * eng.Wc)
        g_inj = M.to_nchw(eng.backward(gP)).cpu()
        print(f'{name}: {int(flips.sum())} differing gates {per_layer}; input gradient rel L2 plain {rel_l2(x.grad, g_ref):.2e}, with the '
              f'reference\'s gates {rel_l2(g_inj, g_ref):.2e}')
        assert rel_l2(g_inj, g_ref) < 1e-5
        assert int(flips.sum()) > 0 or rel_l2(x.grad, g_ref) < 1e-4
        return
    # pcnet_256 feeds white noise: ~1e-5 px coordinate rounding -> ~1e-5 forward error -> a few ReLU gates flip
    assert rel_l2(x.grad, g_ref) < (1e-2 if name == 'pcnet_256' else 1e-4)
    assert outlier_fraction(x.grad, g_ref, 1e-3) < (5e-2 if name == 'pcnet_256' else 2e-3)  # ReLU-gate flips only
    if name == 'pcnet_256':
        # the same network with a SMOOTH projector image (the regime of the attack loop): tight gradient parity
        xs = syn.scenes(5, 1, cam_sz)
        sc = torch.from_numpy(z['s'])
        xc = xs.clone().requires_grad_(True)
        yc = so.pcnet_forward(sd, xc, sc)
        r = torch.from_numpy(z['r'])
        (yc * r).sum().backward()
        xg = xs.clone().to(DEV).requires_grad_(True)
        yg = pc(xg, sc.to(DEV))
        (yg * r.to(DEV)).sum().backward()
        assert rel_inf(yg, yc) < 1e-5
        # every activation agrees to rounding; the only disagreements in the ReLU gates are units whose value is
        # within rounding of zero (those flips are what bounds gradient parity in exact fp32)
        xw_c = (so.warp(sd, xs, cam_sz) * sd['mask'])
        _, acts = so.shading_net(sd, xw_c, (sc, xw_c * sc), return_all=True)
        eng = pc.engine(1, xs.shape[-2:])
        flips = 0
        for k_, v_ in dict(x1='X1', x2='X2', x3='X3', x4='X4', x5='X5', x6='X6', x7='X7', res1_s='S1', res2_s='S2',
                           res3_s='S3', res4_s='S4').items():
            ours = nchw(eng.a[v_].cpu())
            ref = acts[k_]
            assert rel_inf(ours, ref) < 2e-4, v_  # border pixels: zero padding x 1.5e-5 px coordinate rounding
            mism = (ours > 0) != (ref > 0)
            flips += int(mism.sum())
            if mism.any():
                assert torch.maximum(ours.abs(), ref.abs())[mism].max() < 2e-4 * ref.abs().max(), v_
        print(f'pcnet_256 smooth input: {flips} ReLU gates differ (all within rounding of 0); grad rel L2 '
              f'{rel_l2(xg.grad, xc.grad):.2e}, rel Linf {rel_inf(xg.grad, xc.grad):.2e}')
        assert flips < 50
        assert rel_l2(xg.grad, xc.grad) < 2e-3 and outlier_fraction(xg.grad, xc.grad, 1e-3) < 5e-3


@pytest.mark.parametrize('b,c,h,w', [(2, 8, 13, 9), (3, 64, 28, 28), (1, 16, 7, 12)])
def test_maxpool3s2_adjoint_blocks(hip, b, c, h, w):
    """spaa_maxpool3s2_fwd / _bwd (ResNet's stem pool, classifier.py:59-60 of the reference: torchvision maxpool after ReLU): the
    adjoint's thread-per-2x2-block form against autograd of max_pool2d(relu(x)) -- odd sizes (partial blocks, windows past the
    border), with and without the fused ReLU gate."""
    lib = hip['lib']
    torch.manual_seed(h * w)
    x = torch.randn(b, c, h, w)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    g = torch.randn(b, c, ho, wo)
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(F.relu(xr), 3, 2, 1).backward(g)
    xp = F.relu(x).clone().requires_grad_(True)
    F.max_pool2d(xp, 3, 2, 1).backward(g)
    xin = nhwc(F.relu(x), c).to(DEV)
    out = torch.zeros(b, ho, wo, c, device=DEV)
    arg = torch.zeros(b, ho, wo, c, dtype=torch.uint8, device=DEV)
    lib.call('spaa_maxpool3s2_fwd', lib.ptr(xin), lib.ptr(out), lib.ptr(arg), b, h, w, c, ho, wo)
    assert torch.equal(nchw(out.cpu(), c), F.max_pool2d(F.relu(x), 3, 2, 1))
    for gate, ref in ((1, xr.grad), (0, xp.grad)):
        gin = torch.full((b, h, w, c), float('nan'), device=DEV)
        lib.call('spaa_maxpool3s2_bwd', lib.ptr(nhwc(g, c).to(DEV)), lib.ptr(arg), gate, lib.ptr(gin), b, h, w, c, ho, wo)
        got = nchw(gin.cpu(), c)
        if gate:
            assert torch.allclose(got, ref, atol=1e-6), gate
        else:   # (ungated: windows of zeros send their gradient to the first element, as ATen does)
            assert torch.allclose(got, ref, atol=1e-6), gate


@pytest.mark.parametrize('dt', [torch.float32, torch.float16])
@pytest.mark.parametrize('b,c,h,w', [(2, 8, 12, 10), (3, 64, 28, 28), (1, 24, 6, 14)])
def test_maxpool2x2_window_kernels(hip, b, c, h, w, dt):
    """spaa_maxpool_fwd / _bwd (and the fp16-storage entries) with kernel 2 / stride 2 / padding 0 on even sides -- VGG-16's pools
    (classifier.py:21-24 of the reference), a thread per window -- against max_pool2d(relu(x)) and its autograd: pooled values,
    arg-max codes (first maximum in row-major order; bit 7 = maximum > 0), gated and ungated gradients; a concatenation window as
    output (channel stride / offset) and ties (whole windows of zeros after the ReLU)."""
    lib = hip['lib']
    torch.manual_seed(h * w + c)
    x = torch.randn(b, c, h, w)
    x = x.to(dt).float()
    ho, wo = h // 2, w // 2
    g = torch.randn(b, c, ho, wo).to(dt).float()
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(F.relu(xr), 2, 2).backward(g)
    xp = F.relu(x).clone().requires_grad_(True)
    pooled, idx = F.max_pool2d(xp, 2, 2, return_indices=True)
    pooled.backward(g)
    f16 = dt == torch.float16
    sfx = '_f16' if f16 else ''
    hp = lib.hptr
    xin = nhwc(F.relu(x), c).to(DEV).to(dt)
    cs, coff = c + 16, 8
    out = torch.zeros(b, ho, wo, cs, device=DEV, dtype=dt)
    arg = torch.zeros(b, ho, wo, c, dtype=torch.uint8, device=DEV)
    lib.call('spaa_maxpool_fwd' + sfx, hp(xin), hp(out), lib.ptr(arg), b, h, w, c, ho, wo, 2, 2, 0, cs, coff)
    assert torch.equal(nchw(out[..., coff:coff + c].float().cpu(), c), pooled.detach())
    assert float(out[..., :coff].abs().max()) == 0 and float(out[..., coff + c:].abs().max()) == 0
    code = ((idx // w) - 2 * torch.arange(ho).view(1, 1, ho, 1)) * 2 + ((idx % w) - 2 * torch.arange(wo).view(1, 1, 1, wo))
    want = (code + 128 * (pooled.detach() > 0)).permute(0, 2, 3, 1).to(torch.uint8)
    assert torch.equal(arg.cpu(), want)
    gbuf = torch.zeros(b, ho, wo, cs, device=DEV, dtype=dt)
    gbuf[..., coff:coff + c] = nhwc(g, c).to(DEV).to(dt)
    for gate, ref in ((1, xr.grad), (0, xp.grad)):
        gin = torch.full((b, h, w, c), float('nan'), device=DEV, dtype=dt)
        lib.call('spaa_maxpool_bwd' + sfx, hp(gbuf), lib.ptr(arg), gate, hp(gin), b, h, w, c, ho, wo, 2, 2, 0, cs, coff)
        assert torch.equal(nchw(gin.float().cpu(), c), ref), gate


@pytest.mark.parametrize('b,h,w', [(2, 56, 56), (3, 29, 45), (1, 112, 112)])
def test_pool_adjoint_as_stem_dgrad_prologue(hip, b, h, w):
    """ResNet-18's max-pool adjoint as the prologue of the stem's input gradient (csrc/tapconv_thinmf.hip POOL, classifier.py
    FUSE_POOL_ADJOINT; reference: torchvision maxpool + conv1 behind classifier.py:26-28,59-60): the fused launch against
    spaa_maxpool3s2_bwd followed by the plain launch -- same windows in the same order of additions, the same matrix-core products:
    BITWISE equal -- and against autograd of conv(7x7/s2) -> relu -> max_pool2d(3, 2, 1).  Odd sizes: partial windows, ragged tiles."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(h + w)
    wt = torch.randn(64, 3, 7, 7) / 147 ** 0.5
    x = torch.randn(b, 3, 2 * h, 2 * w, requires_grad=True)
    c1 = F.relu(F.conv2d(x, wt, None, 2, 3))
    assert c1.shape[2:] == (h, w)
    mp = F.max_pool2d(c1, 3, 2, 1)
    g = torch.randn_like(mp)
    mp.backward(g)
    hp_, wp_ = mp.shape[2:]
    c1n = nhwc(c1.detach(), 64).to(DEV)
    out = torch.zeros(b, hp_, wp_, 64, device=DEV)
    arg = torch.zeros(b, hp_, wp_, 64, dtype=torch.uint8, device=DEV)
    lib.call('spaa_maxpool3s2_fwd', lib.ptr(c1n), lib.ptr(out), lib.ptr(arg), b, h, w, 64, hp_, wp_)
    gd = nhwc(g, 64).to(DEV)
    dplan = cp.conv_dgrad_plan(wt, 2, 3, DEV, 'stem_dgrad_test')
    g_c1 = torch.zeros(b, h, w, 64, device=DEV)
    lib.call('spaa_maxpool3s2_bwd', lib.ptr(gd), lib.ptr(arg), 1, lib.ptr(g_c1), b, h, w, 64, hp_, wp_)
    gin0 = torch.zeros(b, 2 * h, 2 * w, 4, device=DEV)
    dplan.run(g_c1, gin0)
    assert dplan.last_tile == 72
    gin1 = torch.full((b, 2 * h, 2 * w, 4), float('nan'), device=DEV)
    dplan.run(gd, gin1, pool_adjoint=(arg, (h, w), True))
    assert dplan.last_tile == 72
    assert torch.equal(gin1, gin0)
    assert rel_inf(nchw(gin1.cpu(), 3), x.grad) < 1e-5


def test_resnet18_classifier_vs_oracle(hip):
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    for (h, crop, insz, b) in [(64, (60, 60), (56, 56), 3), (256, (240, 240), (224, 224), 2)]:
        torch.manual_seed(1)
        im = torch.rand(b, 3, h, h, requires_grad=True)
        raw, p, idx = so.OracleClassifier('resnet18', csd, input_sz=insz)(im, crop)
        r = torch.randn(b, 1000)
        (raw * r).sum().backward()
        clf = hip['clf'].Classifier('resnet18', DEV, state_dict=csd, input_sz=insz)
        im2 = im.detach().clone().to(DEV).requires_grad_(True)
        raw2, p2, idx2 = clf(im2, crop)
        (raw2 * r.to(DEV)).sum().backward()
        assert rel_inf(raw2, raw) < 1e-5
        # input gradient: ~1e-6 when every ReLU gate agrees.  A unit whose pre-activation is within rounding of zero may
        # fall on the other side here (other summation order, e.g. split-K) than in the oracle: one such unit in layer1
        # changes the ~1.1e3 gradient elements of its receptive field (measured: the fp32 and fp64 ORACLES disagree with
        # each other in exactly this way at 64x64, rel L2 1.0e-3).  Allow a few flips, nothing else.
        gl2, gout = rel_l2(im2.grad, im.grad), outlier_fraction(im2.grad, im.grad, 1e-3)
        print(f'resnet18 {h}x{h}: input-gradient rel L2 {gl2:.2e}, outliers {gout:.2e}')
        assert gl2 < 1e-4 or (gl2 < 5e-3 and gout < 5e-3)
        assert (idx2[:, 0] == idx[:, 0]).all() and np.allclose(p2[:, 0], p[:, 0], atol=1e-5)
        assert p2.shape == (b, 1000) and idx2.shape == (b, 1000)


# ---------------------------------------------------------------------------------------------------------------
def _setup_case(hip, z):
    sz = tuple(int(v) for v in z['sz'])
    sd = syn.pcnet_state_dict(int(z['seed']), cam_sz=sz, mask=str(z['mask']))
    csd = syn.resnet18_state_dict(2, logit_gain=float(z['gain']))
    insz = tuple(int(v) for v in z['input_sz'])
    scene = syn.scenes(int(z['scene_seed']), 1, sz)
    setup = dict(classifier_crop_sz=tuple(int(v) for v in z['crop']), prj_brightness=0.5, prj_im_sz=sz)
    pc = make_pcnet(hip, sd, sz)
    clf = hip['clf'].Classifier('resnet18', DEV, state_dict=csd, input_sz=insz)
    oclf = so.OracleClassifier('resnet18', csd, input_sz=insz)
    return sd, pc, clf, oclf, scene, setup


def _oracle_activations(sd, csd, x_prev, scene_b, cam_sz, crop, insz, body='resnet18'):
    """The oracle's forward for one teacher-forced iteration, with every gate-carrying activation (tests/gates.py)."""
    import gates
    with torch.no_grad():
        xw = so.warp(sd, x_prev.clamp(0, 1), cam_sz) * sd['mask']
        y, acts = so.shading_net(sd, xw, (scene_b, xw * scene_b), return_all=True)
        pre = so.classifier_preprocess(y, crop, insz)
        if body == 'inception_v3':   # (no per-layer dump in the oracle: every ReLU output, in call order)
            with gates.record_relu() as cacts:
                so.inception_v3_forward(csd, pre)
        else:
            _, cacts = dict(resnet18=so.resnet18_forward, vgg16=so.vgg16_forward)[body](csd, pre, return_all=True)
    return acts, cacts


# (the 256 x 256 cases -- the benchmarked size and tile selection -- run 16 iterations: the oracle's first success is iteration 9, so
# iterations 9..15 take the COLOUR step: the per-sample cotangent choice in spaa_shading_head_bwd_select and the best-so-far copies)
TEACHER_FORCED_ITERS = {'spaa_256_untargeted': 16, 'spaa_256_near': 16}


@pytest.mark.parametrize('name', ['spaa_64_near', 'spaa_64_prjl2', 'spaa_64_caml2_dthr', 'spaa_64_camdE', 'spaa_256_untargeted',
                                  'spaa_256_near', 'spaa_256_untargeted+oracle_grid', 'spaa_256_near+oracle_grid'])
def test_spaa_teacher_forced_iterations(hip, golden_dir, name):
    """One HIP iteration from the oracle's state at iteration k must reproduce the oracle's iteration k: losses, masks,
    top-1, and the updated projector image to 1e-4 relative L-inf (BASELINE.json's bar) — on EVERY sample whose ReLU /
    clamp / max-pool gates agree with the oracle's, and on ALL samples once the oracle's gates are used in the HIP
    backward (tests/gates.py): every excess over 1e-4 is a unit within rounding of zero falling on the other side.
    The best-so-far bookkeeping of the same iteration (projector_based_attack.py:318-328) is checked as well: `best`, and the copies
    into prj_adv_best (the POST-step image, Q4) / cam_infer_best (the pre-step inference) for exactly the successful samples."""
    import gates
    name, _, variant = name.partition('+')
    z = load(golden_dir, name)
    sd, pc, clf, oclf, scene, setup = _setup_case(hip, z)
    targets, targeted = [int(t) for t in z['targets']], bool(z['targeted'])
    d_thr, stealth = float(z['d_thr']), str(z['stealth'])
    csd = syn.resnet18_state_dict(2, logit_gain=float(z['gain']))
    crop, insz, cam_sz = setup['classifier_crop_sz'], tuple(int(v) for v in z['input_sz']), setup['prj_im_sz']
    if variant == 'oracle_grid':
        # the HIP path samples through the ORACLE's fine grid (WarpingNet.fine_grid, the reference's own cache attribute, models.py:176):
        # what is left is the arithmetic of the loop's kernels -- held to the 64 x 64 bars
        pc.warping_net.fine_grid = so.warping_fine_grid(sd, (1, 3) + tuple(cam_sz), tuple(cam_sz)).to(DEV)
    tr = []
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    so.spaa(sd, oclf, targets, targeted, scene, d_thr, stealth, setup, iters=TEACHER_FORCED_ITERS.get(name, 14), trace=tr)
    assert (np.stack([t['top1'] for t in tr])[:3] == z['top1'][:3]).all()  # oracle here == reference golden
    A, M = hip['attack'], hip['models']
    B = len(targets)
    st = A.AttackState(pc, clf, targets, scene, stealth, setup, DEV)
    scene_b = scene.expand(B, -1, -1, -1)
    worst_clean = worst_forced = worst_plain = 0.0
    n_flip_samples = n_flips = n_clean = 0
    n_col_steps = n_best = n_tracked = 0
    # Intermediate activations at 256 x 256 with the engine's OWN fine grid: the two fp32 constructions of the sampling grid (affine o TPS
    # o refine, ~10 roundings each) differ by ~1e-6 in normalised coordinates = 1e-4 pixels; on the high-contrast projector images of the
    # later iterations that moves the warped image by 2e-5 of its scale and the deepest activations by 2e-4 (measured, round 6: 2.2e-4 /
    # 2.0e-4; with the oracle's grid injected -- the `+oracle_grid` variants -- 1.7e-6 / 3.0e-6, image 1.1e-6).  The bars for the
    # activations and for "a flipped unit sits within the value error of zero" are 3 x those measurements here; the bar on the RESULT --
    # the projector image to 1e-4 -- is the same everywhere.
    own_256 = setup['prj_im_sz'][1] > 64 and variant != 'oracle_grid'
    tol_scale = 50.0 if own_256 else 1.0       # (VALUE_TOL 1.3e-5 -> 6.5e-4, NEAR_ZERO 1.3e-6 -> 6.5e-5: measured 2.2e-4 / 3.3e-5)
    if os.environ.get('SPAA_TF_TOL_SCALE'):      # (measurement runs: how large are the intermediate differences)
        tol_scale = float(os.environ['SPAA_TF_TOL_SCALE'])
    measured = dict(value=0.0, near=0.0, tie=0.0) if setup['prj_im_sz'][1] > 64 else None
    layers = {}
    for k in range(len(tr)):
        t = tr[k]
        x_prev = torch.full((B, 3, *setup['prj_im_sz']), 0.5) if k == 0 else torch.from_numpy(tr[k - 1]['prj_adv'])
        ref = torch.from_numpy(t['prj_adv'])
        acts, cacts = _oracle_activations(sd, csd, x_prev, scene_b, cam_sz, crop, insz)
        results = {}
        for mode in ('plain', 'oracle_gates'):
            st.x.copy_(M.to_nhwc4(x_prev.to(DEV)))
            st.stats[:, 5] = torch.from_numpy(t['col_loss_best_before']).to(DEV)
            st.x_best.fill_(-7.0)       # sentinels: the iteration's tracking copies must touch exactly the successful samples
            st.cam_best.fill_(-7.0)
            st.forward_decide(targeted, d_thr, 0.9)
            pairs = gates.pcnet_pairs(st.eng, acts) + gates.resnet18_pairs(st.clf.body, cacts)
            if mode == 'plain':
                flips, per_layer = gates.count_flips(pairs, near_zero=gates.NEAR_ZERO * tol_scale, value_tol=gates.VALUE_TOL * tol_scale,
                                                     measured=measured)
                for kk, vv in per_layer.items():
                    layers[kk] = layers.get(kk, 0) + vv
                stt, sts = st.state.cpu().numpy(), st.stats.cpu().numpy()
                assert np.allclose(sts[:, 1], t['caml2'], rtol=1e-4), (k, 'caml2')
                assert np.allclose(sts[:, 2], t['camdE'], rtol=1e-4), (k, 'camdE')
                assert np.allclose(sts[:, 0], t['p1'], atol=2e-4), (k, 'p1')
                assert np.allclose(sts[:, 6], t['target_logit'], rtol=1e-4, atol=1e-4), (k, 'logit')
                assert rel_inf(M.to_nchw(st.eng.a['Y']), torch.from_numpy(t['cam_infer'])) < 1e-4, (k, 'cam_infer')
                # decisions: equal unless the oracle itself sits on a knife edge
                edge = (np.abs(t['p1'] - 0.9) < 1e-3) | (np.abs(t['caml2'] * 255 - d_thr) < 1e-2)
                assert ((stt[:, 3] == t['top1']) | edge).all(), (k, 'top1')
                assert ((stt[:, 0] == t['succ']) | edge).all() and ((stt[:, 1] == t['best_adv']) | edge).all(), (k, 'masks')
                # best = best_adv and col_loss < the best so far (:318-320): equal unless the two losses are within rounding of each other
                edge_best = edge | (np.abs(t['col_loss'] - t['col_loss_best_before']) < 1e-4 * np.abs(t['col_loss']))
                assert ((stt[:, 2] == t['best']) | edge_best).all(), (k, 'best')
                same = torch.from_numpy(stt[:, 1] == t['best_adv'])
                n_col_steps += int((same & torch.from_numpy(t['best_adv'].astype(bool))).sum())
                n_best += int(((stt[:, 2] == t['best']) & t['best'].astype(bool)).sum())
            else:
                gates.inject(pairs, (st.eng, st.clf.body))
            st.backward_step(2, 1)
            xn = M.to_nchw(st.x).cpu()
            results[mode] = torch.tensor([rel_inf(xn[b], ref[b]) for b in range(B)])
            if mode == 'plain':
                # :323-328: prj_adv_best takes the image AFTER this iteration's step (Q4), cam_infer_best the inference BEFORE it, for the
                # successful samples only (best implies succ); everybody else keeps what they had (here: the sentinel)
                xb, cb, yy = M.to_nchw(st.x_best).cpu(), M.to_nchw(st.cam_best).cpu(), M.to_nchw(st.eng.a['Y']).cpu()
                for b in range(B):
                    if stt[b, 0]:
                        assert torch.equal(xb[b], xn[b]) and torch.equal(cb[b], yy[b]), (k, b, 'best copies')
                        n_tracked += 1
                    else:
                        assert (xb[b] == -7.0).all() and (cb[b] == -7.0).all(), (k, b, 'best copies touched')
        clean = same & (flips == 0)
        flipped = same & (flips > 0)
        n_clean += int(clean.sum())
        n_flip_samples += int(flipped.sum())
        n_flips += int(flips[same].sum())
        if clean.any():
            worst_clean = max(worst_clean, float(results['plain'][clean].max()))
        if same.any():
            worst_forced = max(worst_forced, float(results['oracle_gates'][same].max()))
            worst_plain = max(worst_plain, float(results['plain'][same].max()))
        assert (results['plain'][clean] < 1e-4).all(), (k, results['plain'])
        assert (results['oracle_gates'][same] < 1e-4).all(), (k, results['oracle_gates'])
    print(f'{name}: {len(tr)} teacher-forced iterations x {B} samples: {n_clean} sample-iterations with identical gates: '
          f'image rel Linf max {worst_clean:.2e}; {n_flip_samples} with {n_flips} differing gates {layers} (all within rounding '
          f'of zero): plain max {worst_plain:.2e}, with the oracle\'s gates max {worst_forced:.2e}; {n_col_steps} sample-iterations took the '
          f'colour step, {n_best} set a new best, {n_tracked} best-so-far copies checked')
    assert n_clean > 0
    if measured is not None:
        print(f'    measured at this size: activations {measured["value"]:.2e} (bar {gates.VALUE_TOL * tol_scale:.1e}), flipped units within '
              f'{measured["near"]:.2e} of zero (bar {gates.NEAR_ZERO * tol_scale:.1e}), arg-max ties {measured["tie"]:.2e}')
    if name in TEACHER_FORCED_ITERS:   # the colour-step branch and the best tracking must have been exercised at the benchmarked size
        assert n_col_steps >= 4 and n_best >= 2 and n_tracked >= 4, (n_col_steps, n_best, n_tracked)


def test_spaa_exact_cases_and_quirks(hip, golden_dir):
    """Q7: when no sample ever succeeds the output is exactly the gray image and the scene; first iteration of every
    golden run matches the reference to rounding."""
    A = hip['attack']
    z = load(golden_dir, 'spaa_64_imagenet10')
    sd, pc, clf, oclf, scene, setup = _setup_case(hip, z)
    tr = []
    cam, prj = A.spaa(pc, clf, None, [int(t) for t in z['targets']], True, scene, float(z['d_thr']), str(z['stealth']),
                      DEV, setup, trace=tr)
    assert torch.equal(prj.cpu(), torch.from_numpy(z['prj_adv_best'])) and (prj == 0.5).all()
    assert torch.equal(cam.cpu(), torch.from_numpy(z['cam_infer_best']))
    st = torch.stack([t[0] for t in tr]).cpu().numpy()
    assert not st[:, :, 0].any() and (st[:3, :, 3] == z['top1'][:3]).all()
    for name in ('spaa_64_untargeted', 'spaa_64_near', 'spaa_256_untargeted'):
        z = load(golden_dir, name)
        sz = tuple(int(v) for v in z['sz'])
        st1 = _first_iteration_gate_aware(hip, 'resnet18', syn.resnet18_state_dict(2, logit_gain=float(z['gain'])),
                                          tuple(int(v) for v in z['input_sz']), sz, tuple(int(v) for v in z['crop']),
                                          [int(t) for t in z['targets']], int(z['seed']), mask=str(z['mask']),
                                          targeted=bool(z['targeted']), scene_seed=int(z['scene_seed']),
                                          d_thr=float(z['d_thr']), stealth=str(z['stealth']), golden_it0=z['prj_adv_it0'])
        assert (st1.state[:, 3].cpu().numpy() == z['top1'][0]).all()
    # B < 8 targeted works (the reference raises IndexError: Q10), and a [B,3,H,W] scene batch is accepted (Q9)
    z = load(golden_dir, 'spaa_64_near')
    sd, pc, clf, oclf, scene, setup = _setup_case(hip, z)
    cam, prj = A.spaa(pc, clf, None, [204, 291, 129], True, scene.expand(3, -1, -1, -1), 5, 'camdE_caml2', DEV, setup,
                      iters=3)
    assert cam.shape == (3, 3, 64, 64) and prj.min() >= 0 and prj.max() <= 1


def _oracle_fp64(sd, csd, insz, targets, targeted, scene, d_thr, stealth, setup, iters):
    """The oracle in float64: the common yardstick for 'how far may an fp32 implementation drift'."""
    torch.set_default_dtype(torch.float64)
    try:
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        csd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in csd.items()}
        tr = []
        so.spaa(sd64, so.OracleClassifier('resnet18', csd64, input_sz=insz), targets, targeted, scene.double(), d_thr, stealth,
                setup, iters=iters, trace=tr)
    finally:
        torch.set_default_dtype(torch.float32)
    return tr


# (golden whose configuration is used, iterations, scene seed -- None: the golden's own scene.  The three 64 x 64 goldens share
# their scene, hence their first iterations: other scenes make the four trajectories independent samples)
DRIFT_CASES = [('spaa_64_near', 12, None), ('spaa_64_prjl2', 12, 5), ('spaa_64_caml2_dthr', 12, 9), ('spaa_64_near/b', 12, 13),
               ('spaa_64_near/c', 12, 17), ('spaa_64_prjl2/b', 12, 25), ('spaa_64_caml2_dthr/b', 12, 33),
               ('spaa_64_near/d', 12, 61), ('spaa_64_prjl2/c', 12, 65), ('spaa_64_caml2_dthr/c', 12, 69), ('spaa_64_near/e', 12, 73),
               ('spaa_256_near', 8, None)]


def test_free_running_drift_vs_fp64_oracle(hip, golden_dir):
    """Free-running trajectories (no teacher forcing).  The loop amplifies rounding differences ~3x per iteration (the fp32
    oracle == reference drifts from the SAME code run in float64 by 3e-6, 1e-4, 8e-4, 3e-3, ... 2e-1 relative L-inf), so
    'identical results' for 50 iterations is not defined for any two fp32 implementations.  What is defined: the HIP path
    must not drift from the fp64 trajectory faster than the fp32 reference itself does.  WHICH of two fp32 trajectories takes
    its next gate flip first is itself rounding noise (one scenario reads 0.19 for one build and 1.61 for the next, which
    differ in the summation order of one input-gradient kernel; over independent scenes of one build it ranges from 0.04 to 3.8),
    so the statement is a STATISTIC over seven 64 x 64 scenarios (the goldens' configurations on independent scenes; a 256 x 256
    one joins when a scene with a clean first iteration exists): the median of the per-scenario geometric-mean ratios (HIP
    drift / fp32-oracle drift) is at most 1.5, none above 30, and the HIP path runs 10x ahead of the oracle's drift in at most
    two scenarios more than the oracle runs ahead of the HIP path's."""
    A, M = hip['attack'], hip['models']
    rows, ratios, left_out = [], [], []
    ahead_hip = ahead_f32 = 0
    for name, iters, scene_seed in DRIFT_CASES:
        z = load(golden_dir, name.split('/')[0])
        sd, pc, clf, oclf, scene, setup = _setup_case(hip, z)
        csd = syn.resnet18_state_dict(2, logit_gain=float(z['gain']))
        insz = tuple(int(v) for v in z['input_sz'])
        targets, d_thr, stealth = [int(t) for t in z['targets']], float(z['d_thr']), str(z['stealth'])
        golden_scene = scene
        if True:
            # admission: a scene on which one of the two fp32 implementations ALREADY flips a gate in the first iteration (a unit
            # within rounding of zero) says nothing about drift; take the first of a few candidate scenes with a clean start
            # (the golden's own scene first where the case names one)
            first = int(z['scene_seed']) if scene_seed is None else scene_seed
            for cand in ((first,) if int(z['sz'][0]) > 64 else (first, first + 20, first + 21, first + 22)):   # (256 x 256: the golden's own scene only -- scenes 21, 22, 23 were tried in round 4: none clean, profiles/r04_parity.txt)
                scene = syn.scenes(cand, 1, tuple(int(v) for v in z['sz']))
                r64 = _oracle_fp64(sd, csd, insz, targets, True, scene, d_thr, stealth, setup, 1)[0]['prj_adv']
                t32 = []
                so.spaa(sd, oclf, targets, True, scene, d_thr, stealth, setup, iters=1, trace=t32)
                st0 = A.AttackState(pc, clf, targets, scene, stealth, setup, DEV)
                st0.iteration(True, d_thr, 2, 1, 0.9)
                e_h = rel_inf(M.to_nchw(st0.x).double(), torch.from_numpy(r64))
                e_o = rel_inf(torch.from_numpy(t32[0]['prj_adv']).double(), torch.from_numpy(r64))
                del st0
                if e_h < 1e-4 and e_o < 1e-4:
                    break
                print(f'  {name}: scene {cand} not admitted (first iteration vs fp64: HIP {e_h:.1e}, fp32 oracle {e_o:.1e})')
            else:
                # (at 256 x 256 a batch has ~1e8 gated units: most scenes put one of them within rounding of zero for one or the
                # other implementation -- scene 22: the fp32 ORACLE flips and the HIP path does not; the flip counts of both
                # sides over twelve scenes: test_gate_flip_statistic_256_against_fp64.)  Not a silent skip: the golden scene's
                # first iteration must then pass the GATE-AWARE bar -- 1e-4 on every sample whose gates agree with the oracle's
                # and on every sample with the oracle's gates, each disagreeing unit within rounding of zero on both sides --
                # and only full-size scenarios may be left out.
                assert int(z['sz'][0]) > 64, f'{name}: no 64 x 64 candidate scene with a clean first iteration'
                stg = _first_iteration_gate_aware(hip, 'resnet18', csd, insz, tuple(int(v) for v in z['sz']), tuple(int(v) for v in z['crop']),
                                                  targets, int(z['seed']), mask=str(z['mask']), targeted=True, scene_seed=first,
                                                  d_thr=d_thr, stealth=stealth)
                assert int(stg.flips.sum()) >= 1, f'{name}: first iteration off by more than 1e-4 without a single differing gate'
                left_out.append((name, int(stg.flips.sum())))
                print(f'  {name}: no clean first iteration on the golden scene; gate-aware check passed '
                      f'({int(stg.flips.sum())} gates within rounding of zero differ): left out of the drift statistic')
                del stg
                continue
        tr64 = _oracle_fp64(sd, csd, insz, targets, True, scene, d_thr, stealth, setup, iters)
        tr32 = []
        so.spaa(sd, oclf, targets, True, scene, d_thr, stealth, setup, iters=iters, trace=tr32)
        st = A.AttackState(pc, clf, targets, scene, stealth, setup, DEV)
        d_hip, d_f32 = [], []
        for k in range(iters):
            st.iteration(True, d_thr, 2, 1, 0.9)
            ref = torch.from_numpy(tr64[k]['prj_adv'])
            d_hip.append(rel_inf(M.to_nchw(st.x).double(), ref))
            d_f32.append(rel_inf(torch.from_numpy(tr32[k]['prj_adv']).double(), ref))
        assert d_hip[0] < 1e-4 and d_f32[0] < 1e-4                      # first iteration: BASELINE.json's bar, both
        # drift is multiplicative (chaotic amplification): compare in the log domain, geometric mean over the iterations
        ratio = float(np.exp(np.mean(np.log(np.array(d_hip) / np.array(d_f32)))))
        ratios.append(ratio)
        rows.append((name, d_hip, d_f32, ratio))
        # the drift grows in jumps (a ReLU gate or a per-image decision of the loop flips: one jump), and WHICH of the two fp32
        # trajectories takes its next jump first is rounding noise: count, for either side, the scenarios in which it runs more than
        # 10x ahead of the other's drift up to two iterations later (round 4, conv1 pair fused: HIP ahead in spaa_64_near --
        # 2.3e-2 at iteration 4 where the fp32 oracle reads 2.9e-4 -- and the oracle ahead in spaa_64_prjl2, ratio 0.04)
        ahead_hip += any(h >= 10 * max(max(d_f32[:i + 3]), 1e-6) for i, h in enumerate(d_hip))
        ahead_f32 += any(o >= 10 * max(max(d_hip[:i + 3]), 1e-6) for i, o in enumerate(d_f32))
        if name == 'spaa_64_near':
            # 50-iteration statistics are preserved: camera-side distortion of the best images within 5 % of the reference golden
            cam, prj = A.spaa(pc, clf, None, targets, True, golden_scene, d_thr, stealth, DEV, setup)
            ref_cam = torch.from_numpy(z['cam_infer_best'])
            d_ours = torch.norm(cam.cpu() - golden_scene, dim=1).mean().item()
            d_ref = torch.norm(ref_cam - golden_scene, dim=1).mean().item()
            assert abs(d_ours - d_ref) / d_ref < 0.05
        del st
    print('free-running drift from the fp64 oracle, relative Linf of the projector image per iteration:')
    for name, d_hip, d_f32, ratio in rows:
        print(f'  {name}: geometric-mean ratio HIP / fp32 oracle = {ratio:.2f}')
        print('     HIP          ', ' '.join(f'{v:.1e}' for v in d_hip))
        print('     fp32 oracle  ', ' '.join(f'{v:.1e}' for v in d_f32))
    assert len(ratios) >= 9 and len(left_out) <= 1
    med = float(np.median(ratios))
    srt = sorted(ratios)
    gm = float(np.exp(np.mean(np.log(ratios))))
    print(f'  median of the {len(ratios)} ratios = {med:.2f} (bound 1.5), geometric mean = {gm:.2f} (bound 2), min = {srt[0]:.2f}, '
          f'second largest = {srt[-2]:.2f} (bound 10), max = {srt[-1]:.2f} (bound 30)')
    print(f'  scenarios with one side more than 10x ahead of the other (two iterations of slack): HIP {ahead_hip}, fp32 oracle {ahead_f32}')
    # spread: measured over rounds 3-5 the per-scenario ratio ranges over 0.04 ... 7.8 (log-symmetric: one early gate flip of either
    # side, held for the rest of the twelve iterations).  A fidelity loss moves the whole distribution: median and geometric mean;
    # a single bad scenario shows in the second largest / largest ratio.
    assert med <= 1.5 and gm <= 2.0 and srt[-2] < 10.0 and srt[-1] < 30.0 and ahead_hip <= ahead_f32 + 2


def _three_way_layers(pairs64, pairs32):
    """Per gate-carrying layer: forward error against the fp64 oracle of the HIP engine and of the fp32 oracle (relative to the
    layer's largest fp64 activation), and the per-sample number of gates (ReLU / clamp signs, max-pool arg-maxes) on which each
    differs from fp64.  `pairs64` / `pairs32`: tests/gates.py pair lists built from the fp64 resp. fp32 oracle activations."""
    rows, fl_h, fl_o = [], None, None
    for (name, kind, hbuf, o64), (_n, _k, _h, o32) in zip(pairs64, pairs32):
        if kind == 'argmax':
            h, c64, c32 = hbuf.detach().cpu(), o64.codes, o32.codes
            mh, mo, eh, eo = (h != c64), (c32 != c64), 0.0, 0.0
        else:
            h = hbuf.detach().cpu().double()
            a64, a32 = o64.double(), o32.double()
            scale = float(a64.abs().max()) + 1e-300
            eh, eo = float((h - a64).abs().max()) / scale, float((a32 - a64).abs().max()) / scale
            if kind == 'relu':
                mh, mo = (h > 0) != (a64 > 0), (a32 > 0) != (a64 > 0)
            else:
                g = lambda v: (v > 0) & (v <= 1)   # noqa: E731
                mh, mo = g(h) != g(a64), g(a32) != g(a64)
        nh, no = mh.flatten(1).sum(1), mo.flatten(1).sum(1)
        fl_h = nh if fl_h is None else fl_h + nh
        fl_o = no if fl_o is None else fl_o + no
        rows.append((name, eh, eo, int(nh.sum()), int(no.sum())))
    return rows, fl_h, fl_o


def test_gate_flip_statistic_256_against_fp64(hip):
    """Rounding parity at the benchmarked size, with float64 as the referee.  Twelve independent 256 x 256 scenes, B = 2, first
    iteration of the benchmarked configuration (PCNet seed 0, ResNet-18, camdE_caml2): the oracle run in float64, the fp32
    oracle (== the reference) and the HIP path.  With ~3e6 gated units per sample some sit within rounding of zero, and an
    fp32 implementation puts a few on the other side -- the reference's own fp32 path included.  Asserted:
      * forward: per gate-carrying layer, the HIP engine's error against float64 is at most 3 x the fp32 oracle's (+ 2e-7 of the
        layer scale) and at most 1.5e-6 of the layer scale.  Measured (round 5, profiles/r05_parity.txt): 25 of the 30 layers within
        1.4 x of the oracle's error (most below it); the layers behind the Winograd F(2x2,3x3) kernels (x3, x4, x5, res4_s) carry
        1.0-1.1e-6 where the oracle's direct fp32 sums carry 4-6e-7 (ratio <= 2.5: the transform's own rounding, not lost operand
        bits);
      * gates: the number of scenes whose produced projector image is more than 1e-4 from float64 (= a flipped gate reached the
        gradient) is at most the fp32 oracle's + 2 for the HIP path, and the total number of differing gates at most 2 x + 8;
      * every HIP image without a differing gate is within 1e-4 of float64."""
    import gates
    A, M = hip['attack'], hip['models']
    sz, crop, insz = (256, 256), (240, 240), (224, 224)
    sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='ones')
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    pc = make_pcnet(hip, sd, sz)
    clf = hip['clf'].Classifier('resnet18', DEV, state_dict=csd, input_sz=insz)
    oclf = so.OracleClassifier('resnet18', csd, input_sz=insz)
    setup = dict(classifier_crop_sz=crop, prj_brightness=0.5, prj_im_sz=sz)
    targets, B = [syn.IMAGENET10_TARGETS[0], syn.IMAGENET10_TARGETS[5]], 2
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    csd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in csd.items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    x0 = torch.full((B, 3, *sz), 0.5)
    bad_h = bad_o = tot_h = tot_o = 0
    worst = {}
    print('scene: image vs fp64 (HIP | fp32 oracle), gates differing from fp64 (HIP | fp32 oracle)')
    for seed in range(101, 113):
        scene = syn.scenes(seed, 1, sz)
        scene_b = scene.expand(B, -1, -1, -1)
        r64 = torch.from_numpy(_oracle_fp64(sd, csd, insz, targets, True, scene, 5, 'camdE_caml2', setup, 1)[0]['prj_adv'])
        t32 = []
        so.spaa(sd, oclf, targets, True, scene, 5, 'camdE_caml2', setup, iters=1, trace=t32)
        r32 = torch.from_numpy(t32[0]['prj_adv']).double()
        torch.set_default_dtype(torch.float64)
        try:
            acts64, cacts64 = _oracle_activations(sd64, csd64, x0.double(), scene_b.double(), sz, crop, insz)
        finally:
            torch.set_default_dtype(torch.float32)
        acts32, cacts32 = _oracle_activations(sd, csd, x0, scene_b, sz, crop, insz)
        st = A.AttackState(pc, clf, targets, scene, 'camdE_caml2', setup, DEV)
        st.forward_decide(True, 5, 0.9)
        p64 = gates.pcnet_pairs(st.eng, acts64) + gates.resnet18_pairs(st.clf.body, cacts64)
        p32 = gates.pcnet_pairs(st.eng, acts32) + gates.resnet18_pairs(st.clf.body, cacts32)
        rows, fl_h, fl_o = _three_way_layers(p64, p32)
        st.backward_step(2, 1)
        xh = M.to_nchw(st.x).cpu().double()
        e_h = [rel_inf(xh[b], r64[b]) for b in range(B)]
        e_o = [rel_inf(r32[b], r64[b]) for b in range(B)]
        for name, eh, eo, _nh, _no in rows:
            w = worst.setdefault(name, [0.0, 0.0])
            w[0], w[1] = max(w[0], eh), max(w[1], eo)
        for b in range(B):
            if int(fl_h[b]) == 0:
                assert e_h[b] < 1e-4, (seed, b, e_h[b])
        bad_h += max(e_h) > 1e-4
        bad_o += max(e_o) > 1e-4
        tot_h += int(fl_h.sum())
        tot_o += int(fl_o.sum())
        lay = {n: (nh, no) for n, _a, _b, nh, no in rows if nh or no}
        print(f'  {seed}: {max(e_h):.1e} | {max(e_o):.1e}   gates {int(fl_h.sum())} | {int(fl_o.sum())}  {lay}')
        del st
    print('forward error against fp64 per layer, relative to the layer scale, max over the scenes: HIP | fp32 oracle')
    for name, (eh, eo) in worst.items():
        print(f'  {name:24s} {eh:.2e} | {eo:.2e}')
    ratio = max(eh / (eo + 1e-30) for eh, eo in worst.values() if eh > 0)
    print(f'scenes (of 12) with an image more than 1e-4 from fp64: HIP {bad_h}, fp32 oracle {bad_o}; differing gates in all: HIP {tot_h}, '
          f'fp32 oracle {tot_o}; worst layer error ratio HIP / fp32 oracle {ratio:.2f}')
    assert bad_h <= bad_o + 2 and tot_h <= 2 * tot_o + 8
    for name, (eh, eo) in worst.items():
        assert eh <= 3.0 * eo + 2e-7 and eh <= 1.5e-6, (name, eh, eo)


FIFTY = ['spaa_64_untargeted', 'spaa_64_imagenet10', 'spaa_64_near', 'spaa_64_caml2_dthr', 'spaa_64_prjl2', 'spaa_64_camdE',
         'spaa_256_untargeted', 'spaa_256_near']


@pytest.mark.parametrize('name', FIFTY)
def test_spaa_fifty_iterations_statistics(hip, golden_dir, name):
    """All eight 50-iteration runs of the unmodified reference (fixtures spaa_*): the free-running HIP attack must reach the same
    OUTCOME.  Element-wise equality of the produced images is not defined for this loop (the reference moves its own output by
    0.17-0.24 relative L-inf under a thread-count change / a one-ulp start: fixture sensitivity_64), so what is compared is
    what the caller uses the result for: which samples were attacked successfully, and how visible the attack is on the
    camera side (mean L2 and mean dE2000 of `cam_infer_best` against the scene, projector_based_attack.py:275-283).
    Envelope: success sets equal up to one sample in eight; distortions within 2 % (measured over the eight runs: sets
    identical, distortions within 0.5 %: table in profiles/r03_parity.txt)."""
    A = hip['attack']
    z = load(golden_dir, name)
    sd, pc, clf, oclf, scene, setup = _setup_case(hip, z)
    targets, targeted = [int(t) for t in z['targets']], bool(z['targeted'])
    tr = []
    cam, prj = A.spaa(pc, clf, None, targets, targeted, scene, float(z['d_thr']), str(z['stealth']), DEV, setup, trace=tr)
    cam, prj = cam.cpu(), prj.cpu()
    st = torch.stack([t[0] for t in tr]).cpu().numpy()         # [50, B, 4]: succ, best_adv, best, top-1
    ever_hip = st[:, :, 2].any(axis=0)                          # a best image was recorded at least once
    ever_ref = z['best'].any(axis=0)
    k = z['cam_infer_best'].shape[0]                            # (spaa_256_near keeps the first two samples)
    ref_cam = torch.from_numpy(z['cam_infer_best'])
    sc = scene.expand(len(targets), -1, -1, -1)

    def dist(c):
        l2 = torch.norm(c - sc[:c.shape[0]], dim=1).mean().item()
        de = so.ciede2000_diff(so.rgb2lab_diff(c), so.rgb2lab_diff(sc[:c.shape[0]].contiguous())).mean().item()
        return l2, de

    l2h, deh = dist(cam[:k])
    l2r, der = dist(ref_cam)
    print(f'{name}: samples with a recorded best image: HIP {int(ever_hip.sum())} / reference {int(ever_ref.sum())} of {len(targets)}; '
          f'final-iteration successes HIP {int(st[-1, :, 0].sum())} / reference {int(z["succ"][-1].sum())}; cam_infer_best vs scene: '
          f'mean L2 {l2h:.5f} / {l2r:.5f} ({(l2h / max(l2r, 1e-12) - 1) * 100:+.1f} %), mean dE {deh:.4f} / {der:.4f} '
          f'({(deh / max(der, 1e-12) - 1) * 100:+.1f} %)')
    assert int((ever_hip != ever_ref).sum()) <= (1 if len(targets) >= 8 else 0)
    # (whether the LAST iteration happens to be a success alternates with the adversarial / colour steps of a sample: logged,
    # asserted only as a count over a batch)
    if len(targets) >= 8:
        assert abs(int(st[-1, :, 0].sum()) - int(z['succ'][-1].sum())) <= 2
    if l2r > 0:
        assert abs(l2h / l2r - 1) < 0.02 and abs(deh / der - 1) < 0.02
    else:   # nobody ever succeeded: the reference returns the scene itself (Q7) and so must we
        assert l2h == 0.0 and torch.equal(prj, torch.from_numpy(z['prj_adv_best']))
    assert prj.min() >= 0 and prj.max() <= 1


# ---------------------------------------------------------------------------------------------------------------
def test_full_size_properties_batch64(hip):
    """BASELINE.json config sizes (B=64, 256x256, ResNet-18): size-independent properties of the HIP path."""
    A, M = hip['attack'], hip['models']
    sz = (256, 256)
    sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='ones')
    pc = make_pcnet(hip, sd, sz)
    clf = hip['clf'].Classifier('resnet18', DEV, state_dict=syn.resnet18_state_dict(2, logit_gain=20.0))
    scenes = syn.scenes(11, 8, sz).repeat_interleave(8, dim=0)
    targets = (syn.IMAGENET10_TARGETS[:8]) * 8
    setup = dict(classifier_crop_sz=(240, 240), prj_brightness=0.5, prj_im_sz=sz)
    st = A.AttackState(pc, clf, targets, scenes, 'camdE_caml2', setup, DEV)
    st.iteration(True, 5, 2, 1, 0.9)
    y1 = st.eng.a['Y'].clone()
    x1 = st.x.clone()
    # (1) the per-sample step has exactly the prescribed length: ||x1 - gray||_2 == lr (2 adversarial, 1 colour)
    step = (x1[..., :3] - 0.5).flatten(1).norm(dim=1).cpu()
    lr = torch.where(st.state[:, 1].cpu() != 0, torch.tensor(1.0), torch.tensor(2.0))
    assert torch.allclose(step, lr, rtol=1e-4)
    # (2) samples are independent: a sub-batch of 8 reproduces the corresponding rows of the batch of 64
    st8 = A.AttackState(pc, clf, targets[8:16], scenes[8:16], 'camdE_caml2', setup, DEV)
    st8.iteration(True, 5, 2, 1, 0.9)
    assert rel_inf(st8.eng.a['Y'], y1[8:16]) < 1e-6
    assert rel_inf(st8.x, x1[8:16]) < 1e-4
    # (3) input-gradient pass is linear in the cotangent: backward(2g) == 2 backward(g) (exact in binary fp)
    g = torch.randn(64, 256, 256, 4, device=DEV)
    g[..., 3] = 0
    st.eng.forward(st.x)
    a = st.eng.backward(g).clone()
    b2 = st.eng.backward(2 * g).clone()
    assert rel_l2(b2, 2 * a) < 1e-6
    # (3b) the whole iteration is run-to-run reproducible (gather-form backward passes, fixed-order reductions)
    c2 = st.eng.backward(2 * g).clone()
    assert torch.equal(b2, c2)
    # (4) outputs stay finite and in range over more iterations; best images are tracked only for successes
    for _ in range(3):
        st.iteration(True, 5, 2, 1, 0.9)
    cam, prj = st.results()
    assert torch.isfinite(cam).all() and torch.isfinite(prj).all() and prj.min() >= 0 and prj.max() <= 1
    assert cam.shape == (64, 3, 256, 256) and prj.shape == (64, 3, 256, 256)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('targeted,confidence', [(True, 0), (False, 40), (False, 0)])
def test_perc_al_adversary_projector(hip, golden_dir, targeted, confidence):
    """PerC_AL.adversary_projector on HIP vs the oracle: first iterations tightly (the loop is chaotic afterwards),
    API / error behaviour of perc_al/__init__.py:153,176-178, and 8-bit quantised output."""
    from spaa_amd.perc_al import PerC_AL
    z = load(golden_dir, 'percal_64_targeted' if targeted else 'percal_64_untargeted')
    csd = syn.resnet18_state_dict(2, logit_gain=float(z['gain']))
    insz, crop = tuple(int(v) for v in z['input_sz']), tuple(int(v) for v in z['crop'])
    clf = hip['clf'].Classifier('resnet18', DEV, state_dict=csd, input_sz=insz)
    oclf = so.OracleClassifier('resnet18', csd, input_sz=insz)
    scene = syn.scenes(1, 1, (64, 64)).expand(8, -1, -1, -1).contiguous()
    labels = torch.tensor(z['targets'])
    d_thr = float(z['d_thr'])
    otr = []
    so.perc_al_adversary_projector(oclf, scene, labels, d_thr, targeted, crop, 50, 1., 0.5, confidence, stop_after=3,
                                   trace=otr)
    att = PerC_AL(device=DEV, max_iterations=50, alpha_l_init=1, alpha_c_init=0.5, confidence=confidence)
    with pytest.raises(ValueError):
        att.adversary_projector(clf, scene + 1.0, labels, None, d_thr, targeted, crop)
    if targeted:
        assert PerC_AL(device=DEV, max_iterations=5, confidence=40).adversary_projector(clf, scene, labels, None, d_thr,
                                                                                      True, crop) is None
    tr = []
    out = att.adversary_projector(clf, scene, labels, None, d_thr, targeted, crop, trace=tr)
    assert out.shape == scene.shape and out.min() >= 0 and out.max() <= 1
    assert (torch.round(out * 255) / 255 - out).abs().max() < 1e-6
    # iteration 0 starts from identical state: everything agrees to rounding
    st0, stats0, d0 = tr[0]
    o0 = otr[0]
    assert rel_inf(d0, o0['delta']) < 1e-4
    assert np.allclose(stats0[:, 3].cpu().numpy(), o0['color_dis'].numpy(), rtol=1e-4)
    assert np.allclose(stats0[:, 1].cpu().numpy(), o0['caml2'].numpy(), rtol=1e-4)
    # (top-1 probability of a sharp softmax: dp = p (1 - p) dlogit, the fixtures' logit gain makes 1e-5 relative logit
    # error ~2e-4 in p)
    assert np.allclose(stats0[:, 0].cpu().numpy(), o0['p1'], atol=5e-4)
    assert (st0[:, 3].cpu().numpy() == o0['top1']).all()
    assert (st0[:, 0].cpu().numpy().astype(bool) == o0['isadv'].numpy()).all()
    print(f'PerC-AL targeted={targeted} conf={confidence}: delta rel Linf after it 0/1/2 = '
          f'{[round(rel_inf(tr[k][2], otr[k]["delta"]), 6) for k in range(3)]}')
    assert rel_inf(tr[2][2], otr[2]['delta']) < 1e-4   # (measured 2e-6 ... 3.4e-5: profiles/r02_parity.txt)
    # (the 50-iteration result itself is chaotic, like spaa(): tests/test_oracle_golden.py::test_reference_sensitivity_envelope)
    ref = torch.from_numpy(z['x_adv_best'])
    assert ref.shape == out.shape


def test_vgg16_classifier_vs_oracle(hip):
    """VGG-16 body (config 5 of BASELINE.json) on tapconv + generic pooling vs the oracle restatement."""
    csd = syn.vgg16_state_dict(3, logit_gain=5.0, fc_width=512)
    for (h, crop, insz, b) in [(64, (60, 60), (48, 48), 2), (256, (240, 240), (224, 224), 1)]:
        torch.manual_seed(2)
        im = torch.rand(b, 3, h, h, requires_grad=True)
        raw, p, idx = so.OracleClassifier('vgg16', csd, input_sz=insz)(im, crop)
        r = torch.randn(b, 1000)
        (raw * r).sum().backward()
        clf = hip['clf'].Classifier('vgg16', DEV, state_dict=csd, input_sz=insz)
        im2 = im.detach().clone().to(DEV).requires_grad_(True)
        raw2, p2, idx2 = clf(im2, crop)
        (raw2 * r.to(DEV)).sum().backward()
        assert rel_inf(raw2, raw) < 1e-5
        # 13 ReLU convs + 5 max-pools without skip connections: a gate / arg-max flip reaches the whole image
        assert rel_l2(im2.grad, im.grad) < 5e-3 and outlier_fraction(im2.grad, im.grad, 1e-3) < 5e-2
        assert (idx2[:, 0] == idx[:, 0]).all()


def test_inception_v3_classifier_vs_oracle(hip):
    """Inception-v3 body (config 3 of BASELINE.json: 299x299 area-resized crop, transform_input=True) vs the oracle."""
    csd = syn.inception_v3_state_dict(4, logit_gain=20.0)
    for (h, crop, insz, b) in [(128, (120, 120), (107, 107), 2), (256, (240, 240), (299, 299), 1)]:
        torch.manual_seed(5)
        im = torch.rand(b, 3, h, h, requires_grad=True)
        raw, p, idx = so.OracleClassifier('inception_v3', csd, input_sz=insz)(im, crop)
        r = torch.randn(b, 1000)
        (raw * r).sum().backward()
        clf = hip['clf'].Classifier('inception_v3', DEV, state_dict=csd, input_sz=insz)
        im2 = im.detach().clone().to(DEV).requires_grad_(True)
        raw2, p2, idx2 = clf(im2, crop)
        (raw2 * r.to(DEV)).sum().backward()
        assert rel_inf(raw2, raw) < 2e-5
        assert rel_l2(im2.grad, im.grad) < 5e-3 and outlier_fraction(im2.grad, im.grad, 1e-3) < 5e-2
        assert (idx2[:, 0] == idx[:, 0]).all()


def test_compennet_pp_forward_and_perc_al_glue(hip, golden_dir):
    """Next-row components: CompenNet++ forward on HIP vs the oracle, and perc_al_compennet_pp glue."""
    from spaa_amd.models import CompenNetPlusplus, WarpingNet
    from spaa_amd.perc_al import perc_al_compennet_pp
    sz = (64, 64)
    sd = syn.compennet_pp_state_dict(5, out_size=sz)
    net = CompenNetPlusplus(WarpingNet(out_size=sz))
    net.load_state_dict(sd)
    net = net.to(DEV)
    torch.manual_seed(9)
    x = syn.scenes(21, 3, sz)
    s = syn.scenes(22, 1, sz)
    ref = so.compennet_pp_forward(sd, x, s.expand(3, -1, -1, -1), sz)
    out = net(x.to(DEV), s.to(DEV))
    assert rel_inf(out, ref) < 1e-5
    z = load(golden_dir, 'compennet_pp_64')  # the reference's own output
    out = net(torch.from_numpy(z['x']).to(DEV), torch.from_numpy(z['s']).to(DEV))
    assert np.abs(out.cpu().numpy() - z['y']).max() < 1e-5
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    clf = hip['clf'].Classifier('resnet18', DEV, state_dict=csd, input_sz=(56, 56))
    setup = dict(classifier_crop_sz=(60, 60), prj_brightness=0.5, prj_im_sz=sz)
    cam, prj = perc_al_compennet_pp(net, clf, None, [204, 291], True, s[0], 2, DEV, setup)
    assert cam.shape == (2, 3, 64, 64) and prj.shape == (2, 3, 64, 64) and torch.isfinite(prj).all()


@pytest.mark.parametrize('tile', [0, 25, 27, 30, 34, 36, 37, 39, 40, 41, 42, 43, 45, 46, 48, 49, 53])
def test_folded_deconv_with_epilogue(hip, tile):
    """Kernel-2 stride-2 ConvTranspose2d with the four parity classes folded into the GEMM rows (spaa_tapconv_t.nfold),
    fused residual + ReLU, against torch; every DMA-staged tile shape."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(17)
    for ci, co, h, w in [(64, 32, 9, 7), (32, 8, 16, 16), (96, 36, 5, 6)]:
        x = torch.randn(2, ci, h, w)
        wt = torch.randn(ci, co, 2, 2) / (ci * 4) ** 0.5
        bias, add = torch.randn(co), torch.randn(2, co, 2 * h, 2 * w)
        y = F.relu(F.conv_transpose2d(x, wt, bias, 2, 0) + add)
        plan = cp.deconv_fwd_plan(wt, bias, 2, 0, DEV)
        assert plan.nfold == 4
        out = torch.zeros(2, 2 * h, 2 * w, co, device=DEV)
        try:
            cp.FORCE_TILE = tile
            plan.run(nhwc(x, ci).to(DEV), out, add=nhwc(add).to(DEV), act=lib.ACT_RELU)
        finally:
            cp.FORCE_TILE = 0
        assert rel_inf(nchw(out.cpu(), co), y) < 1e-5


def test_img_dists_metrics(hip, golden_dir):
    """Next-row component: calc_img_dists on HIP vs the reference's values and vs the oracle on other sizes."""
    from spaa_amd import metrics
    z = load(golden_dir, 'img_dists')
    x, y = torch.from_numpy(z['x']), torch.from_numpy(z['y'])
    got = np.array(metrics.calc_img_dists(x.to(DEV), y))          # mixed devices, as the reference tolerates
    assert np.abs(got / z['dists'] - 1).max() < 2e-5, (got, z['dists'])
    torch.manual_seed(4)
    for shape in [(3, 37, 53), (2, 3, 256, 256), (1, 3, 16, 16)]:   # ragged tiles, full size, one tile; 3-D input
        a = torch.rand(*shape)
        b = (a + 0.05 * torch.randn(*shape)).clamp(0, 1)
        want = np.array(so.calc_img_dists(a, b))
        got = np.array(metrics.calc_img_dists(a, b))
        assert np.abs(got / want - 1).max() < 2e-5, (shape, got, want)
        single = np.array([metrics.psnr(a, b), metrics.rmse(a, b), metrics.ssim(a, b), metrics.l2_norm(a, b),
                           metrics.linf_norm(a, b), metrics.deltaE(a, b)])
        assert np.array_equal(single, got)
    with pytest.raises(ValueError):
        metrics.calc_img_dists(torch.rand(3, 8, 8), torch.rand(3, 8, 9))


def test_tapconv_fuzz_all_kernels(hip):
    """Seeded random layer shapes x every kernel family (forced tile; a tile that does not apply to a shape falls back
    inside ConvPlan.run and is still checked): conv / conv-dgrad / transposed conv against torch on the CPU, with ragged
    sizes, channel counts that are not multiples of 4, channel windows (offsets into wider buffers), residual + ReLU."""
    cp, lib = hip['cp'], hip['lib']
    rng = np.random.default_rng(2024)
    tiles = [0, 1, 5, 6, 9, 10, 11, 12, 15, 16, 17, 18, 19, 20, 22, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39,
             40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 225, 234, 236, 242, 434, 248, 948, 949, 950, 953]
    worst = 0.0
    try:
        for case in range(60):
            kind = ['conv', 'dgrad', 'deconv'][case % 3]
            ci = int(rng.choice([3, 4, 6, 8, 32, 64, 96, 128]))
            co = int(rng.choice([2, 3, 4, 5, 32, 48, 64, 96, 130]))
            k = int(rng.choice([1, 2, 3])) if kind == 'deconv' else int(rng.choice([1, 3, 5]))
            s = 2 if kind == 'deconv' else int(rng.choice([1, 2]))
            pad = 0 if kind == 'deconv' and k < 3 else k // 2
            h, w, b = int(rng.integers(5, 23)), int(rng.integers(5, 27)), int(rng.integers(1, 4))
            tile = int(rng.choice(tiles))
            g = torch.Generator().manual_seed(case)
            if kind == 'deconv':
                op = 1 if k == 3 else 0
                x = torch.randn(b, ci, h, w, generator=g)
                wt = torch.randn(ci, co, k, k, generator=g) / (ci * k * k) ** 0.5
                bias = torch.randn(co, generator=g)
                ref = F.conv_transpose2d(x, wt, bias, 2, pad, op)
                plan = cp.deconv_fwd_plan(wt, bias, 2, pad, DEV)
                inp, cout = x, co
            elif kind == 'conv':
                x = torch.randn(b, ci, h, w, generator=g)
                wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
                bias = torch.randn(co, generator=g)
                ref = F.conv2d(x, wt, bias, s, pad)
                plan = cp.conv_fwd_plan(wt, bias, s, pad, DEV)
                inp, cout = x, co
            else:
                wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
                ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
                gy = torch.randn(b, co, ho, wo, generator=g)
                ref = torch.nn.grad.conv2d_input((b, ci, h, w), wt, gy, s, pad)
                plan = cp.conv_dgrad_plan(wt, s, pad, DEV)
                inp, cout = gy, ci
            add = torch.randn(ref.shape, generator=g)
            want = F.relu(ref + add)
            # operands live in wider NHWC buffers at channel offsets (multiples of 4)
            cin_p = plan.cin_p
            ioff, ooff, aoff = 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 3)), 4 * int(rng.integers(0, 3))
            ibuf = torch.randn(inp.shape[0], inp.shape[2], inp.shape[3], cin_p + ioff + 4, generator=g)
            ibuf[..., ioff:ioff + cin_p] = nhwc(inp, cin_p)
            cs_out = (cout + 3) // 4 * 4 + ooff + 4
            obuf = torch.full((ref.shape[0], ref.shape[2], ref.shape[3], cs_out), 7.0)
            abuf = torch.zeros(ref.shape[0], ref.shape[2], ref.shape[3], (cout + 3) // 4 * 4 + aoff)
            abuf[..., aoff:aoff + cout] = nhwc(add)
            od = obuf.to(DEV)
            cp.FORCE_TILE = tile
            plan.run(ibuf.to(DEV), od, add=abuf.to(DEV), act=lib.ACT_RELU, in_coff=ioff, out_coff=ooff, add_coff=aoff)
            cp.FORCE_TILE = 0
            got = od.cpu()
            err = rel_inf(nchw(got[..., ooff:ooff + cout]), want)
            worst = max(worst, err)
            assert err < 2e-5, (case, kind, ci, co, k, s, pad, h, w, b, tile, err)
            # nothing outside the channel window was written
            assert (got[..., :ooff] == 7.0).all() and (got[..., ooff + cout:] == 7.0).all(), (case, kind, tile)
    finally:
        cp.FORCE_TILE = 0
    print(f'fuzz: worst relative error {worst:.2e}')


def test_tapconv_fuzz_epilogues(hip):
    """Seeded random epilogue combinations (activation, gate mode, second gate / second output, bias) x kernel
    families on random conv shapes, against torch on the CPU."""
    cp, lib = hip['cp'], hip['lib']
    rng = np.random.default_rng(7)
    tiles = [0, 6, 16, 18, 22, 25, 27, 30, 31, 33, 34, 35, 36, 37, 39, 40, 41, 42, 43, 44, 45, 46, 48, 49, 50, 52, 53, 234, 436]
    try:
        for case in range(60):
            ci, co = int(rng.choice([32, 64, 96])), int(rng.choice([4, 32, 64, 100, 128]))
            k, s = int(rng.choice([1, 3])), int(rng.choice([1, 2]))
            h, w, b = int(rng.integers(6, 20)), int(rng.integers(6, 24)), int(rng.integers(1, 4))
            tile = int(rng.choice(tiles))
            act = int(rng.choice([lib.ACT_NONE, lib.ACT_RELU, lib.ACT_RELU_CLAMP1, lib.ACT_LEAKY01]))
            gmode = int(rng.choice([0, lib.GATE_POS, lib.GATE_POS_LE1, lib.GATE_MUL]))
            use_gate2 = bool(rng.integers(0, 2)) and act != lib.ACT_RELU_CLAMP1
            use_bias, use_add = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            g = torch.Generator().manual_seed(1000 + case)
            x = torch.randn(b, ci, h, w, generator=g)
            wt = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
            bias = torch.randn(co, generator=g) if use_bias else None
            y = F.conv2d(x, wt, bias, s, k // 2)
            add = torch.randn(y.shape, generator=g)
            gate = torch.randn(y.shape, generator=g) * 0.8
            gate2 = torch.randn(y.shape, generator=g)
            t = y + add if use_add else y
            pre = None
            if act == lib.ACT_RELU:
                t = F.relu(t)
            elif act == lib.ACT_RELU_CLAMP1:
                pre = F.relu(t)
                t = pre.clamp(max=1)
            elif act == lib.ACT_LEAKY01:
                t = F.leaky_relu(t, 0.1)
            if gmode == lib.GATE_POS:
                t = t * (gate > 0)
            elif gmode == lib.GATE_POS_LE1:
                t = t * ((gate > 0) & (gate <= 1))
            elif gmode == lib.GATE_MUL:
                t = t * gate
            plan = cp.conv_fwd_plan(wt, bias, s, k // 2, DEV)
            cs = (co + 3) // 4 * 4
            out = torch.zeros(b, y.shape[2], y.shape[3], cs, device=DEV)
            aux = torch.zeros_like(out) if (use_gate2 or act == lib.ACT_RELU_CLAMP1) else None
            cp.FORCE_TILE = tile
            plan.run(nhwc(x).to(DEV), out, add=nhwc(add, cs).to(DEV) if use_add else None, act=act,
                     gate=nhwc(gate, cs).to(DEV) if gmode else None, gate_mode=gmode if gmode else lib.GATE_POS,
                     aux_out=aux, gate2=nhwc(gate2, cs).to(DEV) if use_gate2 else None)
            cp.FORCE_TILE = 0
            info = (case, ci, co, k, s, h, w, b, tile, act, gmode, use_gate2, use_bias, use_add)
            assert rel_inf(nchw(out.cpu(), co), t) < 2e-5, info
            if use_gate2:
                assert rel_inf(nchw(aux.cpu(), co), t * (gate2 > 0)) < 2e-5, info
            elif act == lib.ACT_RELU_CLAMP1:
                assert rel_inf(nchw(aux.cpu(), co), pre) < 2e-5, info
    finally:
        cp.FORCE_TILE = 0


@pytest.mark.parametrize('tile', [48, 49, 50, 51, 52, 53, 54, 248, 948, 949, 950, 951, 952, 953, 954])
def test_persistent_launch_walks_several_tiles(hip, tile):
    """Persistent launches of the DMA-staged kernel (a workgroup walks its XCD's tiles and overlaps a tile's epilogue
    with the next tile's first gathers), forced down to 8 workgroups so that every workgroup handles many tiles:
    conv, strided-class transposed conv and folded transposed conv, ragged sizes, against torch."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(5)
    cp.DEBUG_PERSIST_CAP = 24 if tile >= 900 else 8   # stream-K: 3 workgroups per XCD, so tiles do get cut
    try:
        for ci, co, k, s, h, w, b in [(64, 96, 3, 1, 37, 41, 3), (32, 160, 3, 2, 50, 33, 2), (128, 40, 1, 1, 29, 31, 4)]:
            x = torch.randn(b, ci, h, w)
            wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
            bias = torch.randn(co)
            y = F.conv2d(x, wt, bias, s, k // 2)
            add, gate = torch.randn_like(y), torch.randn_like(y)
            plan = cp.conv_fwd_plan(wt, bias, s, k // 2, DEV)
            cs = (co + 3) // 4 * 4
            out = torch.zeros(b, y.shape[2], y.shape[3], cs, device=DEV)
            cp.FORCE_TILE = tile
            plan.run(nhwc(x).to(DEV), out, add=nhwc(add, cs).to(DEV), act=lib.ACT_RELU, gate=nhwc(gate, cs).to(DEV))
            cp.FORCE_TILE = 0
            assert rel_inf(nchw(out.cpu(), co), F.relu(y + add) * (gate > 0)) < 2e-5, (ci, co, k, s)
        xt, wtt, bt = torch.randn(2, 64, 19, 23), torch.randn(64, 48, 3, 3) / 24, torch.randn(48)
        for kk, pad, op in [(3, 1, 1), (2, 0, 0)]:
            wk = wtt[:, :, :kk, :kk].contiguous()
            ref = F.conv_transpose2d(xt, wk, bt, 2, pad, op)
            tplan = cp.deconv_fwd_plan(wk, bt, 2, pad, DEV)
            outt = torch.zeros(2, ref.shape[2], ref.shape[3], 48, device=DEV)
            cp.FORCE_TILE = tile
            tplan.run(nhwc(xt).to(DEV), outt)
            cp.FORCE_TILE = 0
            assert rel_inf(nchw(outt.cpu()), ref) < 2e-5, (kk, tile)
    finally:
        cp.FORCE_TILE = 0
        cp.DEBUG_PERSIST_CAP = 0


@pytest.mark.parametrize('cam_sz,prj_sz,b,mask', [((96, 160), (96, 160), 3, 'ones'), ((128, 128), (128, 128), 4, 'rect'),
                                                  ((64, 96), (80, 112), 2, 'ones')])
def test_first_iteration_other_sizes(hip, cam_sz, prj_sz, b, mask):
    """The whole loop body at sizes the tune table has never seen (kernels chosen by ConvPlan._default_tile), non-square
    images, projector size != camera size: the first iteration from identical state against the oracle, gate-aware:
    <= 1e-4 relative L_inf on the updated projector image (BASELINE.json's bar) for every sample whose gates agree, and
    for every sample with the oracle's gates."""
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    crop = (cam_sz[0] - 8, cam_sz[0] - 8) if cam_sz[0] <= cam_sz[1] else (cam_sz[1] - 8, cam_sz[1] - 8)
    insz = (crop[0] - 8, crop[1] - 8)
    _first_iteration_gate_aware(hip, 'resnet18', csd, insz, cam_sz, crop, [204, 291, 7, 950][:b], 3, prj_sz=prj_sz, mask=mask,
                                scene_seed=7)

# ---------------------------------------------------------------------------------------------------------------
# round 2: reference-pinned preprocessing (a6/a7), loop-level runs with the other classifier bodies (configs[2], [4]),
# API-compatibility paths (differentiable colour functions, engine reuse under autograd, foreign classifiers)
@pytest.mark.parametrize('name', ['preproc_240_224', 'preproc_240_299', 'preproc_nonsq_small'])
def test_preproc_and_classifier_wrapper_vs_reference_fixture(hip, golden_dir, name):
    """spaa_preproc_fwd / spaa_preproc_bwd and the Classifier wrapper against what the REFERENCE's own
    Classifier.classify + img_proc.center_crop/resize produced (tests/golden/make_golden.py: gen_preproc): 240->224
    area down-sampling, 240->299 area UP-sampling (Inception-v3), a non-square image."""
    from spaa_amd.classifier import Classifier
    z = load(golden_dir, name)
    rng = np.random.default_rng(int(z['seed']))
    bsz, im_hw, insz = int(z['bsz']), tuple(int(v) for v in z['im_hw']), tuple(int(v) for v in z['input_sz'])
    crop = tuple(int(v) for v in z['crop'])
    im = torch.from_numpy(rng.random((bsz, 3, *im_hw)).astype(np.float32))
    r = torch.from_numpy(rng.standard_normal((bsz, 3, *insz)).astype(np.float32))
    csd = syn.resnet18_state_dict(2, logit_gain=float(z['gain']))
    clf = Classifier('resnet18', DEV, state_dict=csd, input_sz=insz)
    M = hip['models']
    eng = clf.engine(bsz, im_hw, crop)
    eng.forward(M.to_nhwc4(im.to(DEV)))
    pre = M.to_nchw(eng.pre).cpu()
    assert np.abs(pre.numpy() - z['pre']).max() <= 2e-6 * np.abs(z['pre']).max()
    # adjoint of the preprocessing alone: spaa_preproc_bwd on the cotangent r
    lib = hip['lib']
    g_pre = M.to_nhwc4(r.to(DEV))
    g_y = torch.zeros(bsz, *im_hw, 4, device=DEV)
    lib.call('spaa_preproc_bwd', lib.ptr(g_pre), lib.ptr(g_y), bsz, im_hw[0], im_hw[1], eng.cy0, eng.cx0, crop[0], crop[1],
             insz[0], insz[1], eng._std)
    g = M.to_nchw(g_y).cpu().numpy()
    assert np.abs(g - z['grad_im']).max() <= 1e-5 * np.abs(z['grad_im']).max()
    raw, p, idx = clf(im.to(DEV), crop)
    assert np.abs(raw.detach().cpu().numpy() - z['raw_score']).max() <= 1e-4 * np.abs(z['raw_score']).max()
    assert (idx[:, :5] == z['idx5']).all() and np.allclose(p[:, :5], z['p5'], atol=2e-4)
    raw8, _, idx8 = clf((im[0] * 255).to(torch.uint8), crop)   # 3-D uint8 input (classifier.py:56-57)
    assert np.abs(raw8.detach().cpu().numpy() - z['raw_score_u8']).max() <= 1e-4 * np.abs(z['raw_score_u8']).max()
    assert (idx8[:, :5] == z['idx5_u8']).all()


def test_color_functions_are_differentiable(hip, golden_dir):
    """rgb2lab_diff / ciede2000_diff carry gradients like the reference's (`_diff`): autograd through the HIP ops vs
    autograd through the oracle, w.r.t. BOTH colours (PerC-AL differentiates the second, perc_al/__init__.py:197)."""
    dcf = hip['dcf']
    z = load(golden_dir, 'color_kat')
    a0, b0 = torch.from_numpy(z['rgb_a']), torch.from_numpy(z['rgb_b'])
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    wgt = torch.linspace(0.5, 1.5, a0.shape[0] * a0.shape[2] * a0.shape[3]).view(a0.shape[0], a0.shape[2], a0.shape[3])
    (so.ciede2000_diff(so.rgb2lab_diff(a), so.rgb2lab_diff(b)) * wgt).sum().backward()
    ag, bg = a0.clone().to(DEV).requires_grad_(True), b0.clone().to(DEV).requires_grad_(True)
    de = dcf.ciede2000_diff(dcf.rgb2lab_diff(ag, DEV), dcf.rgb2lab_diff(bg, DEV), DEV)
    assert de.requires_grad and de.shape == wgt.shape
    (de * wgt.to(DEV)).sum().backward()
    lab_a_ref, lab_b_ref = torch.from_numpy(z['lab_a']), torch.from_numpy(z['lab_b'])
    chroma = torch.minimum(lab_a_ref[:, 1:].norm(dim=1), lab_b_ref[:, 1:].norm(dim=1))
    for ours, ref in ((ag.grad.cpu(), a.grad), (bg.grad.cpu(), b.grad)):
        fin = torch.isfinite(ref)
        well = (chroma > 1.0)[:, None].expand_as(ref) & fin   # (grey pixels: hue is rounding noise, see the KAT test)
        assert ((ours - ref)[well].abs().max() / ref[fin].abs().max()) < 1e-4
    # the Lab conversion alone
    x = torch.rand(2, 3, 9, 11)
    r = torch.randn(2, 3, 9, 11)
    xc = x.clone().requires_grad_(True)
    (so.rgb2lab_diff(xc) * r).sum().backward()
    xg = x.clone().to(DEV).requires_grad_(True)
    (dcf.rgb2lab_diff(xg) * r.to(DEV)).sum().backward()
    assert rel_inf(xg.grad, xc.grad) < 1e-5


def test_autograd_survives_engine_reuse(hip, golden_dir):
    """y1 = pcnet(x1, s); y2 = pcnet(x2, s); (y1 + y2).backward(): the cached engine's workspaces hold the SECOND forward
    when the first call's backward runs — it must recompute, not silently use the wrong activations.  Same for the
    classifier, and two attack states built from one PCNet/Classifier must not share workspaces."""
    z = load(golden_dir, 'pcnet_64')
    cam_sz = tuple(int(v) for v in z['cam_sz'])
    sd = syn.pcnet_state_dict(int(z['seed']), cam_sz=cam_sz, mask=str(z['mask']))
    pc = make_pcnet(hip, sd, cam_sz)
    s = torch.from_numpy(z['s'])
    x1 = syn.scenes(5, 2, cam_sz)
    x2 = syn.scenes(6, 2, cam_sz)
    r1, r2 = torch.randn(2, 3, *cam_sz), torch.randn(2, 3, *cam_sz)
    g_ref = []
    for x, r in ((x1, r1), (x2, r2)):
        xc = x.clone().requires_grad_(True)
        (so.pcnet_forward(sd, xc, s) * r).sum().backward()
        g_ref.append(xc.grad)
    a, b = x1.clone().to(DEV).requires_grad_(True), x2.clone().to(DEV).requires_grad_(True)
    y1 = pc(a, s.to(DEV))
    y2 = pc(b, s.to(DEV))
    ((y1 * r1.to(DEV)).sum() + (y2 * r2.to(DEV)).sum()).backward()
    assert rel_l2(a.grad, g_ref[0]) < 1e-4 and rel_l2(b.grad, g_ref[1]) < 1e-4
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    clf = hip['clf'].Classifier('resnet18', DEV, state_dict=csd, input_sz=(56, 56))
    oc = so.OracleClassifier('resnet18', csd, input_sz=(56, 56))
    q = torch.randn(2, 1000)
    c_ref = []
    for x in (x1, x2):
        xc = x.clone().requires_grad_(True)
        (oc(xc, (60, 60))[0] * q).sum().backward()
        c_ref.append(xc.grad)
    a, b = x1.clone().to(DEV).requires_grad_(True), x2.clone().to(DEV).requires_grad_(True)
    l1 = clf(a, (60, 60))[0]
    l2 = clf(b, (60, 60))[0]
    ((l1 * q.to(DEV)).sum() + (l2 * q.to(DEV)).sum()).backward()
    # (aliasing would give O(1) errors; a ReLU gate within rounding of zero gives ~5e-3: DESIGN.md section 4)
    assert rel_l2(a.grad, c_ref[0]) < 2e-2 and rel_l2(b.grad, c_ref[1]) < 2e-2
    setup = dict(classifier_crop_sz=(60, 60), prj_brightness=0.5, prj_im_sz=cam_sz)
    A = hip['attack']
    st1 = A.AttackState(pc, clf, [204, 291], s[:1], 'camdE_caml2', setup, DEV)
    st2 = A.AttackState(pc, clf, [7, 950], syn.scenes(9, 1, cam_sz), 'camdE_caml2', setup, DEV)
    assert st1.eng is not st2.eng and st1.clf is not st2.clf
    assert st1.eng.a['Y'].data_ptr() != st2.eng.a['Y'].data_ptr()
    st1.iteration(True, 5, 2, 1, 0.9)
    y_before = st1.eng.a['Y'].clone()
    st2.iteration(True, 5, 2, 1, 0.9)
    assert torch.equal(st1.eng.a['Y'], y_before) and torch.equal(st1.eng.scene, st1.scene4)


def test_warping_net_forward_is_differentiable(hip, golden_dir):
    """WarpingNet.forward through the reference interface: value and (gather-form, deterministic) input gradient vs
    F.grid_sample on the oracle's fine grid."""
    z = load(golden_dir, 'pcnet_nonsq')
    cam_sz = tuple(int(v) for v in z['cam_sz'])
    sd = syn.pcnet_state_dict(int(z['seed']), cam_sz=cam_sz, mask=str(z['mask']))
    pc = make_pcnet(hip, sd, cam_sz)
    x = syn.scenes(3, 2, tuple(int(v) for v in z['prj_sz']))
    r = torch.randn(2, 3, *cam_sz)
    xc = x.clone().requires_grad_(True)
    (so.warp(sd, xc, cam_sz) * r).sum().backward()
    xg = x.clone().to(DEV).requires_grad_(True)
    y = pc.warping_net(xg)
    (y * r.to(DEV)).sum().backward()
    assert rel_inf(y, so.warp(sd, x, cam_sz)) < 1e-5
    g1 = xg.grad.clone()
    assert rel_inf(g1, xc.grad) < 1e-4
    xg.grad = None
    (pc.warping_net(xg) * r.to(DEV)).sum().backward()
    assert torch.equal(xg.grad, g1)   # run-to-run identical: no float atomics on this path either


def test_spaa_accepts_a_foreign_classifier(hip, golden_dir):
    """The reference's spaa() takes any callable classifier(im, crop_sz) -> (raw_score, p, idx)
    (projector_based_attack.py:266).  A foreign callable takes the autograd route (PCNet + stealth loss on HIP, classifier
    by torch.autograd); its first iterations must agree with the fused path and with the oracle."""
    A = hip['attack']
    z = load(golden_dir, 'spaa_64_near')
    sd, pc, clf, oclf, scene, setup = _setup_case(hip, z)
    targets = [int(t) for t in z['targets']]
    d_thr, stealth = float(z['d_thr']), str(z['stealth'])

    class Foreign:   # not a spaa_amd.Classifier: an arbitrary callable with the reference's contract
        def __call__(self, im, crop_sz):
            return clf(im, crop_sz)

    tr_f, tr_o = [], []
    cam_f, prj_f = A.spaa(pc, Foreign(), None, targets, True, scene, d_thr, stealth, DEV, setup, iters=3, trace=tr_f)
    so.spaa(sd, oclf, targets, True, scene, d_thr, stealth, setup, iters=3, trace=tr_o)
    assert cam_f.shape == (len(targets), 3, 64, 64) and prj_f.min() >= 0 and prj_f.max() <= 1
    assert (tr_f[0]['top1'].cpu().numpy() == tr_o[0]['top1']).all()
    assert np.allclose(tr_f[0]['caml2'].cpu().numpy(), tr_o[0]['caml2'], rtol=1e-4)
    assert np.allclose(tr_f[0]['camdE'].cpu().numpy(), tr_o[0]['camdE'], rtol=1e-4)
    e0 = rel_inf(tr_f[0]['prj_adv'], torch.from_numpy(tr_o[0]['prj_adv']))
    print(f'foreign-classifier route: projector image rel Linf vs oracle after iteration 0: {e0:.2e}')
    assert e0 < 1e-4
    # untargeted, prjl2 term, B = 1 (the reference's first call shape, :107)
    cam1, prj1 = A.spaa(pc, Foreign(), None, [int(z['top1'][0][0])], False, scene[0], d_thr, 'camdE_caml2_prjl2', DEV, setup,
                        iters=2)
    assert cam1.shape == (1, 3, 64, 64) and torch.isfinite(prj1).all()


@pytest.mark.parametrize('targeted,confidence', [(True, 0), (False, 40)])
def test_perc_al_foreign_classifier(hip, golden_dir, targeted, confidence):
    """perc_al/__init__.py:181 calls ANY `classifier(inputs + delta, cp_sz)`; a callable that is not a spaa_amd.Classifier takes
    the autograd route of PerC_AL.adversary_projector (colour distance on the HIP ops, the classifier by torch.autograd).  Its
    first iterations must agree with the fused path and with the oracle; output contract as the reference's."""
    from spaa_amd.perc_al import PerC_AL
    z = load(golden_dir, 'percal_64_targeted' if targeted else 'percal_64_untargeted')
    csd = syn.resnet18_state_dict(2, logit_gain=float(z['gain']))
    insz, crop = tuple(int(v) for v in z['input_sz']), tuple(int(v) for v in z['crop'])
    clf = hip['clf'].Classifier('resnet18', DEV, state_dict=csd, input_sz=insz)
    oclf = so.OracleClassifier('resnet18', csd, input_sz=insz)
    scene = syn.scenes(1, 1, (64, 64)).expand(8, -1, -1, -1).contiguous()
    labels = torch.tensor(z['targets'])
    d_thr = float(z['d_thr'])

    class Foreign:
        def __call__(self, im, crop_sz):
            return clf(im, crop_sz)

    otr, ftr = [], []
    so.perc_al_adversary_projector(oclf, scene, labels, d_thr, targeted, crop, 50, 1., 0.5, confidence, stop_after=3, trace=otr)
    att3 = PerC_AL(device=DEV, max_iterations=50, alpha_l_init=1, alpha_c_init=0.5, confidence=confidence)
    # run only three iterations of the 50-iteration schedule: stop through the trace hook
    class Stop(Exception):
        pass

    class Tr(list):
        def append(self, v):
            list.append(self, v)
            if len(self) == 3:
                raise Stop

    ftr = Tr()
    try:
        att3.adversary_projector(Foreign(), scene, labels, None, d_thr, targeted, crop, trace=ftr)
    except Stop:
        pass
    assert len(ftr) == 3
    e = [rel_inf(ftr[k]['delta'], otr[k]['delta']) for k in range(3)]
    print(f'PerC-AL foreign-classifier route targeted={targeted} conf={confidence}: delta rel Linf vs oracle after it 0/1/2 = {e}')
    assert e[0] < 1e-4 and np.allclose(ftr[0]['color_dis'].cpu().numpy(), otr[0]['color_dis'].numpy(), rtol=1e-4)
    assert (ftr[0]['top1'].cpu().numpy() == otr[0]['top1']).all() and (ftr[0]['isadv'].cpu().numpy() == otr[0]['isadv'].numpy()).all()
    assert e[2] < 1e-3   # (free-running after iteration 0: gate flips of the autograd route's own classifier launches)
    with pytest.raises(ValueError):
        att3.adversary_projector(Foreign(), scene + 1.0, labels, None, d_thr, targeted, crop)
    short = PerC_AL(device=DEV, max_iterations=2, alpha_l_init=1, alpha_c_init=0.5, confidence=confidence)
    out = short.adversary_projector(Foreign(), scene, labels, None, d_thr, targeted, crop)
    assert out.shape == scene.shape and out.min() >= 0 and out.max() <= 1
    changed = (out != scene.to(DEV)).flatten(1).any(dim=1)   # (a sample that never became adversarial keeps its input, :165)
    assert (torch.round(out[changed] * 255) / 255 - out[changed]).abs().max() < 1e-6 if changed.any() else True


def _first_iteration_gate_aware(hip, body, csd, insz, im_sz, crop, targets, seed, prj_sz=None, mask='ones', targeted=True,
                                scene_seed=None, d_thr=5, stealth='camdE_caml2', golden_it0=None, storage='f32', tol=None,
                                n_scenes=1):
    """First iteration of the fused loop vs the oracle, gate-aware (tests/gates.py): 1e-4 on every sample whose gates agree
    with the oracle's, and on every sample with the oracle's gates in the HIP backward."""
    import gates
    A, M = hip['attack'], hip['models']
    prj_sz = tuple(prj_sz) if prj_sz is not None else tuple(im_sz)
    sd = syn.pcnet_state_dict(seed, cam_sz=im_sz, mask=mask)
    pc = make_pcnet(hip, sd, im_sz)
    clf = hip['clf'].Classifier(body, DEV, state_dict=csd, input_sz=insz)
    scene = syn.scenes(seed + 1 if scene_seed is None else scene_seed, n_scenes, im_sz)
    setup = dict(classifier_crop_sz=crop, prj_brightness=0.5, prj_im_sz=prj_sz)
    B = len(targets)
    tr = []
    if n_scenes == 1:
        so.spaa(sd, so.OracleClassifier(body, csd, input_sz=insz), targets, targeted, scene, d_thr, stealth, setup, iters=1,
                trace=tr)
    else:
        # the reference takes ONE scene per call (Q9): the oracle of S scenes x K targets is S calls, stacked (the loss
        # scales 1/K against 1/(S K) are powers of two and drop out of the normalised step)
        per, parts = B // n_scenes, []
        assert per * n_scenes == B
        for i in range(n_scenes):
            tri = []
            so.spaa(sd, so.OracleClassifier(body, csd, input_sz=insz), targets[i * per:(i + 1) * per], targeted, scene[i:i + 1],
                    d_thr, stealth, setup, iters=1, trace=tri)
            parts.append(tri[0])
        tr = [{k: np.concatenate([np.asarray(q[k]) for q in parts]) for k in ('prj_adv', 'cam_infer', 'top1', 'target_logit')}]
        scene = scene.repeat_interleave(per, dim=0)
    ref = torch.from_numpy(tr[0]['prj_adv'])
    if golden_it0 is not None:
        # the reference's own first iteration (fixture produced by the unmodified reference in the build container); the
        # oracle run on THIS host's CPU may differ from it by rounding and, rarely, by a gate of its own
        d = float(np.abs(tr[0]['prj_adv'][:golden_it0.shape[0]] - golden_it0).max())
        print(f'oracle on this host vs reference fixture, first iteration: max abs diff {d:.2e}')
        assert d < 5e-3
    st = A.AttackState(pc, clf, targets, scene, stealth, setup, DEV, storage=storage)
    if getattr(st.clf.body, 'fuse_pool', False):
        # (this decomposition reads every convolution's own activation; VGG-16's fp16 pool fusion does not write the ones in front of a pool --
        # fused and separate launches are bitwise equal: test_vgg16_fp16_pool_fusion_is_bitwise)
        st.clf.body.fuse_pool = False
    x0 = torch.full((B, 3, *prj_sz), 0.5)
    acts, cacts = _oracle_activations(sd, csd, x0, scene.expand(B, -1, -1, -1) if n_scenes == 1 else scene, im_sz, crop, insz, body)
    errs = {}
    # fp32: tests/gates.py's tolerances, 1e-5 on the camera image, 1e-4 on the produced projector image.  fp16 storage: `tol` =
    # dict(near_zero, value_tol, cam, logit, image, measured) stated by the caller as multiples of what it measured
    t = tol or dict(near_zero=gates.NEAR_ZERO, value_tol=gates.VALUE_TOL, cam=1e-5, logit=1e-4, image=1e-4, measured=None)
    for mode in ('plain', 'oracle_gates'):
        st.x.copy_(M.to_nhwc4(x0.to(DEV)))
        st.stats[:, 5] = 1e6
        st.forward_decide(targeted, d_thr, 0.9)
        pairs = gates.pcnet_pairs(st.eng, acts) + dict(vgg16=gates.vgg16_pairs, resnet18=gates.resnet18_pairs,
                                                       inception_v3=gates.inception_pairs)[body](st.clf.body, cacts)
        if mode == 'plain':
            flips, per_layer = gates.count_flips(pairs, t['near_zero'], t['value_tol'], t['measured'])
        else:
            gates.inject(pairs, (st.eng, st.clf.body))
        if mode == 'plain':
            e_cam = rel_inf(M.to_nchw(st.eng.a['Y']), torch.from_numpy(tr[0]['cam_infer']))
            if t['measured'] is not None:
                t['measured']['cam'] = max(t['measured'].get('cam', 0.0), e_cam)
                t['measured']['logit'] = max(t['measured'].get('logit', 0.0), float(np.abs(st.stats[:, 6].cpu().numpy() - tr[0]['target_logit']).max()))
            assert e_cam < t['cam']
            assert (st.state[:, 3].cpu().numpy() == tr[0]['top1']).all()
            assert np.allclose(st.stats[:, 6].cpu().numpy(), tr[0]['target_logit'], rtol=t['logit'], atol=t['logit'])
        st.backward_step(2, 1)
        xn = M.to_nchw(st.x).cpu()
        errs[mode] = torch.tensor([rel_inf(xn[b], ref[b]) for b in range(B)])
    print(f'{body} loop, first iteration at cam {im_sz} prj {prj_sz}: gates differing per sample {flips.tolist()} {per_layer}; projector image '
          f'rel Linf plain {errs["plain"].tolist()}, with the oracle\'s gates {errs["oracle_gates"].tolist()}')
    assert (errs['plain'][flips == 0] < t['image']).all() and (errs['oracle_gates'] < t['image']).all()
    st.flips, st.errs = flips, errs
    return st


@pytest.mark.parametrize('targeted', [False, True])
def test_reference_call_shapes_first_iteration(hip, targeted):
    """The two calls `run_projector_based_attack` makes (projector_based_attack.py:107,120 with main.py:19-33's setup):
    projector 256 x 256 -> camera 240 x 320, classifier crop 240 x 240 -> 224 x 224, B = 1 untargeted on the scene's own label
    and B = 10 targeted on the ten ImageNet-10 ids.  First iteration against the oracle, gate-aware."""
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    cam, prj = (240, 320), (256, 256)
    if targeted:
        targets = list(syn.IMAGENET10_TARGETS)
    else:
        _, _, idx = so.OracleClassifier('resnet18', csd)(syn.scenes(22, 1, cam), (240, 240))
        targets = [int(idx[0, 0])]
    st = _first_iteration_gate_aware(hip, 'resnet18', csd, (224, 224), cam, (240, 240), targets, 21, prj_sz=prj,
                                     targeted=targeted)
    assert st.B == len(targets) and st.eng.Hc == 240 and st.eng.Wc == 320 and st.eng.Hp == 256


def test_reference_call_shapes_graph_replay(hip):
    """spaa() at the reference's call shapes replays the iteration as a captured HIP graph (few pixels: the host cannot enqueue
    ~120 launches as fast as the GPU runs them).  The replayed attack must be BITWISE the eagerly launched one."""
    A = hip['attack']
    cam, prj = (240, 320), (256, 256)
    sd = syn.pcnet_state_dict(21, cam_sz=cam, mask='ones')
    pc = make_pcnet(hip, sd, cam)
    clf = hip['clf'].Classifier('resnet18', DEV, state_dict=syn.resnet18_state_dict(2, logit_gain=20.0))
    scene = syn.scenes(22, 1, cam)
    setup = dict(classifier_crop_sz=(240, 240), prj_brightness=0.5, prj_im_sz=prj)
    targets = list(syn.IMAGENET10_TARGETS)
    assert len(targets) * cam[0] * cam[1] <= A.GRAPH_MAX_PIXELS
    cam_g, prj_g = A.spaa(pc, clf, None, targets, True, scene, 5, 'camdE_caml2', DEV, setup, iters=8)
    assert A.LAST_RUN == dict(iterations=8, graph=True)   # 1 eager + 7 replays (the capture executes nothing)
    tr = []   # (a trace switches the graph off: eager launches)
    cam_e, prj_e = A.spaa(pc, clf, None, targets, True, scene, 5, 'camdE_caml2', DEV, setup, iters=8, trace=tr)
    assert len(tr) == 8 and torch.equal(cam_g, cam_e) and torch.equal(prj_g, prj_e)
    assert prj_g.shape == (10, 3, 256, 256) and cam_g.shape == (10, 3, 240, 320)
    # the same with the graph switched off by its size limit (the path a refused capture falls back to)
    old = A.GRAPH_MAX_PIXELS
    try:
        A.GRAPH_MAX_PIXELS = 0
        cam_0, prj_0 = A.spaa(pc, clf, None, targets, True, scene, 5, 'camdE_caml2', DEV, setup, iters=8)
    finally:
        A.GRAPH_MAX_PIXELS = old
    assert A.LAST_RUN == dict(iterations=8, graph=False) and torch.equal(cam_0, cam_g) and torch.equal(prj_0, prj_g)


@pytest.mark.parametrize('cam,prj,b', [((64, 64), (64, 64), 5), ((240, 320), (256, 256), 3), ((48, 80), (64, 64), 9), ((256, 256), (256, 256), 6)])
def test_warp_backward_tiled_equals_gather(hip, cam, prj, b):
    """csrc/warp.hip: the LDS-staged adjoint of grid_sample (16 x 16 projector tiles, the camera-side bounding box of their tap
    lists read once per image) sums every projector pixel's list in the same order as the untiled gather: bitwise equal, also
    for projector != camera sizes, batch sizes that are not multiples of 4 and a masked camera image."""
    M = hip['models']
    sd = syn.pcnet_state_dict(3, cam_sz=cam, mask='rect')
    pc = make_pcnet(hip, sd, cam)
    eng = M.PCNetEngine(pc, b, prj)
    assert eng.tiled is not None
    torch.manual_seed(b)
    x = torch.rand(b, prj[0], prj[1], 4, device=DEV) * 1.4 - 0.2       # (values outside [0, 1]: the clamp gate)
    x[..., 3] = 0
    eng._x, eng._clamp = x, 1
    g = torch.randn(b, cam[0], cam[1], 4, device=DEV)
    g[..., 3] = 0
    a = eng.warp_backward(g).clone()
    tiled, eng.tiled = eng.tiled, None
    ref = eng.warp_backward(g).clone()
    eng.tiled = tiled
    assert torch.equal(a, ref)
    # and against autograd through F.grid_sample on the CPU
    xc = x[..., :3].permute(0, 3, 1, 2).cpu().clone().requires_grad_(True)
    grid = eng.grid[..., :2].cpu()[None].expand(b, -1, -1, -1)
    y = F.grid_sample(xc.clamp(0, 1), grid, mode='bilinear', padding_mode='zeros', align_corners=True) * sd['mask']
    (y * g[..., :3].permute(0, 3, 1, 2).cpu()).sum().backward()
    assert rel_inf(a[..., :3].permute(0, 3, 1, 2), xc.grad) < 5e-5   # (border pixels sum hundreds of taps: another order on the CPU)


@pytest.mark.parametrize('cam,prj,b', [((64, 64), (64, 64), 5), ((240, 320), (256, 256), 3), ((44, 68), (56, 40), 9), ((256, 256), (256, 256), 6), ((48, 40), (45, 35), 4)])
def test_warp_forward_from_tap_table_and_fused_sumsq(hip, cam, prj, b):
    """Round 6, csrc/warp.hip.  (a) spaa_warp_fwd_taps -- grid_sample from the per-attack tap table (32 x 8 camera tiles, four images per
    workgroup) -- against the grid kernel (same taps; weights x mask folded: equal to rounding), against F.grid_sample on the CPU, and as
    the exact adjoint of the tiled backward pass (<fwd(x), g> == <x, bwd(g)> to fp32 summation noise).  (b) spaa_warp_bwd_tiled_sumsq: the
    gradient is bitwise that of spaa_warp_bwd_tiled (+ spaa_grad_sumsq's prjl2 term for colour-step samples), and its per-tile partial sums
    add up to spaa_grad_sumsq's ||g_b||^2; spaa_step_and_track_n on them moves x like spaa_step_and_track on the block sums."""
    M, lib = hip['models'], hip['lib']
    sd = syn.pcnet_state_dict(3, cam_sz=cam, mask='rect')
    pc = make_pcnet(hip, sd, cam)
    eng = M.PCNetEngine(pc, b, prj)
    assert eng.tiled is not None and eng.sumsq_tiles() == ((prj[1] + 15) // 16) * ((prj[0] + 15) // 16)
    torch.manual_seed(b + cam[0])
    x = torch.rand(b, prj[0], prj[1], 4, device=DEV) * 1.4 - 0.2       # (values outside [0, 1]: the clamp and its gate)
    x[..., 3] = 0
    scene = torch.rand(b, cam[0], cam[1], 4, device=DEV)
    scene[..., 3] = 0
    eng.set_scene(scene)
    old = M.TAP_TABLE_FWD
    try:
        M.TAP_TABLE_FWD = True
        xw_t = eng.warp(x, True).clone()
        M.TAP_TABLE_FWD = False
        xw_g = eng.warp(x, True).clone()
    finally:
        M.TAP_TABLE_FWD = old
    assert rel_inf(xw_t, xw_g) < 2e-6 and float(xw_t[..., 3].abs().max()) == 0.0
    xc = x[..., :3].permute(0, 3, 1, 2).cpu()
    grid = eng.grid[..., :2].cpu()[None].expand(b, -1, -1, -1)
    y = F.grid_sample(xc.clamp(0, 1), grid, mode='bilinear', padding_mode='zeros', align_corners=True) * sd['mask']
    assert rel_inf(xw_t[..., :3].permute(0, 3, 1, 2), y) < 2e-5
    # adjoint pair (no clamp: a linear map)
    g = torch.randn(b, cam[0], cam[1], 4, device=DEV)
    g[..., 3] = 0
    M.TAP_TABLE_FWD, keep = True, M.TAP_TABLE_FWD
    try:
        fx = eng.warp(x, False).clone()
    finally:
        M.TAP_TABLE_FWD = keep
    eng._x, eng._clamp = x, 0
    bg = eng.warp_backward(g).clone()
    lhs, rhs = (fx.double() * g.double()).sum(), (x.double() * bg.double()).sum()
    assert abs(float(lhs - rhs)) < 1e-5 * max(1.0, abs(float(lhs)))
    # (b) fused sum of squares, with and without the prjl2 term
    eng._x, eng._clamp = x, 1
    state = torch.zeros(b, 4, dtype=torch.int32, device=DEV)
    state[::2, 1] = 1                                             # every other sample takes the colour step
    state[1::3, 0] = 1
    for scale in (0.0, 0.37):
        ref = eng.warp_backward(g).clone()
        nblk = (prj[0] * prj[1] + 255) // 256
        part_ref = torch.zeros(b, nblk, device=DEV)
        lib.call('spaa_grad_sumsq', lib.ptr(ref), lib.ptr(x), 0.5, scale, lib.ptr(state), lib.ptr(part_ref), b, prj[0] * prj[1])
        part = torch.zeros(b, eng.sumsq_tiles(), device=DEV)
        got = eng.warp_backward(g, sumsq=(part, 0.5, scale, state)).clone()
        # (ref was updated in place by spaa_grad_sumsq.  Without the prjl2 term: the same bits; with it: the same formula, whose
        # multiply-adds the compiler contracts per kernel)
        assert torch.equal(got, ref) if scale == 0.0 else rel_inf(got, ref) < 1e-6, scale
        assert rel_inf(part.double().sum(1), part_ref.double().sum(1)) < 1e-6, scale
        part2 = torch.zeros_like(part)
        eng.warp_backward(g, sumsq=(part2, 0.5, scale, state))
        assert torch.equal(part, part2)                               # fixed order: bitwise run to run
        xa, xb_ = x.clone(), x.clone()
        best_a, best_b = torch.zeros_like(x), torch.zeros_like(x)
        cam_im, cb_a, cb_b = torch.rand(b, cam[0], cam[1], 4, device=DEV), torch.zeros(b, cam[0], cam[1], 4, device=DEV), torch.zeros(b, cam[0], cam[1], 4, device=DEV)
        lib.call('spaa_step_and_track', lib.ptr(xa), lib.ptr(ref), lib.ptr(part_ref), lib.ptr(state), 2.0, 1.0, lib.ptr(best_a), lib.ptr(cam_im),
                 lib.ptr(cb_a), b, prj[0] * prj[1], cam[0] * cam[1])
        bits = torch.full((b, prj[0] * prj[1]), 255, dtype=torch.uint8, device=DEV)
        lib.call('spaa_step_and_track_n', lib.ptr(xb_), lib.ptr(got), lib.ptr(part), part.shape[1], lib.ptr(state), 2.0, 1.0, lib.ptr(best_b),
                 lib.ptr(cam_im), lib.ptr(cb_b), b, prj[0] * prj[1], cam[0] * cam[1], lib.ptr(bits))
        assert rel_inf(xb_, xa) < 1e-6 and rel_inf(best_b, best_a) < 1e-6 and torch.equal(cb_a, cb_b)
        # the step's clamp-gate bytes describe the x it wrote; the adjoint gated by them = the adjoint gated by that x, bit for bit
        ok = (xb_[..., :3] >= 0) & (xb_[..., :3] <= 1)
        want_bits = (ok[..., 0].to(torch.uint8) | (ok[..., 1].to(torch.uint8) << 1) | (ok[..., 2].to(torch.uint8) << 2)).reshape(b, -1)
        assert torch.equal(bits, want_bits) and 0 < int((bits != 7).sum()) < bits.numel()
        eng._x = xb_
        pa, pb = torch.zeros_like(part), torch.zeros_like(part)
        ga = eng.warp_backward(g, sumsq=(pa, 0.5, scale, state)).clone()
        gb = eng.warp_backward(g, sumsq=(pb, 0.5, scale, state), clamp_bits=bits).clone()
        assert torch.equal(ga, gb) and torch.equal(pa, pb)
        eng._x = x


def test_vgg16_attack_loop_first_iteration(hip):
    """configs[4]'s classifier inside the SPAA loop (not just as a bare classifier)."""
    csd = syn.vgg16_state_dict(3, logit_gain=5.0, fc_width=512)
    _first_iteration_gate_aware(hip, 'vgg16', csd, (48, 48), (64, 64), (60, 60), [204, 291, 7], 11)


def test_inception_v3_attack_loop_first_iteration(hip):
    """configs[2]'s classifier (299-style up-sampling preprocessing, scaled down) inside the SPAA loop."""
    csd = syn.inception_v3_state_dict(4, logit_gain=20.0)
    _first_iteration_gate_aware(hip, 'inception_v3', csd, (107, 107), (128, 128), (120, 120), [204, 291], 12)


def test_benchmarked_configuration_first_iteration(hip):
    """ONE oracle-compared iteration of exactly what bench.py times (BASELINE.json configs[1]: B = 64 = 8 scenes x 8 targets,
    256 x 256, ResNet-18, `camdE_caml2`, bench.build_attack's seeds): the batch-64 tile / split-K / Winograd / canvas selection is
    checked against the oracle directly, gate-aware -- 1e-4 on every sample whose gates agree with the oracle's and on all 64
    samples with the oracle's gates (/root/reference/src/python/projector_based_attack.py:264-328)."""
    from spaa_amd import convplan
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    targets = (list(syn.IMAGENET10_TARGETS) * 8)[:8] * 8
    convplan.PROFILE = []      # (records which kernel every launch took: HIP events, no effect on the results)
    try:
        st = _first_iteration_gate_aware(hip, 'resnet18', csd, (224, 224), (256, 256), (240, 240), targets, 0, scene_seed=1,
                                         n_scenes=8)
        tiles = sorted({(name, tid) for name, _k, _f, _e0, _e1, tid, _b in convplan.PROFILE})
    finally:
        convplan.PROFILE = None
    # (Gates that differ from the oracle's sit within rounding of zero: of the ~1.6e8 ReLU / clamp gates per sample a handful at most, and a
    # good part of the samples needs no exchange at all.  The count moves with the summation order of any layer -- 40 of 64 samples without
    # a flip up to round 5, 24 with conv2 / conv2_s on csrc/s2f_x6.hip, whose nine taps accumulate in another order: 1-2 gates per sample.)
    print('samples by number of gates exchanged with the oracle:', torch.bincount(st.flips.flatten().long().cpu()).tolist())
    assert st.B == 64 and int((st.flips == 0).sum()) >= 16 and int(st.flips.max()) <= 4
    kinds = {tid % 100 for _n, tid in tiles}
    print('kernels of the benchmarked configuration:', sorted(kinds), 'split / canvas launches:', sorted({(n, t) for n, t in tiles if t >= 100})[:12])
    assert {70, 71} & kinds and any(t >= 100 and t % 100 in (70, 71) for _n, t in tiles)   # Winograd incl. its K-range (canvas) form


@pytest.mark.parametrize('n_scenes', [4, 12])
def test_untuned_batch_first_iteration(hip, n_scenes):
    """Batch sizes the tune table was never measured at (B = 32 and B = 96 at 256 x 256: every layer's kernel is BORROWED from the
    nearest measured pixel count, spaa_amd/convplan.py tuned_tile, or rule-chosen): one oracle-compared iteration, gate-aware, as
    for the benchmarked batch -- a caller at such a batch runs on selections no other test exercises."""
    from spaa_amd import convplan
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    targets = (list(syn.IMAGENET10_TARGETS) * 8)[:8] * n_scenes
    before = set(convplan.tune_report()['borrowed'])
    st = _first_iteration_gate_aware(hip, 'resnet18', csd, (224, 224), (256, 256), (240, 240), targets, 0, scene_seed=2, n_scenes=n_scenes)
    rep = convplan.tune_report()
    borrowed = set(rep['borrowed']) - before
    print(f'B = {8 * n_scenes}: {len(borrowed)} layer shapes borrowed their kernel from another pixel count, {len(rep["untuned"])} rule-chosen in this process')
    assert st.B == 8 * n_scenes and len(borrowed) >= 10


def test_inception_v3_full_input_size_first_iteration(hip):
    """configs[2]'s classifier at its real input size (299 x 299 from the 240 x 240 crop of a 256 x 256 camera image), B = 8:
    first iteration against the oracle, gate-aware."""
    csd = syn.inception_v3_state_dict(2, logit_gain=20.0)
    targets = list(syn.IMAGENET10_TARGETS)[:8]
    st = _first_iteration_gate_aware(hip, 'inception_v3', csd, (299, 299), (256, 256), (240, 240), targets, 0, scene_seed=1)
    assert st.B == 8


def _hip_gates(st, rows):
    """Every gate of one HIP attack state for the samples `rows` (a slice), as a list of (name, tensor): the engine's byte masks,
    the clamp gate of the output, the classifier body's ReLU signs and arg-max bytes (VGG-16 / Inception-v3 op lists)."""
    out = [(f'pcnet.{k}', st.eng.m[k][rows]) for k in sorted(st.eng.m)]
    yp = st.eng.a['Ypre'][rows]
    out.append(('pcnet.ypre', (yp > 0) & (yp <= 1)))
    body = st.clf.body
    for n, op in enumerate(getattr(body, 'ops', [])):
        if op['kind'] == 'conv':
            t = op['out']
            buf = t.buf[..., t.coff:t.coff + t.c] if hasattr(t, 'buf') else t
            out.append((f'clf.conv{n}', buf[rows] > 0))
        elif 'arg' in op:
            out.append((f'clf.pool{n}', op['arg'][rows]))
    for nm in ('h1', 'h2'):
        if hasattr(body, nm):
            out.append((f'clf.{nm}', getattr(body, nm)[rows] > 0))
    return out


@pytest.mark.parametrize('body', ['inception_v3', 'vgg16'])
def test_full_size_properties_other_classifiers(hip, body):
    """BASELINE.json configs[2] (Inception-v3, 299x299) and the SPAA loop with configs[4]'s VGG-16 at full size:
    batch 64, 256x256, size-independent properties of the HIP path."""
    A = hip['attack']
    sz = (256, 256)
    sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='ones')
    pc = make_pcnet(hip, sd, sz)
    csd = {'vgg16': syn.vgg16_state_dict, 'inception_v3': syn.inception_v3_state_dict}[body](2, logit_gain=20.0)
    clf = hip['clf'].Classifier(body, DEV, state_dict=csd)
    scenes = syn.scenes(11, 8, sz).repeat_interleave(8, dim=0)
    targets = (syn.IMAGENET10_TARGETS[:8]) * 8
    setup = dict(classifier_crop_sz=(240, 240), prj_brightness=0.5, prj_im_sz=sz)
    st = A.AttackState(pc, clf, targets, scenes, 'camdE_caml2', setup, DEV)
    st.iteration(True, 5, 2, 1, 0.9)
    x1, y1, logit1 = st.x.clone(), st.eng.a['Y'].clone(), st.stats[:, 6].clone()
    step = (x1[..., :3] - 0.5).flatten(1).norm(dim=1).cpu()
    lr = torch.where(st.state[:, 1].cpu() != 0, torch.tensor(1.0), torch.tensor(2.0))
    assert torch.allclose(step, lr, rtol=1e-4)                       # prescribed step length per sample
    st_gates = st       # (its buffers hold the first iteration's activations until it is stepped again below)
    rows8 = list(range(3, 64, 8))          # one sample of each of the eight scenes (the eight targets of a scene share its PCNet gates)
    st8 = A.AttackState(pc, clf, [targets[i] for i in rows8], scenes[rows8], 'camdE_caml2', setup, DEV)
    st8.iteration(True, 5, 2, 1, 0.9)
    assert rel_inf(st8.eng.a['Y'], y1[rows8]) < 1e-6                 # samples are independent
    assert torch.allclose(st8.stats[:, 6], logit1[rows8], rtol=1e-4, atol=1e-4)
    # (a batch of 8 takes other tiles / split-K factors than a batch of 64: another summation order, hence possibly another
    # side for a ReLU gate within rounding of zero -- sparse differences)
    # ... so a sample is compared at 1e-4 where ALL its gates (PCNet byte masks, clamp gate, the body's ReLU signs and arg-max bytes)
    # are the same in the two runs, and within the measured effect of a flipped gate otherwise (8.3e-3: profiles/r02_parity.txt)
    g64 = _hip_gates(st_gates, rows8)
    g8 = _hip_gates(st8, slice(0, 8))
    assert [n for n, _ in g64] == [n for n, _ in g8] and len(g8) > 20
    differing = torch.zeros(8, dtype=torch.long)
    for (_, a), (_, b_) in zip(g64, g8):
        differing += (a != b_).flatten(1).sum(dim=1).cpu()
    e8 = torch.tensor([rel_inf(st8.x[i], x1[rows8[i]]) for i in range(8)])
    print(f'{body}: sub-batch of 8 (one sample per scene) against its rows of the batch of 64: differing gates per sample {differing.tolist()}, '
          f'projector image rel Linf {[f"{v:.1e}" for v in e8.tolist()]}')
    assert (e8[differing == 0] < 1e-4).all() and (e8 < 2.5e-2).all() and int(differing.max()) < 64
    st2 = A.AttackState(pc, clf, targets, scenes, 'camdE_caml2', setup, DEV)
    st2.iteration(True, 5, 2, 1, 0.9)
    assert torch.equal(st2.x, x1)                                    # bitwise reproducible
    for _ in range(2):
        st.iteration(True, 5, 2, 1, 0.9)
    cam, prj = st.results()
    assert torch.isfinite(cam).all() and torch.isfinite(prj).all() and prj.min() >= 0 and prj.max() <= 1


def test_perc_al_with_vgg16_at_full_size(hip):
    """configs[4] (fp32 part): PerC_AL.adversary_projector with VGG-16 — iteration 0 from identical state vs the oracle,
    gate-aware (VGG-16 has no skip connections: one ReLU gate or max-pool arg-max within rounding of a tie changes the
    gradient over the whole image), at 256x256 and at 64x64, and output properties over more iterations.  At 256x256 every
    sample usually has a few such units (r02: five of 14.7 M per sample), so the plain (no gates exchanged) 1e-4 assertion is
    carried by the smaller size: the test FAILS if no tested sample anywhere is flip-free."""
    import gates
    from spaa_amd.perc_al import PerC_AL, PerCALState
    n_clean = 0
    for h, crop, insz, fcw, b in [(64, (60, 60), (48, 48), 256, 8), (256, (240, 240), (224, 224), 256, 4)]:
        csd = syn.vgg16_state_dict(3, logit_gain=5.0, fc_width=fcw)
        clf = hip['clf'].Classifier('vgg16', DEV, state_dict=csd, input_sz=insz)
        oclf = so.OracleClassifier('vgg16', csd, input_sz=insz)
        # (iteration 0 starts from delta = 0: copies of ONE scene would have identical gates, so the small size uses b scenes)
        scene = syn.scenes(1, b, (h, h)) if h < 256 else syn.scenes(1, 1, (h, h)).expand(b, -1, -1, -1).contiguous()
        _, _, idx = oclf(scene[:1], crop)
        labels = torch.tensor([int(i) for i in idx[0, 1:1 + b]])
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        otr = []
        so.perc_al_adversary_projector(oclf, scene, labels, 2.0, True, crop, 400, 1., 0.5, 0, stop_after=1, trace=otr)
        with torch.no_grad():
            _, cacts = so.vgg16_forward(csd, so.classifier_preprocess(scene, crop, insz), return_all=True)
        att = PerC_AL(device=DEV, max_iterations=400, alpha_l_init=1, alpha_c_init=0.5, confidence=0)
        res = {}
        for mode in ('plain', 'oracle_gates'):
            with torch.cuda.device(DEV):
                st = PerCALState(att, clf, scene, labels, 2.0, True, crop)
            info = {}

            def hook(eng, mode=mode, info=info):
                pairs = gates.vgg16_pairs(eng.body, cacts)
                if mode == 'plain':
                    info['flips'], info['layers'] = gates.count_flips(pairs)
                else:
                    gates.inject(pairs, (eng.body,))

            st.iteration(0, after_forward=hook)
            d0 = hip['models'].to_nchw(st.delta).cpu()
            res[mode] = torch.tensor([rel_inf(d0[k], otr[0]['delta'][k]) for k in range(b)])
            if mode == 'plain':
                flips, layers = info['flips'], info['layers']
                # (the colour distance is that of the image AFTER the iteration's adversarial step: 1e-4 where the gates agree)
                cd, cdo = st.stats[:, 3].cpu().numpy(), otr[0]['color_dis'].numpy()
                cl = (flips == 0).numpy()
                assert np.allclose(cd[cl], cdo[cl], rtol=1e-4) and np.allclose(cd, cdo, rtol=2e-3), (cd, cdo, flips)
                assert (st.state[:, 3].cpu().numpy() == otr[0]['top1']).all()
        print(f'PerC-AL + VGG-16 at {h}x{h}, iteration 0: gates differing per sample {flips.tolist()} {layers}; delta rel Linf plain '
              f'{res["plain"].tolist()}, with the oracle\'s gates {res["oracle_gates"].tolist()}')
        n_clean += int((flips == 0).sum())
        assert (res['plain'][flips == 0] < 1e-4).all() and (res['oracle_gates'] < 1e-4).all()
    assert n_clean > 0, 'no flip-free sample at any tested size: the plain 1e-4 assertion never ran'
    out = PerC_AL(device=DEV, max_iterations=6, alpha_l_init=1, alpha_c_init=0.5, confidence=0) \
        .adversary_projector(clf, scene, labels, None, 2.0, True, crop)
    assert out.shape == scene.shape and out.min() >= 0 and out.max() <= 1
    assert (torch.round(out * 255) / 255 - out).abs().max() < 1e-6


@pytest.mark.parametrize('tile', [0, 16, 18, 22, 25, 27, 34, 36, 39, 40, 42, 44, 45, 48, 49, 50, 234, 948])
def test_gate_byte_masks(hip, tile):
    """ReLU gates as byte masks (1 byte per 4 channels, include/spaa_hip.h: mask_out / gate_bits / gate2_bits): the mask a
    forward launch writes equals (out > 0), and an input-gradient launch gated by masks is BITWISE equal to the same launch
    gated by the fp32 activations; conv, strided-class and folded transposed conv."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(tile)
    try:
        for ci, co, k, s, h, w, b in [(64, 96, 3, 1, 19, 23, 2), (32, 64, 3, 2, 22, 18, 3), (128, 32, 1, 1, 17, 9, 2)]:
            x = torch.randn(b, ci, h, w)
            wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
            bias = torch.randn(co)
            plan = cp.conv_fwd_plan(wt, bias, s, k // 2, DEV)
            ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
            out = torch.zeros(b, ho, wo, co, device=DEV)
            mask = torch.full((b, ho, wo, co // 4), 255, dtype=torch.uint8, device=DEV)
            cp.FORCE_TILE = tile
            plan.run(nhwc(x).to(DEV), out, act=lib.ACT_RELU, mask_out=mask)
            cp.FORCE_TILE = 0
            assert rel_inf(nchw(out.cpu(), co), F.relu(F.conv2d(x, wt, bias, s, k // 2))) < 2e-5
            assert torch.equal(mask, lib.pack_gate_mask(out)), (tile, ci, co)
            # dgrad of a following layer whose output has `co` channels... use the transposed role: gradient w.r.t. this
            # layer's OUTPUT shape is what a later dgrad produces; gate it by this activation
            wt2 = torch.randn(64, co, 3, 3) / (co * 9) ** 0.5
            dplan = cp.conv_dgrad_plan(wt2, 1, 1, DEV)
            gy = nhwc(torch.randn(b, 64, ho, wo)).to(DEV)
            act2 = torch.randn(b, ho, wo, co, device=DEV)
            g_f, g_m = torch.zeros(b, ho, wo, co, device=DEV), torch.zeros(b, ho, wo, co, device=DEV)
            a_f, a_m = torch.zeros_like(g_f), torch.zeros_like(g_f)
            cp.FORCE_TILE = tile
            dplan.run(gy, g_f, gate=out, aux_out=a_f, gate2=act2)
            dplan.run(gy, g_m, gate_bits=mask, aux_out=a_m, gate2_bits=lib.pack_gate_mask(act2))
            cp.FORCE_TILE = 0
            assert torch.equal(g_f, g_m) and torch.equal(a_f, a_m), (tile, ci, co)
        for kk, pad, op in [(3, 1, 1), (2, 0, 0)]:
            xt, wtt, bt = torch.randn(2, 64, 11, 13), torch.randn(64, 32, kk, kk) / 16, torch.randn(32)
            ref = F.relu(F.conv_transpose2d(xt, wtt, bt, 2, pad, op))
            tplan = cp.deconv_fwd_plan(wtt, bt, 2, pad, DEV)
            outt = torch.zeros(2, ref.shape[2], ref.shape[3], 32, device=DEV)
            maskt = torch.zeros(2, ref.shape[2], ref.shape[3], 8, dtype=torch.uint8, device=DEV)
            cp.FORCE_TILE = tile
            tplan.run(nhwc(xt).to(DEV), outt, act=lib.ACT_RELU, mask_out=maskt)
            cp.FORCE_TILE = 0
            assert rel_inf(nchw(outt.cpu()), ref) < 2e-5 and torch.equal(maskt, lib.pack_gate_mask(outt)), (tile, kk)
    finally:
        cp.FORCE_TILE = 0
    with pytest.raises(ValueError):   # masks are uint8
        plan.run(nhwc(x).to(DEV), out, mask_out=torch.zeros(b, ho, wo, co // 4, device=DEV))


def test_engine_masks_match_activations(hip, golden_dir):
    """After a forward pass the engines' byte masks are exactly the signs of their activation buffers (PCNet + ResNet-18),
    and the max-pool argmax bytes carry the 'maximum is positive' flag."""
    z = load(golden_dir, 'spaa_64_near')
    sd, pc, clf, oclf, scene, setup = _setup_case(hip, z)
    st = hip['attack'].AttackState(pc, clf, [int(t) for t in z['targets']], scene, 'camdE_caml2', setup, DEV)
    st.forward_decide(True, 5, 0.9)
    lib = hip['lib']
    for k, m in st.eng.m.items():
        want = lib.pack_gate_mask(st.eng.a[k])
        if k == 'X7' and st.eng.fuse_tail:
            # the fused tail wrote X7's gates; a['X7'] is recomputed on demand by the stand-alone launch (another summation
            # order): the two may disagree on units within rounding of zero, nowhere else
            bad = (m != want)
            assert bad.float().mean() < 1e-4
            if bad.any():
                assert st.eng.a[k].view(*m.shape, 4)[bad].abs().max() < 1e-5 * st.eng.a[k].abs().max()
        else:
            assert torch.equal(m, want), k
    body = st.clf.body
    for blk in body.blocks:
        assert torch.equal(blk['m_o1'], lib.pack_gate_mask(blk['o1'])) and torch.equal(blk['m_out'], lib.pack_gate_mask(blk['out']))
    assert torch.equal((body.mp_arg & 128) != 0, body.mp > 0)


@pytest.mark.parametrize('cam_sz,prj_sz,b', [((64, 64), (64, 64), 3), ((48, 80), (64, 64), 2), ((240, 320), (256, 256), 2)])
def test_fused_skipconv2_forward_and_backward(hip, cam_sz, prj_sz, b):
    """`transConv1(x5) + skipConv2(x1)` and `conv2^T(g2) + skipConv2^T(g6)` as single launches of the patch-staged stride-2 kernel
    (second source; spaa_amd/models.py PCNetEngine.fuse_skip2) against the engine with the three layers launched separately: the
    same network output, gate masks and input gradient up to the summation order (one accumulation chain instead of two
    rounded results added), and against the oracle's forward."""
    M = hip['models']
    sd = syn.pcnet_state_dict(5, cam_sz=cam_sz, mask='rect')
    pc = make_pcnet(hip, sd, cam_sz)
    old = M.FUSE_SKIP2_MIN_PIXELS
    try:
        M.FUSE_SKIP2_MIN_PIXELS = 0
        e1 = M.PCNetEngine(pc, b, prj_sz)
        e0 = M.PCNetEngine(pc, b, prj_sz, fuse_skip2=False)
    finally:
        M.FUSE_SKIP2_MIN_PIXELS = old
    assert e1.fuse_skip2 and e1.fuse_skip3 and not e0.fuse_skip2 and not e0.fuse_skip3 and e1.d['conv2_s'].fixed_tile == 74
    torch.manual_seed(b)
    scene = syn.scenes(3, b, cam_sz)
    x = torch.rand(b, 3, *prj_sz)
    g = torch.randn(b, *cam_sz, 4, device=DEV)
    g[..., 3] = 0
    outs = []
    for e in (e1, e0):
        e.set_scene(M.to_nhwc4(scene.to(DEV)))
        y = e.forward(M.to_nhwc4(x.to(DEV))).clone()
        gx = e.backward(g).clone()
        outs.append((y, gx, e.a['X6'].clone(), torch.cat([e.m[k].reshape(-1) for k in sorted(e.m)]), e.g['P1'].clone()))
    assert e1.f['transConv1x'].last_tile == 74 and e1.d['conv2x'].last_tile == 74 and e1.d['conv2_s'].last_tile == 74
    assert e1.f['conv5x'].wino.last_tile in (70, 71) and e1.d['conv3x'].wino.last_tile in (70, 71)
    (y1, gx1, x61, m61, p11), (y0, gx0, x60, m60, p10) = outs
    flips = int((m61 != m60).sum())
    print(f'fused vs separate at {cam_sz} B={b}: Y {rel_inf(y1, y0):.1e}, X6 {rel_inf(x61, x60):.1e}, gate bytes differing (all eleven masks) {flips} of {m60.numel()}, '
          f'P1 {rel_l2(p11, p10):.1e}, input gradient rel L2 {rel_l2(gx1, gx0):.1e}')
    assert rel_inf(y1, y0) < 2e-6 and rel_inf(x61, x60) < 2e-6 and flips <= 2e-5 * m60.numel()
    assert rel_l2(p11, p10) < (1e-5 if flips == 0 else 1e-2) and rel_l2(gx1, gx0) < (1e-5 if flips == 0 else 1e-2)
    with torch.no_grad():
        xw = so.warp(sd, x.clamp(0, 1), cam_sz) * sd['mask']
        ref = so.shading_net(sd, xw, (scene, xw * scene))
    assert rel_inf(M.to_nchw(y1), ref) < 1e-5


@pytest.mark.parametrize('cam_sz,prj_sz,b', [((64, 64), (64, 64), 3), ((48, 80), (64, 64), 2), ((240, 320), (256, 256), 2)])
def test_fused_skipconv2_fp16_storage(hip, cam_sz, prj_sz, b):
    """fp16 storage: `transConv1(x5) + skipConv2(x1)` and `conv2^T(g2) + skipConv2^T(g6)` as single launches of the patch-staged fp16
    kernel's folded form (csrc/tapconv_h16p.hip, second source) against the engine with the layers launched separately.  Both see
    the same fp16 operands; the separate launches round the 1 x 1 convolution's result (R2, t1) to fp16 in HBM where the fused
    launch keeps it in the fp32 accumulators: they differ by that rounding (2^-11 relative per element) and nothing else."""
    M = hip['models']
    sd = syn.pcnet_state_dict(5, cam_sz=cam_sz, mask='rect')
    pc = make_pcnet(hip, sd, cam_sz)
    old = M.FUSE_SKIP2_MIN_PIXELS
    try:
        M.FUSE_SKIP2_MIN_PIXELS = 0
        e1 = M.PCNetEngine(pc, b, prj_sz, 'f16')
        e0 = M.PCNetEngine(pc, b, prj_sz, 'f16', fuse_skip2=False)
    finally:
        M.FUSE_SKIP2_MIN_PIXELS = old
    assert e1.fuse_skip2 and not e0.fuse_skip2 and 'transConv1x' in e1.f and 'conv2x' in e1.d
    torch.manual_seed(b)
    scene = syn.scenes(3, b, cam_sz)
    x = torch.rand(b, 3, *prj_sz)
    g = torch.randn(b, *cam_sz, 4, device=DEV)
    g[..., 3] = 0
    outs = []
    for e in (e1, e0):
        e.set_scene(M.to_nhwc4(scene.to(DEV)))
        y = e.forward(M.to_nhwc4(x.to(DEV))).clone()
        x5, x1 = e.a['X5'].clone(), e.a['X1'].clone()
        gx = e.backward(g).clone()
        outs.append((y, gx, e.a['X6'].float().clone(), e.g['P1'].float().clone(), x5, x1, e.g['P2'].clone(), e.g['P6'].clone()))
    # (round 6: the fused pair runs on csrc/fs2_h16.hip -- per input pixel, the nine real (class, tap) products -- instead of the patch-staged
    # kernel's folded form; SPAA_FS2_H16=0 restores that)
    assert e1.fs2 is not None or (e1.f['transConv1x'].last_tile == 68 and e1.d['conv2x'].last_tile == 68)
    # (and `conv5(x4) + skipConv3(x2)` / `conv3^T + skipConv3^T` as two-source launches of the same kernel: K-concatenated channels)
    assert e1.fuse_skip3 and not e0.fuse_skip3 and e1.f['conv5x'].last_tile == 68 and e1.d['conv3x'].last_tile == 68
    (y1, gx1, x61, p11, x51, x11, p21, p61), (y0, gx0, x60, p10, x50, x10, p20, p60) = outs
    # (the engines differ in the conv1 pair as well -- `fuse_skip2=False` launches conv1_s / conv1 separately -- so X1 / X5 agree to fp16
    # rounding, not bitwise; the fused layer's own operands are what the fp64 check below uses)
    assert rel_inf(x51, x50) < 2e-2 and rel_inf(x11, x10) < 2e-3
    print(f'fp16 fused vs separate at {cam_sz} B={b}: X6 {rel_inf(x61, x60):.1e}, Y {rel_inf(y1, y0):.1e}, P1 rel L2 {rel_l2(p11, p10):.1e}, '
          f'input gradient rel L2 {rel_l2(gx1, gx0):.1e}')
    assert rel_inf(x61, x60) < 2e-3 and rel_inf(y1, y0) < 2e-2
    assert rel_l2(p11, p10) < 5e-2 and rel_l2(gx1, gx0) < 5e-2
    # the fused forward layer against fp64 arithmetic on the SAME fp16 operands and fp16-rounded weights
    sn = pc.shading_net
    wt, w2 = sn.transConv1.weight.detach().half().double().cpu(), sn.skipConv2.weight.detach().half().double().cpu()
    ref = F.conv_transpose2d(x51.double().cpu().permute(0, 3, 1, 2), wt, sn.transConv1.bias.detach().double().cpu(), 2, 1, 1) \
        + F.conv2d(x11.double().cpu().permute(0, 3, 1, 2), w2, sn.skipConv2.bias.detach().double().cpu())
    assert rel_inf(x61.cpu().permute(0, 3, 1, 2), F.relu(ref).float()) < 1.5e-3
    with torch.no_grad():
        xw = so.warp(sd, x.clamp(0, 1), cam_sz) * sd['mask']
        ref = so.shading_net(sd, xw, (scene, xw * scene))
    assert rel_inf(M.to_nchw(y1), ref) < 2e-2


@pytest.mark.parametrize('storage', ['f32', 'f16'])
@pytest.mark.parametrize('cam_sz,prj_sz,b', [((64, 64), (64, 64), 3), ((36, 52), (40, 40), 2), ((240, 320), (256, 256), 2)])
def test_fused_conv1_pair(hip, cam_sz, prj_sz, b, storage):
    """csrc/conv1pair.hip: `relu(conv1_s(cat[s, xw * s]))` and `relu(conv1(xw) + res1_s)` (models.py:284-285,295 of the reference) as
    one launch against the two smallcin launches over the warp kernel's 8-channel concatenation: the same S1 / X1 up to the
    summation order of the fp32 chains (fp16 storage: up to one fp16 rounding step of a stored value), the same gate bytes except
    for units within rounding of zero, the same network output and input gradient, and the oracle's forward.  Ragged tiles
    ((36, 52): 18 x 26 outputs per image) included."""
    M = hip['models']
    sd = syn.pcnet_state_dict(6, cam_sz=cam_sz, mask='rect')
    pc = make_pcnet(hip, sd, cam_sz)
    e1 = M.PCNetEngine(pc, b, prj_sz, storage)
    e0 = M.PCNetEngine(pc, b, prj_sz, storage)
    assert e1.pair1 is not None and (e1.pair1_bwd is not None) == (storage == 'f16')
    e0.pair1 = e0.pair1_bwd = None
    torch.manual_seed(b + cam_sz[0])
    scene = syn.scenes(3, b, cam_sz)
    x = torch.rand(b, 3, *prj_sz) * 1.2 - 0.1
    g = torch.randn(b, *cam_sz, 4, device=DEV)
    g[..., 3] = 0
    outs = []
    for e in (e1, e0):
        e.a['cat8'].fill_(float('nan'))     # (the fused engine must not depend on it)
        e.set_scene(M.to_nhwc4(scene.to(DEV)))
        y = e.forward(M.to_nhwc4(x.to(DEV))).clone()
        gx = e.backward(g).clone()
        outs.append((y, gx, e.a['S1'].float().clone(), e.a['X1'].float().clone(), e.m['S1'].clone(), e.m['X1'].clone(),
                     torch.cat([e.m[k].reshape(-1) for k in sorted(e.m)])))
    assert torch.isnan(e1.a['cat8']).all() and not torch.isnan(e0.a['cat8']).any()
    (y1, gx1, s11, x11, ms1, mx1, ma1), (y0, gx0, s10, x10, ms0, mx0, ma0) = outs
    tol = 2e-6 if storage == 'f32' else 1.1e-3
    flips = int((ms1 != ms0).sum()) + int((mx1 != mx0).sum())
    flips_all = int((ma1 != ma0).sum())      # (later layers' units within rounding of zero see the 1e-7 differences)
    print(f'conv1 pair fused vs separate at {cam_sz} B={b} {storage}: S1 {rel_inf(s11, s10):.1e}, X1 {rel_inf(x11, x10):.1e}, gate bytes '
          f'differing {flips} of {2 * ms0.numel()} (all eleven masks: {flips_all}), Y {rel_inf(y1, y0):.1e}, input gradient rel L2 {rel_l2(gx1, gx0):.1e}')
    if storage == 'f32':
        assert rel_inf(s11, s10) < tol and rel_inf(x11, x10) < tol and flips <= 2e-5 * 2 * ms0.numel()
    else:
        # fp16 storage, round 6: the fused kernel multiplies fp16 operands (xw, s and the fp32 product xw * s rounded in registers, fp16
        # weights) on v_mfma_f32_16x16x32_f16, like every other layer of the mode.  Its reference is float64 on the SAME rounded operands
        # (what is left: accumulation order + the fp16 rounding of the stored value); the separate launches (conv1_s over the fp32
        # concatenation: exact image operands) differ by the operands' rounding on top of that
        xw32 = e1.a['xw'][..., :3].permute(0, 3, 1, 2).cpu()
        sc32 = scene.clone()
        hx, hs, hxs = _h(xw32), _h(sc32), _h(xw32 * sc32)
        w1, ws = _h(sd['shading_net.conv1.weight']), _h(sd['shading_net.conv1_s.weight'])
        s_ref = F.relu(F.conv2d(torch.cat([hs, hxs], 1).double(), ws.double(), sd['shading_net.conv1_s.bias'].double(), 2, 1))
        s_ref_h = s_ref.float().half().float()
        x_ref = F.relu(F.conv2d(hx.double(), w1.double(), sd['shading_net.conv1.bias'].double(), 2, 1) + s_ref_h.double())
        es, ex = rel_inf(nchw(s11.cpu(), 32), s_ref), rel_inf(nchw(x11.cpu(), 32), x_ref)
        print(f'    fused fp16 pair vs float64 on the same fp16 operands: S1 {es:.1e}, X1 {ex:.1e}')
        assert es < 1.5e-3 and ex < 1.5e-3
        assert rel_inf(s11, s10) < 4e-3 and rel_inf(x11, x10) < 4e-3 and flips <= 2e-3 * 2 * ms0.numel()
        # the pair's ADJOINT as one launch (spaa_conv1_pair_bwd_f16) against the two thin-output launches on the SAME fp16 gradients:
        # the same fp16 products, fp32 accumulation in another order
        gxw_f = e1.g['xw'].clone()
        e1.d['conv1_s'].run(e1.g['S1'], e1.g['xs'], gate=e1.scene, gate_mode=hip['lib'].GATE_MUL)
        e1.d['conv1'].run(e1.g['P1'], e1.g['xw'], add=e1.g['xs'])
        print(f'    fused adjoint of the pair vs the two thin-output launches: g_xw {rel_inf(gxw_f, e1.g["xw"]):.1e}')
        assert rel_inf(gxw_f, e1.g['xw']) < 5e-6 and float(gxw_f[..., 3].abs().max()) == 0.0
    lib = hip['lib']
    assert torch.equal(ms1, lib.pack_gate_mask(e1.a['S1'])) and torch.equal(mx1, lib.pack_gate_mask(e1.a['X1']))
    assert rel_inf(y1, y0) < (5e-6 if storage == 'f32' else 2e-2)
    assert rel_l2(gx1, gx0) < ((1e-5 if flips_all == 0 else 1e-2) if storage == 'f32' else 5e-2)
    with torch.no_grad():
        xw = so.warp(sd, x.clamp(0, 1), cam_sz) * sd['mask']
        ref = so.shading_net(sd, xw, (scene, xw * scene))
    assert rel_inf(M.to_nchw(y1), ref) < (1e-5 if storage == 'f32' else 2e-2)


@pytest.mark.parametrize('storage', ['f32', 'f16'])
@pytest.mark.parametrize('cam_sz,b', [((64, 96), 2), ((72, 100), 3), ((256, 256), 1)])
def test_fused_shading_tail_and_head(hip, cam_sz, b, storage):
    """csrc/shading_tail.hip: transConv2 + conv6 forward and their input gradients as one kernel each (X7 and its gradient stay
    in LDS) against the separate launches on the same engine: outputs, gate bytes and the gradient handed to transConv1.
    fp32 storage: equal to rounding.  fp16 storage (the <_Float16> instantiations: X6 / P6 fp16 in HBM, read 8 / written 4
    halves per lane): both paths see the SAME fp16 X6 and the same gate bytes and round X7 and its gradient to fp16 (the separate
    launches in HBM, the fused kernels in LDS -- round 5: 32 KB tiles, two workgroups per compute unit) and conv6's weights to fp16
    (forward); what is left is the accumulation order (conv6's taps on v_dot2_f32_f16 against the MFMA of the separate launch)."""
    lib, m_ = hip['lib'], hip['models']
    torch.manual_seed(cam_sz[0] + b)
    sd = syn.pcnet_state_dict(4, cam_sz=cam_sz, mask='ones')
    pc = make_pcnet(hip, sd, cam_sz)
    eng = m_.PCNetEngine(pc, b, cam_sz, storage)
    assert eng.fuse_tail
    tol = 2e-6 if storage == 'f32' else 3e-3     # (fp16: 16 taps x 32 channels of values rounded at 4.9e-4 relative)
    x = torch.rand(b, cam_sz[0], cam_sz[1], 4, device=DEV)
    x[..., 3] = 0
    scene = torch.rand(b, cam_sz[0], cam_sz[1], 4, device=DEV)
    scene[..., 3] = 0
    eng.set_scene(scene)
    y_f = eng.forward(x).clone()
    ypre_f, m7_f = eng.a['Ypre'].clone(), eng.m['X7'].clone()
    # round 6: in the loop the tail writes the output's clamp gate as ONE byte per pixel and no pre-clamp tensor; a['Ypre'] above came
    # from the same kernel run once more (bitwise the values the byte was formed from): byte == (0 < Ypre <= 1) per channel, Y == clamp
    assert eng.gate_y is not None
    ok3 = (ypre_f[..., :3] > 0) & (ypre_f[..., :3] <= 1)
    assert torch.equal(eng.gate_y, ok3[..., 0].to(torch.uint8) | (ok3[..., 1].to(torch.uint8) << 1) | (ok3[..., 2].to(torch.uint8) << 2))
    assert torch.equal(y_f[..., :3], ypre_f[..., :3].clamp(max=1.0)) and torch.equal(eng.m['X7'], m7_f)
    x6 = eng.a['X6'].clone()
    gP = torch.randn(b, cam_sz[0], cam_sz[1], 4, device=DEV)
    gP[..., 3] = 0
    eng.backward(gP)
    p6_f = eng.g['P6'].clone()
    assert p6_f.dtype == (torch.float16 if storage == 'f16' else torch.float32) and torch.isfinite(p6_f.float()).all()
    eng.fuse_tail = False
    y_s = eng.forward(x).clone()
    assert torch.equal(eng.a['X6'], x6)   # the operand of both tails
    ypre_s, m7_s = eng.a['Ypre'].clone(), eng.m['X7'].clone()
    assert rel_inf(y_f, y_s) < tol and rel_inf(ypre_f, ypre_s) < tol
    bad = m7_f != m7_s
    assert bad.float().mean() < (1e-4 if storage == 'f32' else 2e-3)      # (units within rounding of zero may fall on either side)
    if bad.any():   # the channels whose BIT differs (a byte covers 4 channels) are within rounding of zero
        x7 = dict.__getitem__(eng.a, 'X7').float()
        bits = torch.tensor([1, 2, 4, 8], device=DEV, dtype=torch.uint8)
        diff = ((m7_f ^ m7_s).unsqueeze(-1) & bits) != 0
        assert x7.view(*m7_s.shape, 4)[diff].abs().max() < (1e-5 if storage == 'f32' else 2e-3) * x7.abs().max()
    eng.m['X7'].copy_(m7_f)               # the same gates for both backward passes
    eng.backward(gP)
    print(f'{storage} {cam_sz} b={b}: Y {rel_inf(y_f, y_s):.1e}  Ypre {rel_inf(ypre_f, ypre_s):.1e}  gate bytes differing {int(bad.sum())}  '
          f'P6 {rel_inf(p6_f, eng.g["P6"]):.1e}')
    assert rel_inf(p6_f, eng.g['P6']) < tol
    eng.fuse_tail = True


# ---------------------------------------------------------------------------------------------------------------
# fp16-STORAGE mode (BASELINE.json configs[4]: "fp16 with fp32 dE2000"): activations / gradients fp16 in HBM, fp16 weights,
# fp32 accumulation.  Tolerances: the reference for a layer is the fp32 (fp64-accumulated) result on the SAME fp16-rounded
# operands, so what is left is the accumulation order and the final fp16 rounding of the output (2^-11 = 4.9e-4 relative).
def _h(x):
    return x.half().float()


def test_h16p_canvas_and_k_ranges(hip):
    """The patch-staged fp16 kernel's canvas / K-range form (csrc/tapconv_h16p.hip CV: the 3 x 3 layers of the fp16-storage classifiers on
    small maps -- ResNet-18 layer3 / layer4 behind classifier.py:26-28, VGG-16's last block, perc_al/__init__.py:181-238) against float64 on
    the same fp16 operands: ragged canvases (batch not a multiple of the images per canvas), non-square maps, both N tiles, the plan's
    own and forced K ranges (partial sums in fp32, summed in fixed order: bitwise run to run), residual + ReLU + byte mask, the gated
    input-gradient form, fp16 and fp32 output, a channel count that is not a multiple of four (no split, the slow epilogue)."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(683)
    old = cp.H16P_CV
    try:
        for ci, co, h, w, b, cv in [(256, 256, 14, 14, 5, (0, 0, 1)), (512, 512, 7, 7, 9, (0, 0, 1)), (128, 192, 14, 14, 7, (128, 2, 1)),
                                    (128, 192, 14, 14, 7, (64, 4, 0)), (96, 64, 20, 33, 3, (0, 3, 1)), (64, 130, 7, 9, 4, (0, 0, 1)),
                                    (512, 512, 14, 14, 3, (128, 16, 1)), (256, 256, 14, 14, 64, (0, 0, 0))]:
            cp.H16P_CV = cv
            x = _h(torch.randn(b, ci, h, w))
            wt = _h(torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5)
            bias = torch.randn(co)
            y = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1).float()
            add = _h(torch.randn_like(y))
            plan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV)
            cs = (co + 3) // 4 * 4
            out = torch.zeros(b, h, w, cs, device=DEV, dtype=torch.float16)
            mask = torch.zeros(b, h, w, cs // 4, dtype=torch.uint8, device=DEV) if co % 4 == 0 else None
            xin = nhwc(x).half().to(DEV)
            plan.run(xin, out, add=nhwc(add, cs).half().to(DEV), act=lib.ACT_RELU, mask_out=mask)
            assert plan.last_tile == 68 and (plan.last_h16p_plan[2] == 1 or not cv[2]), (plan.last_tile, ci, co, h, w, plan.last_h16p_plan)
            if cv[1] > 1:
                assert plan.last_h16p_plan[1] == min(cv[1], ci // 32), plan.last_h16p_plan
            if cv[0]:
                assert plan.last_h16p_plan[0] == cv[0]
            want = F.relu(y + add)
            got = nchw(out.float().cpu(), co)
            assert rel_inf(got, want) < 1.5e-3, (ci, co, h, w, cv, rel_inf(got, want))
            if mask is not None:
                assert torch.equal(mask, lib.pack_gate_mask(out.float())), (ci, co)
            out_b = torch.zeros_like(out)
            plan.run(xin, out_b, add=nhwc(add, cs).half().to(DEV), act=lib.ACT_RELU)
            assert torch.equal(out_b, out)                      # (fixed summation order: bitwise run to run)
            if plan.last_h16p_plan[1] > 1 and co % 4 == 0:
                # round 6: the K ranges meeting inside the kernel (SPAA_SPLITK_FIXUP=1; last-arriving workgroup, fixed order) is bitwise the
                # separate second pass, and the arrival counters at the head of the workspace are left zero
                keep = cp.WINO_SPLITK_FIXUP
                try:
                    cp.WINO_SPLITK_FIXUP = True
                    out_fx, mask_fx = torch.zeros_like(out), torch.zeros_like(mask)
                    plan.run(xin, out_fx, add=nhwc(add, cs).half().to(DEV), act=lib.ACT_RELU, mask_out=mask_fx)
                    assert torch.equal(out_fx, out) and torch.equal(mask_fx, mask), (ci, co, cv)
                    assert int(plan._ws_fix[:cp.SPLITK_HDR].view(torch.int32).abs().max()) == 0
                finally:
                    cp.WINO_SPLITK_FIXUP = keep
            out32 = torch.zeros(b, h, w, cs, device=DEV)
            plan.run(xin, out32)
            assert plan.last_tile == 68 and rel_inf(nchw(out32.cpu(), co), y) < 2e-5, (ci, co, cv, rel_inf(nchw(out32.cpu(), co), y))
            if co % 4 == 0:
                # the input-gradient form with a byte-mask gate (the ReLU backward of the layer below)
                g = _h(torch.randn(b, co, h, w))
                dplan = cp.conv_dgrad_plan(wt, 1, 1, DEV)
                gate_src = torch.randn(b, h, w, ci, device=DEV)
                gbits = lib.pack_gate_mask(gate_src)
                gin = torch.zeros(b, h, w, ci, device=DEV, dtype=torch.float16)
                dplan.run(nhwc(g).half().to(DEV), gin, gate_bits=gbits)
                assert dplan.last_tile == 68 and (dplan.last_h16p_plan[2] == 1 or not cv[2])
                wantg = F.conv_transpose2d(g.double(), wt.double(), None, 1, 1).float() * (nchw(gate_src.cpu(), ci) > 0)
                assert rel_inf(nchw(gin.float().cpu(), ci), wantg) < 1.5e-3, (ci, co, cv)
    finally:
        cp.H16P_CV = old


@pytest.mark.parametrize('kind,ci,co,ci2,h,w,b', [('deconv', 128, 64, 32, 16, 16, 2), ('deconv', 64, 32, 0, 9, 37, 3), ('dgrad', 64, 32, 64, 12, 20, 2),
                                                  ('dgrad', 64, 32, 0, 16, 33, 2), ('dgrad', 128, 64, 0, 7, 5, 5), ('deconv', 32, 64, 64, 10, 48, 2)])
def test_fs2_h16_fractional_stride_layers(hip, kind, ci, co, ci2, h, w, b):
    """csrc/fs2_h16.hip (round 6, fp16 storage): ConvTranspose2d(k3, s2, p1, op1) forward and the input gradient of Conv2d(k3, s2, p1) per INPUT
    pixel -- the nine real (class, tap) products, weights resident in LDS -- with the optional 1 x 1 second source at output resolution,
    against float64 on the same fp16 operands: plain; bias + ReLU + byte mask out (transConv1 + skipConv2, models.py:293,299); residual + byte-mask
    gate (conv2_s^T); ragged sizes (rows / columns that are not multiples of the 32-pixel segments); bitwise run to run."""
    M, lib = hip['models'], hip['lib']
    torch.manual_seed(ci + co + h + ci2)
    x = _h(torch.randn(b, ci, h, w))
    if kind == 'deconv':
        wt = _h(torch.randn(ci, co, 3, 3) / (ci * 2.25) ** 0.5)
        ref = F.conv_transpose2d(x.double(), wt.double(), None, 2, 1, 1)
        w_eff = wt.permute(2, 3, 1, 0)
    else:   # input gradient of Conv2d(co -> ci, k3, s2, p1): x plays the output gradient
        wc = _h(torch.randn(ci, co, 3, 3) / (ci * 2.25) ** 0.5)
        ref = F.conv_transpose2d(x.double(), wc.double(), None, 2, 1, 1)      # (the adjoint of the strided convolution)
        w_eff = wc.permute(2, 3, 1, 0)
    x2 = w2 = None
    if ci2:
        x2 = _h(torch.randn(b, ci2, 2 * h, 2 * w))
        w2 = _h(torch.randn(co, ci2) / ci2 ** 0.5)
        ref = ref + torch.einsum('bkyx,nk->bnyx', x2.double(), w2.double())
    w_img, w2_img = M.pack_fs2(w_eff, w2)
    w_img, w2_img = w_img.to(DEV), (w2_img.to(DEV) if w2_img is not None else None)
    xin = nhwc(x).half().to(DEV)
    x2in = nhwc(x2).half().to(DEV) if ci2 else None
    bias = torch.randn(co)
    add = _h(torch.randn(b, co, 2 * h, 2 * w))
    gate = torch.randn(b, co, 2 * h, 2 * w)

    def run(bias_=None, add_=None, gate_=None, relu=0, mask=None):
        out = torch.full((b, 2 * h, 2 * w, co), 7.0, device=DEV, dtype=torch.float16)
        lib.call('spaa_fs2_h16', lib.hptr(xin), ci, ci, lib.hptr(w_img), lib.hptr(x2in) if ci2 else None, ci2, ci2,
                 lib.hptr(w2_img) if ci2 else None, lib.ptr(bias_) if bias_ is not None else None, lib.hptr(add_) if add_ is not None else None,
                 lib.ptr(gate_) if gate_ is not None else None, relu, lib.hptr(out), lib.ptr(mask) if mask is not None else None, co, b, h, w)
        return out

    out = run()
    e0 = rel_inf(nchw(out.float().cpu(), co), ref)
    assert e0 < 1.5e-3, (kind, ci, co, e0)
    assert torch.equal(out, run())
    mask = torch.zeros(b, 2 * h, 2 * w, co // 4, dtype=torch.uint8, device=DEV)
    out1 = run(bias_=bias.to(DEV), relu=1, mask=mask)
    want1 = F.relu(ref + bias.view(1, -1, 1, 1).double())
    assert rel_inf(nchw(out1.float().cpu(), co), want1) < 1.5e-3 and torch.equal(mask, lib.pack_gate_mask(out1.float()))
    gbits = lib.pack_gate_mask(nhwc(gate).to(DEV))
    out2 = run(add_=nhwc(add).half().to(DEV), gate_=gbits)
    want2 = (ref + add.double()) * (gate > 0)
    e2 = rel_inf(nchw(out2.float().cpu(), co), want2)
    assert e2 < 1.5e-3, (kind, e2)
    print(f'fs2_h16 {kind} {ci}->{co} (+{ci2}) {h}x{w} B={b}: rel err vs fp64 {e0:.1e} plain, {e2:.1e} residual + gate')


@pytest.mark.parametrize('ci,co,h,w,b', [(32, 64, 32, 32, 2), (64, 128, 20, 36, 2), (32, 64, 18, 70, 3), (64, 64, 8, 6, 5), (128, 64, 12, 12, 2)])
def test_s2f_h16_stride2_forward(hip, ci, co, h, w, b):
    """csrc/s2f_h16.hip (round 6, fp16 storage): Conv2d(k3, s2, p1) forward -- conv2 / conv2_s, the input gradient of transConv1 -- with all weights
    resident in LDS, against float64 on the same fp16 operands: plain; bias + residual + ReLU + byte mask out; byte-mask gate; ragged widths
    (rows that are not multiples of the 32-pixel segments), the zero padding on all four sides; bitwise run to run."""
    M, lib = hip['models'], hip['lib']
    torch.manual_seed(ci + co + h)
    x = _h(torch.randn(b, ci, h, w))
    wt = _h(torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5)
    ref = F.conv2d(x.double(), wt.double(), None, 2, 1)
    w_img = M.pack_s2f(wt.permute(2, 3, 0, 1)).to(DEV)
    xin = nhwc(x).half().to(DEV)
    ho, wo = h // 2, w // 2
    bias = torch.randn(co)
    add = _h(torch.randn(b, co, ho, wo))
    gate = torch.randn(b, co, ho, wo)

    def run(bias_=None, add_=None, gate_=None, relu=0, mask=None):
        out = torch.full((b, ho, wo, co), 7.0, device=DEV, dtype=torch.float16)
        lib.call('spaa_s2f_h16', lib.hptr(xin), ci, ci, lib.hptr(w_img), lib.ptr(bias_) if bias_ is not None else None,
                 lib.hptr(add_) if add_ is not None else None, lib.ptr(gate_) if gate_ is not None else None, relu, lib.hptr(out),
                 lib.ptr(mask) if mask is not None else None, co, b, h, w)
        return out

    out = run()
    e0 = rel_inf(nchw(out.float().cpu(), co), ref)
    assert e0 < 1.5e-3, (ci, co, e0)
    assert torch.equal(out, run())
    mask = torch.zeros(b, ho, wo, co // 4, dtype=torch.uint8, device=DEV)
    out1 = run(bias_=bias.to(DEV), add_=nhwc(add).half().to(DEV), relu=1, mask=mask)
    want1 = F.relu(ref + bias.view(1, -1, 1, 1).double() + add.double())
    e1 = rel_inf(nchw(out1.float().cpu(), co), want1)
    assert e1 < 1.5e-3 and torch.equal(mask, lib.pack_gate_mask(out1.float())), (ci, co, e1)
    out2 = run(gate_=lib.pack_gate_mask(nhwc(gate).to(DEV)))
    assert rel_inf(nchw(out2.float().cpu(), co), ref * (gate > 0)) < 1.5e-3
    print(f's2f_h16 {ci}->{co} {h}x{w} B={b}: rel err vs fp64 {e0:.1e} plain, {e1:.1e} bias + residual + ReLU')
    if ci == 128:     # a weight image that does not fit the 160 KB of LDS is refused, not truncated
        with pytest.raises(RuntimeError):
            lib.call('spaa_s2f_h16', lib.hptr(xin), ci, ci, lib.hptr(w_img), None, None, None, 0, lib.hptr(out), None, 128, b, h, w)


@pytest.mark.parametrize('h,w,b', [(32, 32, 2), (18, 70, 3), (8, 6, 5), (128, 128, 2)])
def test_s2f_x6_stride2_forward(hip, h, w, b):
    """csrc/s2f_x6.hip (round 6, fp32): Conv2d(32, 64, 3, 2, 1) forward -- conv2 / conv2_s -- with all weights (three bf16 planes) resident in
    LDS and the bf16x6 arithmetic, against float64 on the same fp32 operands at the accuracy of the other fp32 kernels: plain; bias +
    residual + ReLU + byte mask out; byte-mask gate; ragged widths, the zero padding on all four sides; bitwise run to run; and against the
    implicit-GEMM tile it replaces."""
    M, lib = hip['models'], hip['lib']
    ci, co = 32, 64
    torch.manual_seed(h + w)
    x = torch.randn(b, ci, h, w)
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, 2, 1)
    w_img = M.pack_s2f_x6(wt.permute(2, 3, 0, 1)).to(DEV)
    xin = nhwc(x).contiguous().to(DEV)
    ho, wo = h // 2, w // 2
    bias = torch.randn(co)
    add = torch.randn(b, co, ho, wo)
    gate = torch.randn(b, co, ho, wo)

    def run(bias_=None, add_=None, gate_=None, relu=0, mask=None):
        out = torch.full((b, ho, wo, co), 7.0, device=DEV)
        lib.call('spaa_s2f_x6', lib.ptr(xin), ci, ci, M.C_ptr(w_img), lib.ptr(bias_) if bias_ is not None else None,
                 lib.ptr(add_) if add_ is not None else None, lib.ptr(gate_) if gate_ is not None else None, relu, lib.ptr(out),
                 lib.ptr(mask) if mask is not None else None, co, b, h, w)
        return out

    out = run()
    e0 = rel_inf(nchw(out.cpu(), co), ref)
    assert e0 < 2e-6, e0
    assert torch.equal(out, run())
    mask = torch.zeros(b, ho, wo, co // 4, dtype=torch.uint8, device=DEV)
    out1 = run(bias_=bias.to(DEV), add_=nhwc(add).contiguous().to(DEV), relu=1, mask=mask)
    want1 = F.relu(ref + bias.view(1, -1, 1, 1).double() + add.double())
    e1 = rel_inf(nchw(out1.cpu(), co), want1)
    assert e1 < 2e-6 and torch.equal(mask, lib.pack_gate_mask(out1)), e1
    out2 = run(gate_=lib.pack_gate_mask(nhwc(gate).contiguous().to(DEV)))
    assert rel_inf(nchw(out2.cpu(), co), ref * (gate > 0)) < 2e-6
    print(f's2f_x6 {ci}->{co} {h}x{w} B={b}: rel err vs fp64 {e0:.1e} plain, {e1:.1e} bias + residual + ReLU')


def test_h16p_two_workgroups_per_cu(hip):
    """The patch-staged fp16 kernel's 64-wide stride-1 form with TWO workgroups per compute unit (csrc/tapconv_h16p.hip LEAN: one patch
    buffer reloaded per channel block, weight stages packed to 12 KiB, pad DMA slots into the patch buffer's pad piece -- VGG-16
    features.2, ResNet-18 layer1, the two-source input gradient of conv3 / skipConv3) is the SAME arithmetic as the one-workgroup form:
    bitwise equal outputs and masks, and right against float64; one to four channel blocks, ragged 16 x 32-pixel tiles, six- and
    nine-tap lists (a padded and an unpadded border), residual + ReLU + byte mask, gated input gradient, two sources."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(684)
    try:
        for ci, co, h, w, b in [(64, 64, 40, 70, 2), (32, 64, 17, 33, 3), (128, 48, 35, 20, 2), (96, 64, 64, 64, 2)]:
            x = _h(torch.randn(b, ci, h, w))
            wt = _h(torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5)
            bias = torch.randn(co)
            y = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1).float()
            add = _h(torch.randn_like(y))
            plan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV)
            xin = nhwc(x).half().to(DEV)
            outs = []
            for lean in (True, False):
                (cp.DEFAULT_DISABLE.discard if lean else cp.DEFAULT_DISABLE.add)('h16plean')
                out = torch.zeros(b, h, w, co, device=DEV, dtype=torch.float16)
                mask = torch.zeros(b, h, w, co // 4, dtype=torch.uint8, device=DEV)
                cp.FORCE_TILE = 68
                plan.run(xin, out, add=nhwc(add, co).half().to(DEV), act=lib.ACT_RELU, mask_out=mask)
                cp.FORCE_TILE = 0
                assert plan.last_tile == 68
                outs.append((out, mask))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (ci, co, h, w)
            assert rel_inf(nchw(outs[0][0].float().cpu(), co), F.relu(y + add)) < 1.5e-3, (ci, co)
            assert torch.equal(outs[0][1], lib.pack_gate_mask(outs[0][0].float()))
            if co % 32:
                continue
            # gated input gradient (Cout of the gradient GEMM = ci): 64-wide when ci <= 64
            g = _h(torch.randn(b, co, h, w))
            dplan = cp.conv_dgrad_plan(wt, 1, 1, DEV)
            gbits = lib.pack_gate_mask(torch.randn(b, h, w, ci, device=DEV))
            gins = []
            for lean in (True, False):
                (cp.DEFAULT_DISABLE.discard if lean else cp.DEFAULT_DISABLE.add)('h16plean')
                gin = torch.zeros(b, h, w, ci, device=DEV, dtype=torch.float16)
                cp.FORCE_TILE = 68
                dplan.run(nhwc(g).half().to(DEV), gin, gate_bits=gbits)
                cp.FORCE_TILE = 0
                gins.append(gin)
            assert torch.equal(gins[0], gins[1]), (ci, co)
        # two sources: conv(a, Wa) + conv(b, Wb) as one 64-wide launch (conv3^T(g3) + skipConv3^T(g5), models.py:294,298 backwards)
        wa, wb = _h(torch.randn(128, 64, 3, 3) / 24.0), _h(torch.randn(64, 64, 3, 3) / 24.0)
        ga, gb = _h(torch.randn(2, 128, 30, 37)), _h(torch.randn(2, 64, 30, 37))
        want = (F.conv_transpose2d(ga.double(), wa.double(), None, 1, 1) + F.conv_transpose2d(gb.double(), wb.double(), None, 1, 1)).float()
        plan2 = cp.conv_dgrad_plan_2src(wa, wb, DEV)
        res = []
        for lean in (True, False):
            (cp.DEFAULT_DISABLE.discard if lean else cp.DEFAULT_DISABLE.add)('h16plean')
            o = torch.zeros(2, 30, 37, 64, device=DEV, dtype=torch.float16)
            plan2.run(nhwc(ga).half().to(DEV), o, inp2=nhwc(gb).half().to(DEV))
            assert plan2.last_tile == 68
            res.append(o)
        assert torch.equal(res[0], res[1]) and rel_inf(nchw(res[0].float().cpu(), 64), want) < 1.5e-3
    finally:
        cp.FORCE_TILE = 0
        cp.DEFAULT_DISABLE.discard('h16plean')


def test_h16p_unpadded_3x3_and_its_input_gradient(hip):
    """The patch-staged fp16 kernel on an UNPADDED 3 x 3 / stride-1 layer (taps 0..2: output 2 smaller) and on its input gradient (taps
    -2..0: output 2 larger) -- Inception-v3's Conv2d_2a_3x3 / Conv2d_4a_3x3 behind classifier.py:29-33 -- against float64."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(685)
    try:
        for ci, co, h, w, b in [(96, 192, 37, 41, 2), (32, 64, 20, 70, 3), (64, 96, 16, 32, 1)]:
            x = _h(torch.randn(b, ci, h, w))
            wt = _h(torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5)
            bias = torch.randn(co)
            y = F.conv2d(x.double(), wt.double(), bias.double(), 1, 0).float()
            plan = cp.conv_fwd_plan(wt, bias, 1, 0, DEV)
            out = torch.zeros(b, h - 2, w - 2, co, device=DEV, dtype=torch.float16)
            mask = torch.zeros(b, h - 2, w - 2, co // 4, dtype=torch.uint8, device=DEV)
            cp.FORCE_TILE = 68
            plan.run(nhwc(x).half().to(DEV), out, act=lib.ACT_RELU, mask_out=mask)
            assert plan.last_tile == 68, plan.last_tile
            assert rel_inf(nchw(out.float().cpu(), co), F.relu(y)) < 1.5e-3, (ci, co, rel_inf(nchw(out.float().cpu(), co), F.relu(y)))
            assert torch.equal(mask, lib.pack_gate_mask(out.float()))
            g = _h(torch.randn(b, co, h - 2, w - 2))
            dplan = cp.conv_dgrad_plan(wt, 1, 0, DEV)
            gin = torch.zeros(b, h, w, ci, device=DEV, dtype=torch.float16)
            dplan.run(nhwc(g).half().to(DEV), gin)
            assert dplan.last_tile == 68, dplan.last_tile
            cp.FORCE_TILE = 0
            wantg = F.conv_transpose2d(g.double(), wt.double(), None, 1, 0).float()
            assert rel_inf(nchw(gin.float().cpu(), ci), wantg) < 1.5e-3, (ci, co)
    finally:
        cp.FORCE_TILE = 0


def test_h16p_fused_relu_maxpool(hip):
    """conv -> ReLU -> MaxPool2d(2, 2) (torchvision VGG-16 `features`, classifier.py:21-24) as ONE launch of the patch-staged fp16 kernel
    (csrc/tapconv_h16p.hip POOL: the pool in the epilogue, the full-size activation never written) against the same plan run as
    convolution + spaa_maxpool_fwd: bitwise the pooled values and the arg-max bytes (code of the FIRST maximum | 0x80 if positive);
    64- and 128-wide layers (two workgroups per CU and one), ragged 16 x 32-pixel tiles, the decision-only form without arg-max bytes,
    and the fall-back inside ConvPlan.run where another kernel serves the layer (14 x 14 maps: canvas form)."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(686)
    try:
        for ci, co, h, w, b, fused in [(64, 64, 32, 64, 2, True), (64, 128, 28, 28, 3, True), (128, 256, 20, 70, 2, True), (256, 256, 56, 56, 2, True),
                                       (128, 128, 14, 14, 4, False)]:
            x = _h(torch.randn(b, ci, h, w))
            wt = _h(torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5)
            bias = torch.randn(co)
            plan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV)
            xin = nhwc(x).half().to(DEV)
            full = torch.zeros(b, h, w, co, device=DEV, dtype=torch.float16)
            res = []
            for fuse in (True, False):
                (cp.DEFAULT_DISABLE.discard if fuse else cp.DEFAULT_DISABLE.add)('h16ppool')
                pooled = torch.full((b, h // 2, w // 2, co), -1.0, device=DEV, dtype=torch.float16)
                arg = torch.full((b, h // 2, w // 2, co), 77, device=DEV, dtype=torch.uint8)
                if fused:
                    cp.FORCE_TILE = 68
                plan.run(xin, full, act=lib.ACT_RELU, pool=(pooled, arg, True))
                cp.FORCE_TILE = 0
                assert plan.last_pool_fused == (fuse and fused), (ci, co, h, w, plan.last_tile)
                res.append((pooled, arg))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (ci, co, h, w)
            y = F.max_pool2d(F.relu(F.conv2d(x.double(), wt.double(), bias.double(), 1, 1)), 2, 2).float()
            assert rel_inf(nchw(res[0][0].float().cpu(), co), y) < 1.5e-3
            cp.DEFAULT_DISABLE.discard('h16ppool')
            pooled2 = torch.zeros_like(res[0][0])
            arg2 = torch.full_like(res[0][1], 5)
            if fused:
                cp.FORCE_TILE = 68
            plan.run(xin, full, act=lib.ACT_RELU, pool=(pooled2, arg2, False))     # a forward pass nobody differentiates
            cp.FORCE_TILE = 0
            assert torch.equal(pooled2, res[0][0]) and (not fused or bool((arg2 == 5).all()))
    finally:
        cp.FORCE_TILE = 0
        cp.DEFAULT_DISABLE.discard('h16ppool')


def test_vgg16_fp16_pool_fusion_is_bitwise(hip):
    """VGG-16 in fp16 storage with the 2 x 2 max-pools fused into the convolutions before them (VGG16Body.fuse_pool) against the same
    engine with separate pool launches: logits and input gradient bitwise equal."""
    M = hip['models']
    csd = syn.vgg16_state_dict(2, logit_gain=20.0)
    clf = hip['clf'].Classifier('vgg16', DEV, state_dict=csd)
    ce = clf.engine(2, (256, 256), (240, 240), storage='f16')
    assert ce.body.fuse_pool
    im = M.to_nhwc4(syn.scenes(9, 2, (256, 256)).to(DEV))
    q = torch.zeros(2, 1000, device=DEV)
    q[0, 3] = -64.0
    q[1, 17] = -64.0
    l1 = ce.forward(im).clone()
    g1 = ce.backward(q).clone()
    ce.body.fuse_pool = False
    l0 = ce.forward(im).clone()
    g0 = ce.backward(q).clone()
    ce.body.fuse_pool = True
    assert torch.equal(l1, l0) and torch.equal(g1, g0) and torch.isfinite(g1).all() and float(g1.abs().max()) > 0


def test_h16p_pool_adjoint_prologue(hip):
    """The input gradient of a convolution whose ReLU output fed a 2 x 2 / stride-2 max-pool (torchvision VGG-16, classifier.py:21-24) with
    the pool's adjoint as the patch prologue of the patch-staged fp16 kernel (csrc/tapconv_h16p.hip UNP: the pooled gradient and the arg-max
    bytes in, the full-size gradient never written) against spaa_maxpool_bwd + the plain launch: bitwise equal; 64-wide layers and wider ones
    with short K (two workgroups per CU), byte-mask gate of the layer below, ragged tiles, and the fall-back inside ConvPlan.run (long K)."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(687)
    try:
        for ci, co, h, w, b, fused in [(64, 64, 32, 64, 2, True), (64, 128, 28, 36, 3, True), (128, 256, 20, 70, 2, True), (64, 512, 16, 32, 2, True), (128, 512, 16, 32, 2, False)]:
            # (forward layer ci -> co at h x w, then ReLU and the pool; the input gradient maps co -> ci channels)
            wt = _h(torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5)
            dplan = cp.conv_dgrad_plan(wt, 1, 1, DEV)
            act = _h(torch.randn(b, h, w, co, device=DEV)).half()                           # the layer's ReLU output is max(act, 0)
            pooled = torch.zeros(b, h // 2, w // 2, co, device=DEV, dtype=torch.float16)
            arg = torch.zeros(b, h // 2, w // 2, co, device=DEV, dtype=torch.uint8)
            lib.call('spaa_maxpool_fwd_f16', lib.hptr(torch.relu(act)), lib.hptr(pooled), lib.ptr(arg), b, h, w, co, h // 2, w // 2, 2, 2, 0, co, 0)
            gp = _h(torch.randn(b, h // 2, w // 2, co, device=DEV)).half()
            gbits = lib.pack_gate_mask(torch.randn(b, h, w, ci, device=DEV))
            res = []
            for fuse in (True, False):
                (cp.DEFAULT_DISABLE.discard if fuse else cp.DEFAULT_DISABLE.add)('h16punp')
                gfull = torch.full((b, h, w, co), 3.0, device=DEV, dtype=torch.float16)
                gin = torch.zeros(b, h, w, ci, device=DEV, dtype=torch.float16)
                cp.FORCE_TILE = 68
                dplan.run(gp, gin, gate_bits=gbits, unpool=(arg, gfull))
                cp.FORCE_TILE = 0
                assert dplan.last_tile == 68 and dplan.last_unpool_fused == (fuse and fused), (ci, co, dplan.last_unpool_fused)
                res.append(gin)
            assert torch.equal(res[0], res[1]) and float(res[0].float().abs().max()) > 0, (ci, co, h, w)
    finally:
        cp.FORCE_TILE = 0
        cp.DEFAULT_DISABLE.discard('h16punp')


def test_h16p_stride2_forward(hip):
    """The patch-staged fp16 kernel's stride-2 FORWARD form (csrc/tapconv_h16p.hip S = 2: ShadingNetSPAA.conv2 / conv2_s,
    models.py:224,230 of the reference, the input gradient of transConv1, the classifiers' 3 x 3 / stride-2 layers) against float64
    on the same fp16 operands: one and several 32-channel blocks (the single patch buffer reloaded), 64- and 128-wide N tiles,
    padded and unpadded, ragged 8 x 32-pixel tiles, residual + ReLU + gate bytes, fp16 and fp32 output."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(682)
    try:
        for ci, co, p, h, w, b in [(32, 64, 1, 44, 70, 2), (64, 128, 1, 37, 41, 2), (96, 192, 0, 35, 35, 1), (32, 64, 1, 128, 128, 1), (128, 96, 1, 20, 66, 2)]:
            x = _h(torch.randn(b, ci, h, w))
            wt = _h(torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5)
            bias = torch.randn(co)
            y = F.conv2d(x.double(), wt.double(), bias.double(), 2, p).float()
            add = _h(torch.randn_like(y))
            plan = cp.conv_fwd_plan(wt, bias, 2, p, DEV)
            ho, wo = y.shape[2:]
            out = torch.zeros(b, ho, wo, co, device=DEV, dtype=torch.float16)
            mask = torch.zeros(b, ho, wo, co // 4, dtype=torch.uint8, device=DEV)
            cp.FORCE_TILE = 68
            plan.run(nhwc(x).half().to(DEV), out, add=nhwc(add, co).half().to(DEV), act=lib.ACT_RELU, mask_out=mask)
            assert plan.last_tile == 68, (plan.last_tile, ci, co)
            want = F.relu(y + add)
            got = nchw(out.float().cpu(), co)
            assert rel_inf(got, want) < 1.5e-3, (ci, co, p, rel_inf(got, want))
            assert torch.equal(mask, lib.pack_gate_mask(out.float()))
            out32 = torch.zeros(b, ho, wo, co, device=DEV)
            plan.run(nhwc(x).half().to(DEV), out32)
            cp.FORCE_TILE = 0
            assert plan.last_tile == 68 and rel_inf(nchw(out32.cpu(), co), y) < 2e-5, (ci, co, p, rel_inf(nchw(out32.cpu(), co), y))
    finally:
        cp.FORCE_TILE = 0


@pytest.mark.parametrize('tile', [0, 60, 61, 62, 63, 64, 65, 68])
def test_tapconv_fp16_storage(hip, tile):
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(31 + tile)
    try:
        for ci, co, k, s, h, w, b in [(64, 96, 3, 1, 19, 23, 2), (32, 64, 3, 2, 22, 18, 3), (128, 40, 1, 1, 17, 9, 2),
                                      (32, 3, 3, 1, 20, 24, 2), (96, 130, 3, 1, 7, 9, 1), (64, 64, 5, 1, 11, 10, 2)]:
            x = _h(torch.randn(b, ci, h, w))
            wt = _h(torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5)
            bias = torch.randn(co)
            y = F.conv2d(x.double(), wt.double(), bias.double(), s, k // 2).float()
            add = _h(torch.randn_like(y))
            plan = cp.conv_fwd_plan(wt, bias, s, k // 2, DEV)
            cs = (co + 3) // 4 * 4
            ho, wo = y.shape[2:]
            cp.FORCE_TILE = tile
            # fp16 in -> fp16 out, residual + ReLU + byte mask
            out = torch.zeros(b, ho, wo, cs, device=DEV, dtype=torch.float16)
            mask = torch.zeros(b, ho, wo, cs // 4, dtype=torch.uint8, device=DEV) if co % 4 == 0 else None
            plan.run(nhwc(x).half().to(DEV), out, add=nhwc(add, cs).half().to(DEV), act=lib.ACT_RELU, mask_out=mask)
            want = F.relu(y + add)
            got = nchw(out.float().cpu(), co)
            assert rel_inf(got, want) < 1.5e-3, (tile, ci, co, k, s, rel_inf(got, want))
            if mask is not None:
                assert torch.equal(mask, lib.pack_gate_mask(out.float())), (tile, ci, co)
            # fp16 in -> fp32 out (the image-side layers of the fp16 path), pre-clamp second output
            out32, aux32 = torch.zeros(b, ho, wo, cs, device=DEV), torch.zeros(b, ho, wo, cs, device=DEV)
            plan.run(nhwc(x).half().to(DEV), out32, act=lib.ACT_RELU_CLAMP1, aux_out=aux32)
            assert rel_inf(nchw(out32.cpu(), co), F.relu(y).clamp(max=1)) < 2e-5 * max(1.0, float(F.relu(y).max()))
            assert rel_inf(nchw(aux32.cpu(), co), F.relu(y)) < 2e-5
            # input gradient: fp16 gradient in -> fp16 gradient out, gated by a byte mask
            gy = _h(torch.randn(b, co, ho, wo))
            gx_ref = torch.nn.grad.conv2d_input((b, ci, h, w), wt.double(), gy.double(), s, k // 2).float()
            gate_act = torch.randn(b, h, w, ci, device=DEV)
            dplan = cp.conv_dgrad_plan(wt, s, k // 2, DEV)
            if dplan.cin_p % 32 == 0:
                gx = torch.zeros(b, h, w, ci, device=DEV, dtype=torch.float16)
                dplan.run(nhwc(gy, dplan.cin_p).half().to(DEV), gx, gate_bits=lib.pack_gate_mask(gate_act))
                want_g = gx_ref * (nchw(gate_act.cpu(), ci) > 0)
                assert rel_inf(nchw(gx.float().cpu(), ci), want_g) < 1.5e-3, (tile, 'dgrad', ci, co, k, s)
            cp.FORCE_TILE = 0
        # transposed convolutions: strided classes (k3) and folded classes (k2)
        for kk, pad, op in [(3, 1, 1), (2, 0, 0)]:
            xt = _h(torch.randn(2, 64, 11, 13))
            wtt = _h(torch.randn(64, 32, kk, kk) / 16)
            bt = torch.randn(32)
            ref = F.relu(F.conv_transpose2d(xt.double(), wtt.double(), bt.double(), 2, pad, op).float())
            tplan = cp.deconv_fwd_plan(wtt, bt, 2, pad, DEV)
            outt = torch.zeros(2, ref.shape[2], ref.shape[3], 32, device=DEV, dtype=torch.float16)
            cp.FORCE_TILE = tile
            tplan.run(nhwc(xt).half().to(DEV), outt, act=lib.ACT_RELU)
            cp.FORCE_TILE = 0
            assert rel_inf(nchw(outt.float().cpu()), ref) < 1.5e-3, (tile, 'deconv', kk)
        # fp32 image in -> fp16 activation out (first layers: the small-Cin kernels with the fp16 epilogue)
        for ci, co, k, s in [(3, 32, 3, 2), (6, 32, 3, 2), (3, 64, 7, 2), (3, 64, 3, 1)]:
            x = torch.rand(2, ci, 24, 28)
            wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
            bias = torch.randn(co)
            y = F.relu(F.conv2d(x, wt, bias, s, k // 2))
            plan = cp.conv_fwd_plan(wt, bias, s, k // 2, DEV)
            out = torch.zeros(2, y.shape[2], y.shape[3], co, device=DEV, dtype=torch.float16)
            mask = torch.zeros(2, y.shape[2], y.shape[3], co // 4, dtype=torch.uint8, device=DEV)
            plan.run(nhwc(x, plan.cin_p).to(DEV), out, act=lib.ACT_RELU, mask_out=mask)
            assert rel_inf(nchw(out.float().cpu(), co), y) < 1e-3, ('image->f16', ci, co, k, s)
            assert torch.equal(mask, lib.pack_gate_mask(out.float()))
    finally:
        cp.FORCE_TILE = 0


@pytest.mark.parametrize('m,k,n', [(64, 512, 1000), (8, 2048, 1000), (3, 4096, 1000), (5, 12, 10), (64, 1000, 512)])
def test_small_linear(hip, m, k, n):
    """csrc/linear_small.hip (the classifiers' last layer and its input gradient at batch <= 64: classifier.py:60 of the reference,
    torchvision `fc`): against fp64, at the error of an fp32 dot product, bitwise run to run, and through the plan wrapper
    (convplan.SmallLinearPlan: the kernel for a plain call, the 1 x 1 convolution tiles when an epilogue is asked for)."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(m + k)
    w, bias, x = torch.randn(n, k) / k ** 0.5, torch.randn(n), torch.randn(m, k)
    ref = x.double() @ w.double().t() + bias.double()
    plan = cp.linear_fwd_plan(w, bias, DEV, 'fc')
    assert isinstance(plan, cp.SmallLinearPlan)
    y = torch.full((m, 1, 1, n), float('nan'), device=DEV)
    plan.run(x.view(m, 1, 1, k).to(DEV), y)
    assert plan.last_tile == 75
    y2 = torch.zeros_like(y)
    plan.run(x.view(m, 1, 1, k).to(DEV), y2)
    assert torch.equal(y, y2) and rel_inf(y.cpu().view(m, n), ref.float()) < 2e-6
    if k % 32 == 0:
        y3 = torch.zeros_like(y)
        plan.run(x.view(m, 1, 1, k).to(DEV), y3, act=lib.ACT_RELU)
        assert plan.last_tile != 75 and rel_inf(y3.cpu().view(m, n), F.relu(ref).float()) < 2e-6
    if n % 4 == 0:
        g = torch.randn(m, n)
        dplan = cp.linear_dgrad_plan(w, DEV, 'fc_dgrad')
        gx = torch.zeros(m, 1, 1, k, device=DEV)
        dplan.run(g.view(m, 1, 1, n).to(DEV), gx)
        assert dplan.last_tile == 75 and rel_inf(gx.cpu().view(m, k), (g.double() @ w.double()).float()) < 2e-6


def test_tapconv_fp16_split_k(hip):
    """Split-K of the fp16 implicit-GEMM kernel (skinny GEMMs: a fully connected layer at batch 64, a 7 x 7 x 512 layer): raw fp32
    partial sums + a fixed-order second pass with the epilogue; against fp64 on the same fp16-rounded operands and against the
    unsplit kernel (same products, another summation order), fp16 and fp32 output, bias + ReLU + float gate."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(9)
    try:
        # linear layer: 64 samples, K = 6272 -> 520
        k, n = 6272, 520
        w = _h(torch.randn(n, k) / k ** 0.5)
        bias = torch.randn(n)
        x = _h(torch.randn(64, k))
        ref = F.relu(x.double() @ w.double().t() + bias.double()).float()
        plan = cp.linear_fwd_plan(w, bias, DEV, 'fc')
        outs = {}
        for ks in (1, 4, 7):
            cp.FORCE_KSPLIT = ks
            cp.FORCE_TILE = 61
            y = torch.zeros(64, 1, 1, n, device=DEV, dtype=torch.float16)
            plan.run(x.view(64, 1, 1, k).half().to(DEV), y, act=lib.ACT_RELU)
            assert rel_inf(y.float().cpu().view(64, n), ref) < 1.5e-3, ks
            y32 = torch.zeros(64, 1, 1, n, device=DEV)
            plan.run(x.view(64, 1, 1, k).half().to(DEV), y32, act=lib.ACT_RELU)
            assert rel_inf(y32.cpu().view(64, n), ref) < 2e-5, ks
            outs[ks] = y32
        assert rel_inf(outs[4], outs[1]) < 2e-6 and rel_inf(outs[7], outs[1]) < 2e-6
        # 3x3 convolution on 7 x 7 maps, 512 -> 256, and its input gradient with a float gate
        xc = _h(torch.randn(5, 512, 7, 7))
        wc = _h(torch.randn(256, 512, 3, 3) / (512 * 9) ** 0.5)
        yc = F.conv2d(xc.double(), wc.double(), None, 1, 1).float()
        cplan = cp.conv_fwd_plan(wc, None, 1, 1, DEV)
        cp.FORCE_KSPLIT, cp.FORCE_TILE = 6, 60
        oc = torch.zeros(5, 7, 7, 256, device=DEV, dtype=torch.float16)
        cplan.run(nhwc(xc).half().to(DEV), oc)
        assert rel_inf(nchw(oc.float().cpu(), 256), yc) < 1.5e-3
        gy = _h(torch.randn(5, 256, 7, 7))
        gx_ref = torch.nn.grad.conv2d_input((5, 512, 7, 7), wc.double(), gy.double(), 1, 1).float()
        gate = torch.randn(5, 7, 7, 512, device=DEV).half()
        dplan = cp.conv_dgrad_plan(wc, 1, 1, DEV)
        gx = torch.zeros(5, 7, 7, 512, device=DEV, dtype=torch.float16)
        dplan.run(nhwc(gy).half().to(DEV), gx, gate=gate)
        assert rel_inf(nchw(gx.float().cpu(), 512), gx_ref * (nchw(gate.float().cpu(), 512) > 0)) < 1.5e-3
    finally:
        cp.FORCE_KSPLIT, cp.FORCE_TILE = 0, 0


def test_tapconv_fp16_patch_staged(hip):
    """fp16-storage 3x3 / stride-1 layers on the patch-staged kernel (csrc/tapconv_h16p.hip, tile 68): several workgroup tiles
    per image with ragged right / bottom edges, one and two channel tiles, the 64-wide instantiation; forward (residual + ReLU +
    byte mask, fp16 out; fp32 out) and input gradient (byte-mask gate), against fp64 on the same fp16-rounded operands."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(68)
    try:
        for ci, co, h, w, b in [(128, 256, 40, 70, 2), (256, 128, 33, 32, 1), (64, 64, 17, 65, 2), (32, 192, 16, 32, 1), (96, 72, 5, 3, 3)]:
            x = _h(torch.randn(b, ci, h, w))
            wt = _h(torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5)
            bias = torch.randn(co)
            y = F.conv2d(x.double(), wt.double(), bias.double(), 1, 1).float()
            add = _h(torch.randn_like(y))
            plan = cp.conv_fwd_plan(wt, bias, 1, 1, DEV)
            cp.FORCE_TILE = 68
            out = torch.zeros(b, h, w, co, device=DEV, dtype=torch.float16)
            mask = torch.zeros(b, h, w, co // 4, dtype=torch.uint8, device=DEV)
            plan.run(nhwc(x).half().to(DEV), out, add=nhwc(add).half().to(DEV), act=lib.ACT_RELU, mask_out=mask)
            assert plan.last_tile == 68, plan.last_tile
            assert rel_inf(nchw(out.float().cpu(), co), F.relu(y + add)) < 1.5e-3, (ci, co, h, w)
            assert torch.equal(mask, lib.pack_gate_mask(out.float()))
            out32 = torch.zeros(b, h, w, co, device=DEV)
            plan.run(nhwc(x).half().to(DEV), out32)
            assert rel_inf(nchw(out32.cpu(), co), y) < 2e-5, (ci, co, h, w, rel_inf(nchw(out32.cpu(), co), y))
            cp.FORCE_TILE = 60
            ref60 = torch.zeros(b, h, w, co, device=DEV)
            plan.run(nhwc(x).half().to(DEV), ref60)
            cp.FORCE_TILE = 68
            assert rel_inf(out32, ref60) < 2e-6       # (the same products, another summation order)
            gy = _h(torch.randn(b, co, h, w))
            gx_ref = torch.nn.grad.conv2d_input((b, ci, h, w), wt.double(), gy.double(), 1, 1).float()
            gate_act = torch.randn(b, h, w, ci, device=DEV)
            dplan = cp.conv_dgrad_plan(wt, 1, 1, DEV)
            if dplan.cin_p % 32 == 0:
                gx = torch.zeros(b, h, w, ci, device=DEV, dtype=torch.float16)
                dplan.run(nhwc(gy, dplan.cin_p).half().to(DEV), gx, gate_bits=lib.pack_gate_mask(gate_act))
                assert dplan.last_tile == 68
                assert rel_inf(nchw(gx.float().cpu(), ci), gx_ref * (nchw(gate_act.cpu(), ci) > 0)) < 1.5e-3, ('dgrad', ci, co)
        # folded stride-2 transposed layers (four taps, GEMM columns = parity class * Cout + channel): a 3x3 / s2 transposed
        # convolution (ReLU + byte mask) and the input gradient of a 3x3 / s2 convolution (byte-mask gate, residual)
        for ci, co, h, w, b in [(128, 64, 40, 70, 2), (64, 32, 33, 32, 3), (32, 96, 17, 65, 1)]:
            xt = _h(torch.randn(b, ci, h, w))
            wtt = _h(torch.randn(ci, co, 3, 3) / (ci * 9) ** 0.5)
            bt = torch.randn(co)
            ref = F.relu(F.conv_transpose2d(xt.double(), wtt.double(), bt.double(), 2, 1, 1).float())
            tplan = cp.deconv_fwd_plan(wtt, bt, 2, 1, DEV)
            outt = torch.zeros(b, 2 * h, 2 * w, co, device=DEV, dtype=torch.float16)
            maskt = torch.zeros(b, 2 * h, 2 * w, co // 4, dtype=torch.uint8, device=DEV)
            cp.FORCE_TILE = 68
            tplan.run(nhwc(xt).half().to(DEV), outt, act=lib.ACT_RELU, mask_out=maskt)
            if tplan.nfold == 4:
                assert tplan.last_tile == 68, tplan.last_tile
            assert rel_inf(nchw(outt.float().cpu(), co), ref) < 1.5e-3, ('deconv', ci, co)
            assert torch.equal(maskt, lib.pack_gate_mask(outt.float()))
            # input gradient of a 3x3 / s2 convolution co <- ci ... (weights [ci_out = co2, ci_in]): gradient wrt a (2h x 2w) input
            wt2 = _h(torch.randn(ci, co, 3, 3) / (co * 9) ** 0.5)           # conv co -> ci, stride 2
            gy = _h(torch.randn(b, ci, h, w))
            gx_ref = torch.nn.grad.conv2d_input((b, co, 2 * h, 2 * w), wt2.double(), gy.double(), 2, 1).float()
            dplan = cp.conv_dgrad_plan(wt2, 2, 1, DEV)
            gate_act = torch.randn(b, 2 * h, 2 * w, co, device=DEV)
            addt = _h(torch.randn(b, 2 * h, 2 * w, co))
            gx = torch.zeros(b, 2 * h, 2 * w, co, device=DEV, dtype=torch.float16)
            dplan.run(nhwc(gy).half().to(DEV), gx, add=addt.half().to(DEV), gate_bits=lib.pack_gate_mask(gate_act))
            if dplan.nfold == 4:
                assert dplan.last_tile == 68, dplan.last_tile
            want = (gx_ref.permute(0, 2, 3, 1) + addt) * (gate_act.cpu() > 0)
            assert rel_inf(gx.float().cpu(), want) < 1.5e-3, ('s2 dgrad', ci, co)
    finally:
        cp.FORCE_TILE = 0


def test_thin_output_on_matrix_cores(hip):
    """Thin outputs with the output-parity classes folded into the N dimension of a matrix-core tile (csrc/tapconv_thinmf.hip,
    tile 72; fp32 input: bf16x6 arithmetic): the input gradients of the stride-2 first layers (ResNet stem 7 x 7, conv1 / Inception
    3 x 3), of a stride-1 first layer (VGG-16: one class, 3 x 3 box) and a 1 x 3 box, several workgroup tiles with ragged edges,
    against fp64; equal to the VALU kernel (tile 29) to fp32 rounding; residual / multiplicative-gate epilogue and the generic one."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(72)
    for ci, co, k, s, pad, h, w, b in [(3, 64, 7, 2, 3, 62, 90, 2), (3, 32, 3, 2, 1, 52, 70, 3), (3, 64, 3, 1, 1, 27, 40, 2),
                                       (3, 32, 3, 2, 0, 31, 33, 2), (2, 96, (1, 3), 1, (0, 1), 13, 35, 1)]:
        kh, kw = (k, k) if isinstance(k, int) else k
        wt = torch.randn(co, ci, kh, kw) / (ci * kh * kw) ** 0.5
        x = torch.zeros(b, ci, h, w, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x, wt.double(), None, s, pad)
        gy = torch.randn(y.shape[0], co, y.shape[2], y.shape[3])
        y.backward(gy.double())
        dplan = cp.conv_dgrad_plan(wt, s, pad, DEV)
        assert dplan.thin_ok()
        add = torch.randn(b, h, w, 4, device=DEV)
        add[..., ci:] = 0
        gate = torch.rand(b, h, w, 4, device=DEV)
        res = {}
        try:
            for tile in (72, 29):
                cp.FORCE_TILE = tile
                gx = torch.zeros(b, h, w, 4, device=DEV)
                dplan.run(nhwc(gy, dplan.cin_p).to(DEV), gx)
                assert dplan.last_tile == tile, dplan.last_tile
                assert rel_inf(nchw(gx.cpu(), ci), x.grad.float()) < 2e-6, (ci, co, k, s, tile, rel_inf(nchw(gx.cpu(), ci), x.grad.float()))
                if ci < 4:
                    assert (gx[..., ci:] == 0).all()
                gx2 = torch.zeros(b, h, w, 4, device=DEV)
                dplan.run(nhwc(gy, dplan.cin_p).to(DEV), gx2, add=add, gate=gate, gate_mode=lib.GATE_MUL)
                want = (x.grad.float().permute(0, 2, 3, 1) + add.cpu()[..., :ci]) * gate.cpu()[..., :ci]
                assert rel_inf(gx2.cpu()[..., :ci], want) < 2e-6, (ci, co, k, s, tile, 'epilogue')
                res[tile] = gx
            assert rel_inf(res[72], res[29]) < 1e-6
            # the generic epilogue (shared store4_t): a channel window of a wider buffer
            cp.FORCE_TILE = 72
            wide = torch.zeros(b, h, w, 8, device=DEV)
            dplan.run(nhwc(gy, dplan.cin_p).to(DEV), wide, out_coff=4)
            assert dplan.last_tile == 72 and (wide[..., :4] == 0).all()
            assert rel_inf(wide.cpu()[..., 4:4 + ci], x.grad.float().permute(0, 2, 3, 1)) < 2e-6
        finally:
            cp.FORCE_TILE = 0


def test_thin_output_from_fp16_activation(hip):
    """fp16-storage mode: the image-side input gradients (conv1 / conv1_s: 32 -> 3 over four parity classes; ResNet stem: 64 -> 3,
    7 x 7) read an fp16 gradient and write the fp32 image gradient: the patch-staged VALU kernel's fp16-input form (tile 29)
    against fp64 on the same fp16-rounded operands, with the residual / multiplicative-gate epilogues the engine uses."""
    cp, lib = hip['cp'], hip['lib']
    torch.manual_seed(5)
    for ci, co, k, s, h, w, b in [(3, 32, 3, 2, 24, 40, 3), (3, 64, 7, 2, 30, 28, 2), (2, 32, 3, 1, 16, 18, 2)]:
        wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        gy = _h(torch.randn(b, co, ho, wo))
        x = torch.zeros(b, ci, h, w, dtype=torch.float64, requires_grad=True)
        F.conv2d(x, wt.double(), None, s, k // 2).backward(gy.double())
        dplan = cp.conv_dgrad_plan(wt, s, k // 2, DEV)
        add = torch.randn(b, h, w, 4, device=DEV)
        add[..., ci:] = 0
        gate = torch.rand(b, h, w, 4, device=DEV)
        for tile in (0, 29, 63):   # (0 -> 72, the folded matrix-core kernel, and 63: fp16 WEIGHTS as well: 2^-11 per product)
            cp.FORCE_TILE = tile
            tol = 2e-5 if tile == 29 else 1.5e-3
            try:
                gx = torch.zeros(b, h, w, 4, device=DEV)
                dplan.run(nhwc(gy, dplan.cin_p).half().to(DEV), gx)
                assert dplan.last_tile == (72 if tile == 0 else tile), dplan.last_tile
                assert rel_inf(nchw(gx.cpu(), ci), x.grad.float()) < tol, (ci, co, k, tile)
                gx2 = torch.zeros(b, h, w, 4, device=DEV)
                dplan.run(nhwc(gy, dplan.cin_p).half().to(DEV), gx2, add=add, gate=gate, gate_mode=lib.GATE_MUL)
                want = (x.grad.float().permute(0, 2, 3, 1) + add.cpu()[..., :ci]) * gate.cpu()[..., :ci]
                assert rel_inf(gx2.cpu()[..., :ci], want) < tol, (ci, co, k, tile, 'epilogue')
            finally:
                cp.FORCE_TILE = 0


def test_fp16_storage_pcnet_and_classifier(hip, golden_dir):
    """fp16-storage engines vs the fp32 oracle on the reference's golden PCNet case: forward values to fp16 rounding
    accumulated over the 14 layers, input gradients to a few percent (fp16-rounded activations flip more ReLU gates than
    fp32 rounding does).  Stated tolerances = 3x the measured values (profiles/r02_parity.txt)."""
    M = hip['models']
    z = load(golden_dir, 'pcnet_64')
    cam_sz = tuple(int(v) for v in z['cam_sz'])
    sd = syn.pcnet_state_dict(int(z['seed']), cam_sz=cam_sz, mask=str(z['mask']))
    pc = make_pcnet(hip, sd, cam_sz)
    s = torch.from_numpy(z['s'])
    x = syn.scenes(5, 2, cam_sz)
    r = torch.from_numpy(z['r'])
    xc = x.clone().requires_grad_(True)
    yc = so.pcnet_forward(sd, xc, s)
    (yc * r).sum().backward()
    eng = pc.engine(2, cam_sz, storage='f16')
    assert eng.a['X4'].dtype == torch.float16 and eng.a['Y'].dtype == torch.float32 and eng.g['P7'].dtype == torch.float16
    eng.set_scene(M.to_nhwc4(s.to(DEV)))
    y4 = eng.forward(M.to_nhwc4(x.to(DEV)), clamp01=False)
    e_y = rel_inf(M.to_nchw(y4), yc)
    gP = M.to_nhwc4((r * ((yc > 0) & (yc < 1))).to(DEV))          # cotangent at conv6's pre-activation, oracle's clamp gate
    gx = M.to_nchw(eng.backward(gP)).cpu()
    e_g = rel_l2(gx, xc.grad)
    print(f'fp16-storage PCNet 64x64: forward rel Linf {e_y:.2e}, input-gradient rel L2 {e_g:.2e}')
    assert e_y < 1.5e-2 and e_g < 1.5e-1
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    for body, csd_, insz, tol_l, tol_g in (('resnet18', csd, (56, 56), 3e-2, 3e-1),
                                           ('vgg16', syn.vgg16_state_dict(3, logit_gain=5.0, fc_width=512), (224, 224), 3e-2, 3e-1),
                                           ('inception_v3', syn.inception_v3_state_dict(2, logit_gain=20.0), (107, 107), 3e-2, 3e-1)):
        hsz = {'resnet18': 64, 'vgg16': 256, 'inception_v3': 128}[body]
        crop = {'resnet18': (60, 60), 'vgg16': (240, 240), 'inception_v3': (120, 120)}[body]
        im = syn.scenes(8, 2, (hsz, hsz))
        imc = im.clone().requires_grad_(True)
        raw, p, idx = so.OracleClassifier(body, csd_, input_sz=insz)(imc, crop)
        q = torch.zeros(2, 1000)
        q[0, int(idx[0, 1])] = -1.0
        q[1, int(idx[1, 2])] = -1.0
        (raw * q).sum().backward()
        clf = hip['clf'].Classifier(body, DEV, state_dict=csd_, input_sz=insz)
        ce = clf.engine(2, (hsz, hsz), crop, storage='f16')
        logits = ce.forward(M.to_nhwc4(im.to(DEV)))
        e_l = rel_inf(logits, raw)
        g = M.to_nchw(ce.backward((q * 64).to(DEV).contiguous())).cpu() / 64
        e_gc = rel_l2(g, imc.grad)
        print(f'fp16-storage {body}: logits rel Linf {e_l:.2e}, input-gradient rel L2 {e_gc:.2e}, top-1 equal: '
              f'{(logits.argmax(1).cpu().numpy() == idx[:, 0]).all()}')
        assert e_l < tol_l and e_gc < tol_g and torch.isfinite(g).all()


def test_fp16_storage_attack_loops(hip, golden_dir):
    """BASELINE.json configs[4] in small: the SPAA loop and PerC_AL.adversary_projector (VGG-16) in fp16-storage mode, first
    iteration from identical state vs the fp32 oracle; losses / dE2000 / norms are fp32 in this mode and must agree
    tightly, the step direction to fp16 accuracy."""
    from spaa_amd.perc_al import PerC_AL
    A, M = hip['attack'], hip['models']
    z = load(golden_dir, 'spaa_64_near')
    sd, pc, clf, oclf, scene, setup = _setup_case(hip, z)
    targets = [int(t) for t in z['targets']]
    tr = []
    so.spaa(sd, oclf, targets, True, scene, float(z['d_thr']), str(z['stealth']), setup, iters=1, trace=tr)
    st = A.AttackState(pc, clf, targets, scene, str(z['stealth']), setup, DEV, storage='f16')
    st.iteration(True, float(z['d_thr']), 2, 1, 0.9)
    sts = st.stats.cpu().numpy()
    assert np.allclose(sts[:, 1], tr[0]['caml2'], rtol=2e-2) and np.allclose(sts[:, 2], tr[0]['camdE'], rtol=2e-2)
    assert (st.state[:, 3].cpu().numpy() == tr[0]['top1']).all()
    x1, ref = M.to_nchw(st.x).cpu(), torch.from_numpy(tr[0]['prj_adv'])
    e_step = rel_l2(x1 - 0.5, ref - 0.5)
    step_len = (x1 - 0.5).flatten(1).norm(dim=1)
    print(f'fp16-storage SPAA first iteration: step rel L2 vs fp32 oracle {e_step:.2e}; image rel Linf {rel_inf(x1, ref):.2e}')
    assert e_step < 0.3 and torch.allclose(step_len, torch.full_like(step_len, 2.0), rtol=1e-3) and torch.isfinite(x1).all()
    for _ in range(3):
        st.iteration(True, float(z['d_thr']), 2, 1, 0.9)
    cam, prj = st.results()
    assert torch.isfinite(cam).all() and torch.isfinite(prj).all()
    # PerC-AL + VGG-16 (the configs[4] pairing), 224-style geometry scaled to what the oracle finishes quickly
    csd = syn.vgg16_state_dict(3, logit_gain=5.0, fc_width=256)
    vclf = hip['clf'].Classifier('vgg16', DEV, state_dict=csd)
    ov = so.OracleClassifier('vgg16', csd)
    sc = syn.scenes(1, 1, (256, 256)).expand(2, -1, -1, -1).contiguous()
    _, _, idx = ov(sc[:1], (240, 240))
    labels = torch.tensor([int(i) for i in idx[0, 1:3]])
    otr = []
    so.perc_al_adversary_projector(ov, sc, labels, 2.0, True, (240, 240), 400, 1., 0.5, 0, stop_after=1, trace=otr)
    ptr = []
    att = PerC_AL(device=DEV, max_iterations=4, alpha_l_init=1, alpha_c_init=0.5, confidence=0, storage='f16')
    out = att.adversary_projector(vclf, sc, labels, None, 2.0, True, (240, 240), trace=ptr)
    e_d = rel_l2(ptr[0][2], otr[0]['delta'])
    print(f'fp16-storage PerC-AL + VGG-16 at 256x256: delta rel L2 after iteration 0 vs fp32 oracle {e_d:.2e}')
    assert e_d < 0.3 and out.min() >= 0 and out.max() <= 1 and (torch.round(out * 255) / 255 - out).abs().max() < 1e-6
    assert np.allclose(ptr[0][1][:, 3].cpu().numpy(), otr[0]['color_dis'].numpy(), rtol=5e-2)


F16_MEASURED = {'value': 0.0, 'near': 0.0, 'tie': 0.0}


@pytest.mark.parametrize('body', ['resnet18', 'vgg16', 'inception_v3'])
def test_fp16_storage_first_iteration_gate_aware(hip, body):
    """fp16-storage mode against the fp32 oracle, decomposed like the fp32 path (tests/gates.py): every ReLU / clamp / arg-max
    gate on which the two disagree sits within fp16 rounding of its threshold (near_zero), the activations agree to fp16
    accuracy (value_tol), and with the ORACLE's gates in the HIP backward the produced projector image agrees to `image` --
    what is left of the plain step error (0.08-0.14 relative L2 of the step, test_fp16_storage_attack_loops) is gate flips of
    units that fp16 rounding moves across zero.  Tolerances = 3 x the largest value measured (printed; profiles/r03_parity.txt)."""
    # measured (profiles/r03_parity.txt): value 1.86e-3, near 9.6e-4, tie 8.2e-4, camera image 1.7e-4, target logit 1.4e-2,
    # projector image with the oracle's gates 3.5e-4 (ResNet-18) / 2.1e-3 (VGG-16: thirteen fp16 layers deep) / 2.4e-3 (Inception-v3)
    tol = dict(near_zero=2.9e-3, value_tol=5.6e-3, cam=5.2e-4, logit=4.3e-2,
               image={'resnet18': 1.1e-3, 'vgg16': 6.2e-3, 'inception_v3': 7.3e-3}[body], measured=F16_MEASURED)
    if body == 'resnet18':
        st = _first_iteration_gate_aware(hip, body, syn.resnet18_state_dict(2, logit_gain=20.0), (64, 64), (64, 64), (60, 60),
                                         [204, 291, 7], 3, storage='f16', tol=tol)
    elif body == 'inception_v3':
        st = _first_iteration_gate_aware(hip, body, syn.inception_v3_state_dict(2, logit_gain=20.0), (107, 107), (128, 128),
                                         (120, 120), [204, 291], 12, storage='f16', tol=tol)
    else:   # (fp16-storage VGG-16: 224 x 224 input, the configs[4] geometry)
        st = _first_iteration_gate_aware(hip, body, syn.vgg16_state_dict(3, logit_gain=5.0, fc_width=256), (224, 224), (256, 256),
                                         (240, 240), [204, 291], 11, storage='f16', tol=tol)
    print(f'fp16 storage, {body}: projector image rel Linf plain {st.errs["plain"].tolist()}, with the oracle\'s gates '
          f'{st.errs["oracle_gates"].tolist()}; gates differing {st.flips.tolist()}; largest so far {F16_MEASURED}')
    assert (st.errs['oracle_gates'] <= st.errs['plain'] + 1e-6).all()


def test_perc_al_vgg16_f16_full_batch_properties(hip):
    """BASELINE.json configs[4]'s per-GPU workload at full size: batch 64, 256x256, VGG-16, PerC_AL.adversary_projector loop
    body in fp16 storage (fp32 dE2000).  Size-independent properties: samples independent (a sub-batch of 8 reproduces its
    rows), run-to-run bitwise reproducible, the box and 8-bit constraints hold, every step has the prescribed length."""
    from spaa_amd.perc_al import PerC_AL, PerCALState
    csd = syn.vgg16_state_dict(2, logit_gain=20.0)
    clf = hip['clf'].Classifier('vgg16', DEV, state_dict=csd)
    scenes = syn.scenes(11, 8, (256, 256)).repeat_interleave(8, dim=0)
    labels = torch.tensor((syn.IMAGENET10_TARGETS[:8]) * 8)
    att = PerC_AL(device=DEV, max_iterations=400, alpha_l_init=1, alpha_c_init=0.5, confidence=0, storage='f16')

    def run(sc, lb, iters):
        with torch.cuda.device(DEV):
            st = PerCALState(att, clf, sc, lb, 5.0, True, (240, 240))
        deltas = []
        for i in range(iters):
            st.iteration(i)
            deltas.append(st.delta.clone())
        return st, deltas

    st, d = run(scenes, labels, 3)
    # iteration 0: delta = alpha_l g / ||g|| (nobody is adversarial yet), then the box clamp: length <= alpha_l(0) = 1
    n0 = d[0][..., :3].flatten(1).norm(dim=1).cpu()
    assert (n0 <= 1.0 + 1e-4).all() and (n0 > 0.5).all(), n0
    x = (st.x_in + st.delta)[..., :3]
    assert x.min() >= -1e-6 and x.max() <= 1 + 1e-6                                  # (inputs + delta) in the box (:211)
    out = st.result()
    assert out.shape == (64, 3, 256, 256) and torch.isfinite(out).all()
    xr = hip['models'].to_nchw(st.x_round)                                         # the quantised image of the last iteration (:212)
    assert (torch.round(xr * 255) / 255 - xr).abs().max() < 1e-6 and xr.min() >= 0 and xr.max() <= 1
    changed = (out != scenes.to(DEV)).flatten(1).any(dim=1)                         # (never adversarial: the input is kept, :165)
    if changed.any():
        assert (torch.round(out[changed] * 255) / 255 - out[changed]).abs().max() < 1e-6
    st2, d2 = run(scenes, labels, 3)
    assert all(torch.equal(a, b) for a, b in zip(d, d2))                           # bitwise reproducible
    # samples are independent, stated EXACTLY: rows 8..15 keep their images and labels while the other 56 rows of the batch get
    # different ones (same batch size = same kernels, tiles and K ranges): their deltas must be bitwise what they were
    other = scenes.roll(24, dims=0).flip(-1).contiguous()
    other[8:16] = scenes[8:16]
    lab2 = labels.roll(3).clone()
    lab2[8:16] = labels[8:16]
    assert not torch.equal(other[:8], scenes[:8])
    _, dmix = run(other, lab2, 2)
    assert all(torch.equal(a[8:16], b[8:16]) for a, b in zip(d, dmix)), 'rows 8..15 depend on the rest of the batch'
    assert not torch.equal(dmix[0][:8], d[0][:8])
    st8, d8 = run(scenes[8:16].contiguous(), labels[8:16], 1)
    e8 = rel_l2(d8[0], d[0][8:16])
    print(f'PerC-AL + VGG-16, fp16 storage, B=64 at 256x256: iteration-0 step lengths {float(n0.min()):.4f}..{float(n0.max()):.4f}; '
          f'rows 8..15 bitwise independent of the other 56 rows; sub-batch of 8 vs rows 8..15 of the batch of 64: delta rel L2 {e8:.2e}')
    # (ANOTHER BATCH SIZE takes other kernel forms -- batch 64: the patch-staged fp16 kernel, batch 8: its canvas / K-range forms and
    # the implicit-GEMM tile -- which sum the same products in another order; fp16-rounded activations then land on the other side of
    # a ReLU gate here and there, and 16 layers amplify that to the mode's own noise floor: the SAME batch in fp16 against fp32 storage
    # differs by 0.135 in this statistic, fp32 storage by 3e-3 between the batch sizes.  Bisected in round 6 (tools/lab/vgg_f16_bisect.py
    # -> profiles/r06_vgg_f16_bisect.txt): 0.046 in rounds 3-4 (one kernel family at both batch sizes), 0.098 with round 5's
    # batch-size-dependent forms, 0.137 at the end of round 5 when the fp16-operand first layer (tile 76) ran at batch 64 only --
    # fixed in round 6 (it runs at every batch size): 0.0686 now, 0.0136 without the canvas / K-range form that only the batch of 8
    # takes, 0.0 (bitwise) with one kernel family and no K ranges at both sizes.  The bound is 1.3 x the measured value; what a dependence between samples
    # would look like is asserted exactly above, and bounded here: hardly an element may move by a tenth of the largest step)
    diff = (d8[0] - d[0][8:16]).abs()
    far = float((diff > 0.1 * d[0].abs().max()).float().mean())
    print(f'    elements further apart than 10 % of the largest |delta|: {far:.2e}; mean |difference| / mean |delta| {float(diff.mean() / d[0].abs().mean()):.3f}')
    assert e8 < 0.09 and far < 1e-4 and torch.isfinite(st.stats).all() and torch.isfinite(st8.stats).all()


# ---------------------------------------------------------------------------------------------------------------
# the PCNet training step (train_network.py:235-363, compute_loss :367-392)
def test_tapconv_weight_gradients(hip):
    """spaa_tapconv_wgrad (weight and bias gradients, exact fp32 MFMA, fixed-order pixel chunks) vs torch.autograd:
    convolutions (stride 1/2, 1x1/3x3/7x7, thin channel counts), transposed convolutions (k2 and k3, unfolded classes)."""
    cp = hip['cp']
    torch.manual_seed(3)
    for ci, co, k, s, h, w, b in [(32, 64, 3, 2, 22, 18, 3), (3, 32, 3, 2, 16, 20, 2), (64, 3, 3, 1, 12, 14, 2),
                                  (6, 32, 3, 2, 16, 16, 2), (128, 256, 3, 1, 9, 8, 2), (32, 64, 1, 1, 10, 10, 2), (3, 3, 1, 1, 9, 7, 2)]:
        x = torch.randn(b, ci, h, w)
        wt = (torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5).requires_grad_(True)
        bias = torch.randn(co, requires_grad=True)
        y = F.conv2d(x, wt, bias, s, k // 2)
        gy = torch.randn_like(y)
        y.backward(gy)
        builder = lambda wv, s=s, k=k: cp.conv_fwd_plan(wv, None, s, k // 2, 'cpu')
        pl = builder(wt.detach())
        pl.weights, pl.taps, pl.w_split = pl.weights.to(DEV), pl.taps.to(DEV), None
        cp.attach_maps(pl, builder, wt.detach())
        for nchunk in (None, 1, 7):
            dw, db = pl.wgrad(nhwc(x, pl.cin_p).to(DEV), nhwc(gy, (co + 3) // 4 * 4).to(DEV), nchunk=nchunk)
            assert rel_inf(pl.unpack_grad(dw), wt.grad) < 2e-5, (ci, co, k, s, nchunk)
            assert rel_inf(db, bias.grad) < 2e-5, (ci, co, k, s, nchunk)
    for ci, co, k, pad, op, h, w in [(64, 32, 2, 0, 0, 9, 7), (128, 64, 3, 1, 1, 8, 8), (32, 2, 2, 0, 0, 8, 10)]:
        x = torch.randn(2, ci, h, w)
        wt = (torch.randn(ci, co, k, k) / (ci * k * k) ** 0.5).requires_grad_(True)
        bias = torch.randn(co, requires_grad=True)
        y = F.conv_transpose2d(x, wt, bias, 2, pad, op)
        gy = torch.randn_like(y)
        y.backward(gy)
        builder = lambda wv, pad=pad: cp.deconv_fwd_plan(wv, None, 2, pad, 'cpu', fold=False)
        pl = builder(wt.detach())
        pl.weights, pl.taps, pl.w_split = pl.weights.to(DEV), pl.taps.to(DEV), None
        cp.attach_maps(pl, builder, wt.detach())
        dw, db = pl.wgrad(nhwc(x).to(DEV), nhwc(gy, (co + 3) // 4 * 4).to(DEV))
        assert rel_inf(pl.unpack_grad(dw), wt.grad) < 2e-5, ('deconv', ci, co, k)
        assert rel_inf(db, bias.grad) < 2e-5


def test_training_loss_and_gradient(hip):
    """compute_loss 'l1' / 'l1+ssim' (train_network.py:367-392; SSIM with replicate padding, pytorch_ssim/__init__.py:26-58):
    value and gradient w.r.t. the inferred image vs the oracle, on sizes that are not multiples of the 16x16 tile."""
    lib, M = hip['lib'], hip['models']
    from spaa_amd import train_network as tn
    torch.manual_seed(8)
    for (b, h, w) in [(2, 37, 53), (3, 64, 64), (1, 16, 12)]:
        t = syn.scenes(3, b, (h, w))
        y = (t + 0.08 * torch.randn(b, 3, h, w)).clamp(0, 1)
        for opt in ('l1', 'l1+ssim'):
            yc = y.clone().requires_grad_(True)
            loss, l2 = so.compute_loss(yc, t, opt)
            loss.backward()
            y4, t4 = M.to_nhwc4(y.to(DEV)), M.to_nhwc4(t.to(DEV))
            nblk = ((h + 15) // 16) * ((w + 15) // 16)
            ws = [torch.zeros_like(y4) for _ in range(4)]
            part = torch.zeros(b * nblk, 3, device=DEV)
            lib.call('spaa_train_loss_fwd_bwd', lib.ptr(y4), lib.ptr(t4), lib.ptr(tn._window().to(DEV)), 1.0,
                     1.0 if 'ssim' in opt else 0.0, lib.ptr(ws[0]), lib.ptr(ws[1]), lib.ptr(ws[2]), lib.ptr(part), lib.ptr(ws[3]), b, h, w)
            s = part.sum(dim=0).cpu() / (3.0 * b * h * w)
            val = float(s[1]) + ((1 - float(s[0])) if 'ssim' in opt else 0.0)
            assert abs(val - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss.detach()))) and abs(float(s[2]) - float(l2)) < 1e-6
            assert rel_inf(M.to_nchw(ws[3]), yc.grad) < 1e-4, (b, h, w, opt)
            got, _ = tn.compute_loss(y.to(DEV), t, opt)
            assert abs(float(got) - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss.detach())))


def test_pcnet_training_step(hip):
    """Two iterations of the reference's PCNet training loop body (forward with the current WarpingNet parameters, l1+ssim
    resp. l1 loss, gradients of all 44 parameter tensors, three Adam optimisers) on HIP vs the oracle (torch.autograd +
    torch.optim on the CPU): loss values, every gradient, every updated parameter."""
    from spaa_amd.train_network import PCNetTrainer
    sz = (64, 64)
    sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='rect')
    pc = make_pcnet(hip, sd, sz)
    scene = syn.scenes(1, 1, sz)
    B = 4
    prj = [syn.scenes(20 + i, B, sz) for i in range(2)]
    cam = [syn.scenes(30 + i, B, sz) * 0.8 + 0.05 for i in range(2)]
    orc = so.PCNetTrainOracle(sd, scene, B)
    tr = PCNetTrainer(pc, scene, B, device=DEV)
    import math
    adam_m, adam_v = {}, {}
    for it, opt in enumerate(('l1+ssim', 'l1')):
        p_before = {n: v.detach().cpu().clone() for n, v in pc.named_parameters()}
        lo, l2o = orc.step(prj[it], cam[it], opt)
        lh, l2h = tr.step(prj[it], cam[it], opt)
        assert abs(lh - lo) < 2e-5 * max(1.0, abs(lo)) and abs(l2h - l2o) < 1e-6, (it, lh, lo)
        worst = ('', 0.0)
        for name, g_ref in orc.grads.items():
            g = tr.grads[name].reshape(g_ref.shape)
            e = rel_l2(g, g_ref)
            if e > worst[1]:
                worst = (name, e)
            # (ReLU gates within rounding of zero: sparse differences; a wrong kernel gives O(1))
            assert e < 2e-3, (it, name, e)
        print(f'training step {it} ({opt}): loss {lh:.6f} vs oracle {lo:.6f}; worst gradient rel L2 {worst[1]:.2e} ({worst[0]})')
        # the optimiser step itself: torch.optim.Adam semantics applied to the HIP gradients (the gradients were compared above;
        # comparing parameters directly would amplify rounding noise, Adam's first steps are ~lr * sign(g))
        hp = dict(pc.named_parameters())
        for name in orc.p:
            lr = 1e-2 if name in ('warping_net.affine_mat', 'warping_net.theta') else (5e-3 if 'grid_refine_net' in name else 1e-3)
            wd = 1e-4 if 'warping_net' not in name else 0.0
            g = tr.grads[name].reshape(p_before[name].shape).cpu().double() + wd * p_before[name].double()
            adam_m[name] = 0.9 * adam_m.get(name, 0.0) + 0.1 * g
            adam_v[name] = 0.999 * adam_v.get(name, 0.0) + 0.001 * g * g
            t_ = it + 1
            want = p_before[name].double() - (lr / (1 - 0.9 ** t_)) * adam_m[name] / (adam_v[name].sqrt() / math.sqrt(1 - 0.999 ** t_) + 1e-8)
            got = hp[name].detach().cpu().double()
            assert float((got - want).abs().max()) < 1e-6 + 2e-3 * lr, (it, name, float((got - want).abs().max()))
            # and the oracle's parameters where the gradient is well above rounding noise
            sig = orc.grads[name].abs() > 1e-2 * orc.grads[name].abs().max()
            assert float((hp[name].detach().cpu() - orc.p[name].detach())[sig].abs().max()) < 0.05 * lr, (it, name)
    assert tr.iters == 2


def test_training_iteration_vs_reference_fixture(hip, golden_dir):
    """The HIP training step against what the REFERENCE's own modules produced (tests/golden/make_golden.py gen_train):
    first-iteration loss and gradients."""
    from spaa_amd.train_network import PCNetTrainer
    z = load(golden_dir, 'train_32')
    sz, bsz, seed = tuple(int(v) for v in z['sz']), int(z['bsz']), int(z['seed'])
    sd = syn.pcnet_state_dict(seed, cam_sz=sz, mask='rect')
    pc = make_pcnet(hip, sd, sz)
    tr = PCNetTrainer(pc, syn.scenes(seed + 1, 1, sz), bsz, l2_reg=1e-4, lr_drop_ratio=0.2, device=DEV)
    lh, l2h = tr.step(syn.scenes(seed + 20, bsz, sz), syn.scenes(seed + 30, bsz, sz) * 0.8 + 0.05, 'l1+ssim')
    assert abs(lh - float(z['loss0'])) < 2e-5 and abs(l2h - float(z['l2_0'])) < 1e-6
    names = [str(n) for n in z['names']]
    gn = np.array([float(tr.grads[k].double().norm()) for k in names])
    assert np.allclose(gn, z['gradnorm0'], rtol=2e-3), np.abs(gn / z['gradnorm0'] - 1).max()
    for key in z.files:
        if key.startswith('grad0.'):
            k = key[len('grad0.'):]
            assert rel_l2(tr.grads[k].reshape(z[key].shape), torch.from_numpy(z[key])) < 2e-3, key


# ---------------------------------------------------------------------------------------------------------------
def test_rccl_path_single_rank():
    """The multi-GPU path's RCCL side on the one-GPU box (the 8-GPU curve is the driver's to run): a fresh child process
    initialises the `nccl` process group on the device, runs `spaa_sharded` THROUGH its all_gather (world size 1) and
    bench.py's `gather_final` / `reduce_times`, and checks that the gathered results are the attack's, in order.  A second
    child runs bench.py itself with SPAA_BENCH_FORCE_DIST=1 (process-group init, barriers around the timed region,
    preallocated gather) for two steps."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'nccl_child.py')], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and 'RCCL_SINGLE_RANK_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-modes'],
                       capture_output=True, text=True, timeout=600, env=dict(env, SPAA_BENCH_FORCE_DIST='1'))
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['gather_ms'] is not None and line['gather_ms'] > 0
    print(f"bench.py under the nccl process group (1 rank): {line['value']} it/s, gather {line['gather_ms']} ms")
