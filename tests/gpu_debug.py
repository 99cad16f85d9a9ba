"""Diagnostic script for the GPU box (not a pytest file): runs every HIP component against the oracle / torch and
prints max differences.  Usage on the GPU box:  python tests/gpu_debug.py [stage ...]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import spaa_oracle as so  # noqa: E402
from spaa_amd import synthetic as syn, convplan as cp, _lib  # noqa: E402
from spaa_amd.models import PCNet, WarpingNet, to_nhwc4, to_nchw  # noqa: E402
from spaa_amd.classifier import Classifier, ClassifierEngine  # noqa: E402
from spaa_amd.projector_based_attack import spaa  # noqa: E402
from spaa_amd import differential_color_functions as dcf  # noqa: E402
from tapconv_emu import nhwc, nchw  # noqa: E402

G = os.path.join(ROOT, 'tests', 'golden')
dev = 'cuda'


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def stage_tapconv():
    torch.manual_seed(0)
    cases = [(3, 32, 3, 2, 1, 32, 32), (6, 32, 3, 2, 1, 16, 24), (32, 64, 3, 2, 1, 32, 32), (64, 128, 3, 1, 1, 16, 16),
             (128, 256, 3, 1, 1, 16, 16), (256, 128, 3, 1, 1, 16, 16), (32, 64, 1, 1, 0, 24, 24), (32, 3, 3, 1, 1, 32, 32),
             (3, 64, 7, 2, 3, 56, 56), (64, 128, 1, 2, 0, 28, 28), (512, 512, 3, 1, 1, 7, 7)]
    for ci, co, k, s, p, h, w in cases:
        x = torch.randn(3, ci, h, w, requires_grad=True)
        wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
        b = torch.randn(co)
        y = F.conv2d(x, wt, b, s, p)
        plan = cp.conv_fwd_plan(wt, b, s, p, dev)
        out = torch.zeros(3, y.shape[2], y.shape[3], ((co + 3) // 4) * 4, device=dev)
        plan.run(nhwc(x.detach(), plan.cin_p).to(dev), out)
        e1 = rel(nchw(out.cpu(), co), y)
        gy = torch.randn_like(y)
        y.backward(gy)
        dplan = cp.conv_dgrad_plan(wt, s, p, dev)
        gx = torch.zeros(3, h, w, ((ci + 3) // 4) * 4, device=dev)
        dplan.run(nhwc(gy, dplan.cin_p).to(dev), gx)
        e2 = rel(nchw(gx.cpu(), ci), x.grad)
        print(f'  conv ci={ci} co={co} k={k} s={s} {h}x{w}: fwd {e1:.2e} dgrad {e2:.2e}')
    for ci, co, k, p, op, h, w in [(128, 64, 3, 1, 1, 16, 16), (64, 32, 2, 0, 0, 16, 16), (32, 2, 2, 0, 0, 8, 8)]:
        x = torch.randn(2, ci, h, w, requires_grad=True)
        wt = torch.randn(ci, co, k, k) / (ci * k * k) ** 0.5
        b = torch.randn(co)
        y = F.conv_transpose2d(x, wt, b, 2, p, op)
        plan = cp.deconv_fwd_plan(wt, b, 2, p, dev)
        out = torch.zeros(2, y.shape[2], y.shape[3], ((co + 3) // 4) * 4, device=dev)
        plan.run(nhwc(x.detach(), plan.cin_p).to(dev), out)
        e1 = rel(nchw(out.cpu(), co), y)
        gy = torch.randn_like(y)
        y.backward(gy)
        dplan = cp.deconv_dgrad_plan(wt, 2, p, dev)
        gx = torch.zeros(2, h, w, ci, device=dev)
        dplan.run(nhwc(gy, dplan.cin_p).to(dev), gx)
        e2 = rel(nchw(gx.cpu(), ci), x.grad)
        print(f'  deconv ci={ci} co={co} k={k}: fwd {e1:.2e} dgrad {e2:.2e}')
    # epilogue: add + relu + gate
    x = torch.randn(2, 32, 12, 12)
    wt = torch.randn(64, 32, 3, 3) / 17
    b = torch.randn(64)
    add = torch.randn(2, 64, 12, 12)
    gate = torch.randn(2, 64, 12, 12)
    plan = cp.conv_fwd_plan(wt, b, 1, 1, dev)
    out = torch.zeros(2, 12, 12, 64, device=dev)
    plan.run(nhwc(x).to(dev), out, add=nhwc(add).to(dev), act=_lib.ACT_RELU, gate=nhwc(gate).to(dev))
    ref = F.relu(F.conv2d(x, wt, b, 1, 1) + add) * (gate > 0)
    print(f'  epilogue add+relu+gate: {rel(nchw(out.cpu()), ref):.2e}')


def stage_color():
    z = np.load(os.path.join(G, 'color_kat.npz'))
    a, b = torch.from_numpy(z['rgb_a']).to(dev), torch.from_numpy(z['rgb_b']).to(dev)
    lab_a = dcf.rgb2lab_diff(a)
    lab_b = dcf.rgb2lab_diff(b)
    print(f'  lab_a {rel(lab_a, torch.from_numpy(z["lab_a"])):.2e} lab_b {rel(lab_b, torch.from_numpy(z["lab_b"])):.2e}')
    de = dcf.ciede2000_diff(lab_a, lab_b)
    print(f'  dE map {rel(de, torch.from_numpy(z["de"])):.2e} (max abs {(de.cpu() - torch.from_numpy(z["de"])).abs().max():.2e})')
    l2, dE, g = dcf.stealth_loss_with_grad(a, b, 0.0, 1.0)
    g_ref = torch.from_numpy(z['grad_a'])
    fin = torch.isfinite(g_ref)
    ours_at_nan = g.cpu()[~fin]
    print(f'  reference grad has {int((~fin).sum())} non-finite entries (near-grey dark pixels); ours there: finite='
          f'{int(torch.isfinite(ours_at_nan).sum())}')
    gd = torch.where(fin, g.cpu() * (a.shape[2] * a.shape[3]) - g_ref, torch.zeros_like(g_ref))
    g_ref = torch.where(fin, g_ref, torch.zeros_like(g_ref))
    print(f'  dE grad rel {gd.abs().max() / g_ref.abs().max():.2e}; worst idx {np.unravel_index(gd.abs().argmax(), gd.shape)}'
          f' ref max {g_ref.abs().max():.3e}')
    print(f'  dE mean {dE.cpu().numpy()} vs {z["de"].mean(axis=(1, 2))}')
    s = torch.tensor([50., 2.6772, -79.7751]).view(1, 3, 1, 1).to(dev)
    s2 = torch.tensor([50., 0., -82.7485]).view(1, 3, 1, 1).to(dev)
    print(f'  Sharma pair {dcf.ciede2000_diff(s, s2).item():.4f} (reference constant -> {float(z["sharma"]):.4f})')


def load_pcnet(sd, cam_sz):
    pc = PCNet(sd['mask'], WarpingNet(out_size=tuple(cam_sz)))
    pc.load_state_dict(sd)
    return pc.to(dev)


def stage_pcnet():
    for name in ('pcnet_64', 'pcnet_nonsq', 'pcnet_256'):
        z = np.load(os.path.join(G, name + '.npz'))
        cam_sz = tuple(int(v) for v in z['cam_sz'])
        sd = syn.pcnet_state_dict(int(z['seed']), cam_sz=cam_sz, mask=str(z['mask']))
        pc = load_pcnet(sd, cam_sz)
        x = torch.from_numpy(z['x']).to(dev).requires_grad_(True)
        s = torch.from_numpy(z['s']).to(dev)
        eng = pc.engine(x.shape[0], x.shape[-2:])
        fg = torch.from_numpy(z['fine_grid'])[0]
        print(f'  {name}: fine grid maxabs diff {(eng.grid[..., :2].cpu() - fg).abs().max():.2e}')
        y = pc(x, s)
        print(f'  {name}: fwd rel {rel(y, torch.from_numpy(z["y"])):.2e}')
        (y * torch.from_numpy(z['r']).to(dev)).sum().backward()
        print(f'  {name}: grad rel {rel(x.grad, torch.from_numpy(z["grad_x"])):.2e}')
        if name == 'pcnet_256':
            import torch.nn.functional as F
            w_ = so._sd(sd, 'warping_net.')
            xc = torch.from_numpy(z['x'])
            ca = F.affine_grid(w_['affine_mat'], torch.Size([1, 3, *xc.shape[-2:]]), align_corners=True).permute(0, 3, 1, 2)
            ct = so.tps_grid(w_['theta'], w_['ctrl_pts'], (1, 3) + cam_sz)
            coarse_ref = F.grid_sample(ca, ct, align_corners=True)[0].permute(1, 2, 0)
            wn = pc.warping_net
            coarse = torch.zeros(1, *cam_sz, 4, device=dev)
            _lib.call('spaa_warp_coarse_grid', _lib.ptr(wn.affine_mat.detach().view(-1).contiguous()),
                      _lib.ptr(wn.theta.detach().view(-1).contiguous()), _lib.ptr(wn.ctrl_pts.view(-1).contiguous()), 36,
                      xc.shape[-2], xc.shape[-1], cam_sz[0], cam_sz[1], _lib.ptr(coarse))
            dcoarse = (coarse[0, ..., :2].cpu() - coarse_ref).abs()
            print(f'   coarse grid diff max {dcoarse.max():.2e} at {np.unravel_index(dcoarse.argmax(), dcoarse.shape)}; '
                  f'fine diff argmax {np.unravel_index((eng.grid[..., :2].cpu() - fg).abs().argmax(), fg.shape)}')
            # backward per stage vs oracle autograd
            xc = xc.clone().requires_grad_(True)
            sc = torch.from_numpy(z['s'])
            xw = (so.warp(sd, xc, cam_sz) * sd['mask'])
            xw.retain_grad()
            yy, acts = so.shading_net(sd, xw, (sc, xw * sc), return_all=True)
            for v in acts.values():
                if v.requires_grad:
                    v.retain_grad()
            (yy * torch.from_numpy(z['r'])).sum().backward()
            print(f'   g wrt xw total: ours {rel(nchw((eng.g["xw"] + eng.g["xs"] * eng.scene).cpu(), 3), xw.grad):.2e}')
            for k, v in dict(x7='P7', x6='P6', x5='P5', x4='P4', x3='P3', x2='P2', x1='P1').items():
                ref_g = acts[k].grad * (acts[k] > 0)
                print(f'   gP {v}: {rel(nchw(eng.g[v].cpu()), ref_g):.2e}', end=';')
            print()
            gx_ref = torch.from_numpy(z['grad_x'])
            dd = (x.grad.cpu() - gx_ref).abs()
            print(f'   grad_x diff max {dd.max():.3e} at {np.unravel_index(dd.argmax(), dd.shape)} ref max {gx_ref.abs().max():.3e}'
                  f' mean abs diff {dd.mean():.3e} ref mean abs {gx_ref.abs().mean():.3e}')
        if name == 'pcnet_64':  # per-layer check against the oracle
            xc, sc = torch.from_numpy(z['x']), torch.from_numpy(z['s'])
            xw = so.warp(sd, xc, cam_sz) * sd['mask']
            _, acts = so.shading_net(sd, xw, (sc, xw * sc), return_all=True)
            m = dict(res1_s='S1', res2_s='S2', res3_s='S3', res4_s='S4', res1='R1', x1='X1', res2='R2', x2='X2', res3='R3',
                     x3='X3', x4='X4', x5='X5', x6='X6', x7='X7')
            print('   xw', f'{rel(nchw(eng.a["xw"].cpu(), 3), xw):.1e}', end=' ')
            for k, v in m.items():
                t = eng.a[v].cpu()
                print(v, f'{rel(nchw(t, acts[k].shape[1]), acts[k]):.1e}', end=' ')
            print()


def stage_classifier():
    csd = syn.resnet18_state_dict(2, logit_gain=20.0)
    for (h, crop, insz, b) in [(64, (60, 60), (56, 56), 3), (256, (240, 240), (224, 224), 2)]:
        torch.manual_seed(1)
        im = torch.rand(b, 3, h, h, requires_grad=True)
        oc = so.OracleClassifier('resnet18', csd, input_sz=insz)
        raw, p, idx = oc(im, crop)
        r = torch.randn(b, 1000)
        (raw * r).sum().backward()
        clf = Classifier('resnet18', dev, state_dict=csd, input_sz=insz)
        im2 = im.detach().clone().to(dev).requires_grad_(True)
        raw2, p2, idx2 = clf(im2, crop)
        (raw2 * r.to(dev)).sum().backward()
        print(f'  resnet18 {h}px: logits rel {rel(raw2, raw):.2e} grad rel {rel(im2.grad, im.grad):.2e} '
              f'top1 {idx[:, 0]} vs {idx2[:, 0]} p1 {p[:, 0]} vs {p2[:, 0]}')


def stage_spaa(names=None):
    names = names or ['spaa_64_untargeted', 'spaa_64_imagenet10', 'spaa_64_near', 'spaa_64_caml2_dthr', 'spaa_64_prjl2',
                      'spaa_64_camdE', 'spaa_256_untargeted', 'spaa_256_near']
    for name in names:
        z = np.load(os.path.join(G, name + '.npz'))
        sz = tuple(int(v) for v in z['sz'])
        sd = syn.pcnet_state_dict(int(z['seed']), cam_sz=sz, mask=str(z['mask']))
        pc = load_pcnet(sd, sz)
        csd = syn.resnet18_state_dict(2, logit_gain=float(z['gain']))
        clf = Classifier('resnet18', dev, state_dict=csd, input_sz=tuple(int(v) for v in z['input_sz']))
        scene = syn.scenes(int(z['scene_seed']), 1, sz)
        setup = dict(classifier_crop_sz=tuple(int(v) for v in z['crop']), prj_brightness=0.5, prj_im_sz=sz)
        tr = []
        t0 = time.time()
        cam, prj = spaa(pc, clf, None, [int(t) for t in z['targets']], bool(z['targeted']), scene, float(z['d_thr']),
                        str(z['stealth']), dev, setup, trace=tr)
        torch.cuda.synchronize()
        dt = time.time() - t0
        st = torch.stack([t[0] for t in tr]).cpu().numpy()
        fs = torch.stack([t[1] for t in tr]).cpu().numpy()
        k = z['prj_adv_best'].shape[0]
        succ_ok = (st[:, :, 0] == z['succ']).all(axis=1)
        badv_ok = (st[:, :, 1] == z['best_adv']).all(axis=1)
        first_bad = int(np.argmin(succ_ok & badv_ok)) if not (succ_ok & badv_ok).all() else -1
        print(f'  {name}: prj rel {rel(prj[:k], torch.from_numpy(z["prj_adv_best"])):.2e} cam rel '
              f'{rel(cam[:k], torch.from_numpy(z["cam_infer_best"])):.2e} | masks equal: {bool((succ_ok & badv_ok).all())} '
              f'(first mismatch it {first_bad}) | caml2 it0 rel {np.abs(fs[0, :, 1] - z["caml2"][0]).max():.1e} '
              f'camdE it0 {np.abs(fs[0, :, 2] - z["camdE"][0]).max():.1e} p1 it0 {np.abs(fs[0, :, 0] - z["p1"][0]).max():.1e}'
              f' | {dt:.2f}s')
        if first_bad >= 0:
            i = first_bad
            print(f'     it {i}: p1 {fs[i, :, 0]} ref {z["p1"][i]}  top1 {st[i, :, 3]} ref {z["top1"][i]} caml2*255 '
                  f'{fs[i, :, 1] * 255} ref {z["caml2"][i] * 255}')


STAGES = dict(tapconv=stage_tapconv, color=stage_color, pcnet=stage_pcnet, classifier=stage_classifier, spaa=stage_spaa)

if __name__ == '__main__':
    which = sys.argv[1:] or list(STAGES)
    print(_lib.load().spaa_version().decode(), torch.cuda.get_device_name(0))
    for s in which:
        print(f'[{s}]')
        t0 = time.time()
        STAGES[s]()
        torch.cuda.synchronize()
        print(f'[{s}] done in {time.time() - t0:.1f}s')
