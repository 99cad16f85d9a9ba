"""PCNet training step on HIP.

Mirrors `train_pcnet` of /root/reference/src/python/train_network.py:235-363 and `compute_loss` :367-392: one iteration =
forward of PCNet (WarpingNet with its CURRENT parameters: the sampling grid is rebuilt every step, models.py:163-185) ->
l1 [+ (1 - SSIM)] loss -> gradients of all 44 parameter tensors -> three Adam optimisers (affine/TPS lr 1e-2, grid-refine
net lr 5e-3, ShadingNet lr 1e-3 with L2 weight decay `l2_reg`) with MultiStepLR(milestones 100 / 1200 / 1800, gamma
`lr_drop_ratio`) -> `l1` only for the first 400 iterations, then `l1+ssim` (:300-303).

Everything arithmetic runs in libspaa_hip.so:
  forward / input gradients      PCNetEngine (tapconv kernels; the packed weights are refreshed on the device each step)
  weight / bias gradients        spaa_tapconv_wgrad          (csrc/tapconv_wgrad.hip)
  loss + its gradient            spaa_train_loss_fwd_bwd     (csrc/color.hip)
  grid / affine / TPS gradients  spaa_warp_bwd_grid, spaa_warp_finish_grid_bwd, spaa_warp_coarse_grid_bwd (csrc/train_ops.hip)
  optimiser                      spaa_adam_step
PyTorch supplies device memory and index plumbing (re-packing a changed parameter into the kernels' layout through
precomputed index maps).  No CPU fallback.
"""
import math
import random

import torch

from . import _lib
from . import convplan as cp
from .models import PCNet, PCNetEngine, to_nhwc4, to_nchw, C_ptr, transposed_taps

# (parameter module name, forward builder, input-gradient builder) exactly as PCNetEngine builds them (models.py)
_SHADING = {
    'conv1': (2, 1), 'conv2': (2, 1), 'conv3': (1, 1), 'conv4': (1, 1), 'conv5': (1, 1), 'conv1_s': (2, 1),
    'conv2_s': (2, 1), 'conv3_s': (1, 1), 'conv4_s': (1, 1), 'conv6': (1, 1), 'skipConv3': (1, 1), 'skipConv2': (1, 0)}


def _window(window_size=11, sigma=1.5):
    g = torch.Tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float().reshape(-1).contiguous()


class PCNetTrainer:
    """State of one PCNet training run: engine, per-parameter Adam moments, schedules.  `step(prj_batch, cam_batch)` is one
    iteration of the reference's loop body (train_network.py:293-357)."""

    def __init__(self, pcnet, cam_scene, batch_size, l2_reg=1e-4, lr_drop_ratio=0.2, device='cuda'):
        if not isinstance(pcnet, PCNet):
            raise TypeError('PCNetTrainer needs a spaa_amd.PCNet')
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise RuntimeError('spaa_amd training runs on the GPU only (no CPU fallback)')
        self.dev, self.pc, self.B = dev, pcnet, batch_size
        self.l2_reg, self.gamma = float(l2_reg), float(lr_drop_ratio)
        wn, sn = pcnet.warping_net, pcnet.shading_net
        if not wn.with_refine:
            raise NotImplementedError('training covers the SPAA configuration (WarpingNet with the grid-refine net)')
        if not pcnet.use_rough:
            raise NotImplementedError('training covers the SPAA configuration (use_rough=True)')
        with _lib.on_device(dev):
            s = cam_scene.detach().float().to(dev)
            while s.ndim < 4:
                s = s[None]
            self.scene4 = to_nhwc4(s.expand(batch_size, -1, -1, -1).contiguous())
            self.Hc, self.Wc = wn.out_size
            self.window = _window().to(dev)
            self._build(wn, sn)
        self.iters = 0

    # ------------------------------------------------------------------------------------------------------------
    def _build(self, wn, sn):
        dev, B = self.dev, self.B
        self.eng = None   # created at the first step (needs the projector size)
        # parameter groups of train_network.py:247-256
        self.params = dict(self.pc.named_parameters())
        self.groups = {
            'w1': dict(names=['warping_net.affine_mat', 'warping_net.theta'], lr=1e-2, wd=0.0, milestone=100),
            'w2': dict(names=[n for n in self.params if 'warping_net.grid_refine_net' in n], lr=5e-3, wd=0.0, milestone=1200),
            's': dict(names=[n for n in self.params if 'warping_net' not in n], lr=1e-3, wd=self.l2_reg, milestone=1800)}
        self.m = {n: torch.zeros_like(p, device=dev) for n, p in self.params.items()}
        self.v = {n: torch.zeros_like(p, device=dev) for n, p in self.params.items()}
        self.grads = {}

    def _make_engine(self, prj_size):
        """PCNetEngine + index maps to refresh its packed weights, wgrad plans (unfolded geometry), refine-net plans."""
        wn, sn = self.pc.warping_net, self.pc.shading_net
        dev = self.dev
        eng = PCNetEngine(self.pc, self.B, prj_size, fuse_skip2=False)
        eng.fuse_tail = False   # the weight gradients of conv6 / transConv2 read X7 and its gradient
        self.maps = []   # (plan, parameter, bias parameter or None)

        def reg(plan, builder, mod, with_bias):
            cp.attach_maps(plan, builder, mod.weight.detach().cpu())
            self.maps.append((plan, mod.weight, mod.bias if with_bias else None))

        for nm, (st, pad) in _SHADING.items():
            mod = getattr(sn, nm)
            reg(eng.f[nm], lambda w, st=st, pad=pad: cp.conv_fwd_plan(w, None, st, pad, 'cpu'), mod, True)
            in_ch = (3, 6) if nm == 'conv1_s' else None
            reg(eng.d[nm], lambda w, st=st, pad=pad, in_ch=in_ch: cp.conv_dgrad_plan(w, st, pad, 'cpu', in_ch=in_ch), mod, False)
        for nm, pad in (('transConv1', 1), ('transConv2', 0)):
            mod = getattr(sn, nm)
            reg(eng.f[nm], lambda w, pad=pad: cp.deconv_fwd_plan(w, None, 2, pad, 'cpu'), mod, True)
            reg(eng.d[nm], lambda w, pad=pad: cp.deconv_dgrad_plan(w, 2, pad, 'cpu'), mod, False)
        for key, i, pad in (('skip1a', 0, 0), ('skip1b', 2, 1), ('skip1c', 4, 1)):
            reg(eng.f[key], lambda w, pad=pad: cp.conv_fwd_plan(w, None, 1, pad, 'cpu'), sn.skipConv1[i], True)
        # weight-gradient plans: only the geometry matters (taps / classes / packing layout), built once; transposed
        # convolutions unfolded (one weight matrix per output-parity class)
        self.wg = {}
        for nm, (st, pad) in _SHADING.items():
            self.wg[nm] = self._wg_plan(getattr(sn, nm), lambda w, st=st, pad=pad: cp.conv_fwd_plan(w, None, st, pad, 'cpu'))
        for nm, pad in (('transConv1', 1), ('transConv2', 0)):
            self.wg[nm] = self._wg_plan(getattr(sn, nm), lambda w, pad=pad: cp.deconv_fwd_plan(w, None, 2, pad, 'cpu', fold=False))
        for key, i, pad in (('skip1a', 0, 0), ('skip1b', 2, 1), ('skip1c', 4, 1)):
            self.wg[key] = self._wg_plan(sn.skipConv1[i], lambda w, pad=pad: cp.conv_fwd_plan(w, None, 1, pad, 'cpu'))
        # input-gradient plans of the two inner skipConv1 layers (the attack never needs them: skipConv1 sees the scene only)
        self.skip_d = {}
        for key, i in (('skip1b', 2), ('skip1c', 4)):
            mod = sn.skipConv1[i]
            pl = cp.conv_dgrad_plan(mod.weight, 1, 1, dev, key + '_dgrad')
            reg(pl, lambda w: cp.conv_dgrad_plan(w, 1, 1, 'cpu'), mod, False)
            self.skip_d[key] = pl
        # grid-refine net (models.py:123-134): forward, input-gradient and weight-gradient plans
        g = wn.grid_refine_net
        self.rf, self.rd, self.rwg = {}, {}, {}
        for i, kind in ((0, 'conv'), (2, 'conv'), (4, 'deconv'), (6, 'deconv')):
            mod = g[i]
            if kind == 'conv':
                fb = lambda w: cp.conv_fwd_plan(w, None, 2, 1, 'cpu')
                db = lambda w: cp.conv_dgrad_plan(w, 2, 1, 'cpu')
                self.rf[i] = cp.conv_fwd_plan(mod.weight, mod.bias, 2, 1, dev, f'refine{i}')
                self.rd[i] = cp.conv_dgrad_plan(mod.weight, 2, 1, dev, f'refine{i}_dgrad')
                wb = fb
            else:
                fb = lambda w: cp.deconv_fwd_plan(w, None, 2, 0, 'cpu')
                db = lambda w: cp.deconv_dgrad_plan(w, 2, 0, 'cpu')
                self.rf[i] = cp.deconv_fwd_plan(mod.weight, mod.bias, 2, 0, dev, f'refine{i}')
                self.rd[i] = cp.deconv_dgrad_plan(mod.weight, 2, 0, dev, f'refine{i}_dgrad')
                wb = lambda w: cp.deconv_fwd_plan(w, None, 2, 0, 'cpu', fold=False)
            reg(self.rf[i], fb, mod, True)
            reg(self.rd[i], db, mod, False)
            self.rwg[i] = self._wg_plan(mod, wb)
        H, W = self.Hc, self.Wc

        def z(*shape):
            return torch.zeros(*shape, device=dev)

        self.grid_ws = dict(coarse=z(1, H, W, 4), r0=z(1, H // 2, W // 2, 32), r2=z(1, H // 4, W // 4, 64),
                            r4=z(1, H // 2, W // 2, 32), refine=z(1, H, W, 4), fine=z(H, W, 4),
                            g_fine=z(H, W, 4), g_sum=z(1, H, W, 4), g_r6=z(1, H, W, 4), g_r4=z(1, H // 2, W // 2, 32),
                            g_r2=z(1, H // 4, W // 4, 64), g_r0=z(1, H // 2, W // 2, 32), g_c0=z(1, H, W, 4),
                            partial=z(((H * W + 255) // 256) * (6 + 2 * (wn.nctrl + 2))), g_params=z(6 + 2 * (wn.nctrl + 2)))
        B = self.B
        self.t0, self.t1 = z(B, H, W, 4), z(B, H, W, 4)              # skipConv1 intermediates (kept for its gradients)
        self.g_r1, self.g_t1, self.g_t0 = z(B, H, W, 4), z(B, H, W, 4), z(B, H, W, 4)
        nblk = ((H + 15) // 16) * ((W + 15) // 16)
        self.loss_ws = dict(mmu=z(B, H, W, 4), m11=z(B, H, W, 4), m12=z(B, H, W, 4), partial=z(B * nblk, 3), gY=z(B, H, W, 4),
                            gP=z(B, H, W, 4))
        self.ones_state = torch.ones(B, 4, dtype=torch.int32, device=dev)
        self.eng = eng

    def _wg_plan(self, mod, builder):
        """A plan used for its geometry only (taps, classes, packing layout): tap list on the device, unpack map attached."""
        w = mod.weight.detach().cpu()
        pl = builder(w)
        pl.weights, pl.taps, pl.w_split = pl.weights.to(self.dev), pl.taps.to(self.dev), None
        cp.attach_maps(pl, builder, w)
        return pl

    # ------------------------------------------------------------------------------------------------------------
    def _refresh_weights(self):
        for plan, w, b in self.maps:
            plan.refresh(w, b)

    def _build_grid(self, prj_size):
        """models.py:168-178 with the current parameters, keeping every intermediate for the backward pass."""
        wn, ws = self.pc.warping_net, self.grid_ws
        hi, wi = prj_size
        H, W = self.Hc, self.Wc
        self._aff = wn.affine_mat.detach().float().contiguous().view(-1)
        self._theta = wn.theta.detach().float().contiguous().view(-1)
        self._ctrl = wn.ctrl_pts.detach().float().contiguous().view(-1)
        _lib.call('spaa_warp_coarse_grid', _lib.ptr(self._aff), _lib.ptr(self._theta), _lib.ptr(self._ctrl), wn.nctrl, hi, wi,
                  H, W, _lib.ptr(ws['coarse']))
        R, L = _lib.ACT_RELU, _lib.ACT_LEAKY01
        self.rf[0].run(ws['coarse'], ws['r0'], act=R)
        self.rf[2].run(ws['r0'], ws['r2'], act=R)
        self.rf[4].run(ws['r2'], ws['r4'], act=R)
        self.rf[6].run(ws['r4'], ws['refine'], act=L)
        _lib.call('spaa_warp_finish_grid', _lib.ptr(ws['coarse']), _lib.ptr(ws['refine']), _lib.ptr(ws['fine']), H * W)
        eng = self.eng
        eng.grid = ws['fine']
        eng.tap_off, eng.tap_order, eng.tap_wm, eng.tap_src = transposed_taps(eng.grid, prj_size, (H, W), eng.mask, want_table=True)
        eng.tiled = None   # (the grid changes every step: the per-tile boxes of the LDS-staged gather are not rebuilt)

    def _set_scene(self):
        """PCNetEngine.set_scene, keeping the skipConv1 intermediates."""
        eng = self.eng
        eng.version += 1
        eng.scene = self.scene4
        R = _lib.ACT_RELU
        eng.f['skip1a'].run(self.scene4, self.t0, act=R)
        eng.f['skip1b'].run(self.t0, self.t1, act=R)
        eng.f['skip1c'].run(self.t1, eng.a['R1'], act=R)

    # ------------------------------------------------------------------------------------------------------------
    def step(self, prj_batch, cam_batch, loss=None):
        """One training iteration (train_network.py:293-357).  Returns (loss value, l2 (MSE) value) as Python floats — the one
        host sync of the step, which the reference also has (`.item()` :307,346)."""
        if loss is None:
            loss = 'l1' if self.iters <= 400 else 'l1+ssim'                     # :300-303
        if loss == '':
            raise TypeError('Loss type not specified')                            # compute_loss :368-369
        if 'l2' in loss or 'huber' in loss:
            raise NotImplementedError("spaa_amd training implements the reference's PCNet losses 'l1' and 'l1+ssim'")
        with _lib.on_device(self.dev):
            return self._step(prj_batch, cam_batch, loss)

    def _step(self, prj_batch, cam_batch, loss):
        p = _lib.ptr
        B, H, W = self.B, self.Hc, self.Wc
        x4 = to_nhwc4(prj_batch.to(self.dev))
        t4 = to_nhwc4(cam_batch.to(self.dev))
        if self.eng is None:
            self._make_engine(tuple(prj_batch.shape[-2:]))
        eng = self.eng
        prj_size = (eng.Hp, eng.Wp)
        # ---- forward with the current parameters
        self._refresh_weights()
        self._build_grid(prj_size)
        self._set_scene()
        y4 = eng.forward(x4, clamp01=False)                                      # model(prj, scene) :306
        # ---- loss and its gradient w.r.t. the inferred image (compute_loss :367-392)
        lw = self.loss_ws
        l1_w, ssim_w = (1.0 if 'l1' in loss else 0.0), (1.0 if 'ssim' in loss else 0.0)
        _lib.call('spaa_train_loss_fwd_bwd', p(y4), p(t4), p(self.window), l1_w, ssim_w, p(lw['mmu']), p(lw['m11']), p(lw['m12']),
                  p(lw['partial']), p(lw['gY']), B, H, W)
        # clamp / ReLU gate of the output layer: gradient w.r.t. conv6's pre-activation
        _lib.call('spaa_select_grad', p(lw['gY']), p(lw['gY']), p(self.ones_state), p(eng.a['Ypre']), p(lw['gP']), B, H * W)
        # ---- backward: input gradients (fills every layer's pre-activation gradient), then weight gradients
        eng.backward(lw['gP'], input_grad=False)   # (no gradient w.r.t. the projector image: it is data here)
        a, g = eng.a, eng.g
        gr = self.grads
        wplan = self.wg

        def wgrad(name, pname, inp, gout, in_coff=0):
            dw, db = wplan[name].wgrad(inp, gout, in_coff=in_coff)
            gr[pname + '.weight'] = wplan[name].unpack_grad(dw)
            gr[pname + '.bias'] = db

        sp = 'shading_net.'
        wgrad('conv6', sp + 'conv6', a['X7'], lw['gP'])
        wgrad('transConv2', sp + 'transConv2', a['X6'], g['P7'])
        wgrad('transConv1', sp + 'transConv1', a['X5'], g['P6'])
        wgrad('conv5', sp + 'conv5', a['X4'], g['P5'])
        wgrad('conv4', sp + 'conv4', a['X3'], g['P4'])
        wgrad('conv3', sp + 'conv3', a['X2'], g['P3'])
        wgrad('conv2', sp + 'conv2', a['X1'], g['P2'])
        wgrad('conv1', sp + 'conv1', a['xw'], g['P1'])
        wgrad('skipConv3', sp + 'skipConv3', a['X2'], g['P5'])
        wgrad('skipConv2', sp + 'skipConv2', a['X1'], g['P6'])
        wgrad('conv4_s', sp + 'conv4_s', a['S3'], g['S4'])
        wgrad('conv3_s', sp + 'conv3_s', a['S2'], g['S3'])
        wgrad('conv2_s', sp + 'conv2_s', a['S1'], g['S2'])
        wgrad('conv1_s', sp + 'conv1_s', a['cat8'], g['S1'])
        # skipConv1 (on the scene; its output is added to conv6's pre-activation, models.py:291,301)
        _lib.call('spaa_relu_gate', p(lw['gP']), p(a['R1']), p(self.g_r1), lw['gP'].numel())     # ReLU after skipConv1.4
        wgrad('skip1c', sp + 'skipConv1.4', self.t1, self.g_r1)
        self.skip_d['skip1c'].run(self.g_r1, self.g_t1, gate=self.t1)
        wgrad('skip1b', sp + 'skipConv1.2', self.t0, self.g_t1)
        self.skip_d['skip1b'].run(self.g_t1, self.g_t0, gate=self.t0)
        wgrad('skip1a', sp + 'skipConv1.0', self.scene4, self.g_t0)
        # ---- WarpingNet: grid gradient (summed over the batch), refine net, TPS / affine parameters
        ws = self.grid_ws
        _lib.call('spaa_warp_bwd_grid', p(g['xw']), p(x4), p(eng.grid), p(eng.mask), p(ws['g_fine']), B, eng.Hp, eng.Wp, H, W)
        _lib.call('spaa_warp_finish_grid_bwd', p(ws['g_fine']), p(ws['coarse']), p(ws['refine']), p(ws['g_sum']), p(ws['g_r6']),
                  H * W)
        wp = 'warping_net.grid_refine_net.'

        def rwgrad(i, inp, gout):
            dw, db = self.rwg[i].wgrad(inp, gout)
            gr[wp + f'{i}.weight'] = self.rwg[i].unpack_grad(dw)
            gr[wp + f'{i}.bias'] = db

        rwgrad(6, ws['r4'], ws['g_r6'])
        self.rd[6].run(ws['g_r6'], ws['g_r4'], gate=ws['r4'])
        rwgrad(4, ws['r2'], ws['g_r4'])
        self.rd[4].run(ws['g_r4'], ws['g_r2'], gate=ws['r2'])
        rwgrad(2, ws['r0'], ws['g_r2'])
        self.rd[2].run(ws['g_r2'], ws['g_r0'], gate=ws['r0'])
        rwgrad(0, ws['coarse'], ws['g_r0'])
        self.rd[0].run(ws['g_r0'], ws['g_c0'], add=ws['g_sum'])               # + the skip connection (models.py:176)
        wn = self.pc.warping_net
        _lib.call('spaa_warp_coarse_grid_bwd', p(ws['g_c0']), p(self._aff), p(self._theta), p(self._ctrl), wn.nctrl, eng.Hp,
                  eng.Wp, H, W, p(ws['partial']), p(ws['g_params']))
        gr['warping_net.affine_mat'] = ws['g_params'][:6].view(1, 2, 3)
        gr['warping_net.theta'] = ws['g_params'][6:].view(1, wn.nctrl + 2, 2)
        # ---- optimiser steps (:318-320) and schedulers (:354-356)
        self.iters += 1
        for gname, grp in self.groups.items():
            lr = grp['lr'] * (self.gamma if (self.iters - 1) >= grp['milestone'] else 1.0)
            for n in grp['names']:
                prm = self.params[n]
                gt = gr[n].contiguous()
                assert gt.numel() == prm.numel(), n
                _lib.call('spaa_adam_step', p(prm.data.view(-1)), p(gt.view(-1)), p(self.m[n].view(-1)), p(self.v[n].view(-1)),
                          prm.numel(), lr, 0.9, 0.999, 1e-8, grp['wd'], self.iters)
        self.pc.invalidate()
        part = lw['partial'].sum(dim=0).cpu()
        n_el = 3.0 * B * H * W
        l1, l2 = float(part[1]) / n_el, float(part[2]) / n_el
        total = l1_w * l1 + ssim_w * (1.0 - float(part[0]) / n_el)
        return total, l2


def compute_loss(prj_infer, prj_train, loss_option):
    """train_network.py:367-392 on HIP for the options PCNet training uses ('l1', 'l1+ssim'): returns (train_loss, l2_loss)
    as 0-dim tensors (no gradient: `PCNetTrainer.step` carries the gradient path)."""
    if loss_option == '':
        raise TypeError('Loss type not specified')
    dev = prj_infer.device
    with _lib.on_device(dev):
        y4, t4 = to_nhwc4(prj_infer), to_nhwc4(prj_train.to(dev))
        b, h, w, _ = y4.shape
        nblk = ((h + 15) // 16) * ((w + 15) // 16)
        ws = [torch.zeros_like(y4) for _ in range(4)]
        part = torch.zeros(b * nblk, 3, device=dev)
        _lib.call('spaa_train_loss_fwd_bwd', _lib.ptr(y4), _lib.ptr(t4), _lib.ptr(_window().to(dev)), 1.0, 1.0, _lib.ptr(ws[0]),
                  _lib.ptr(ws[1]), _lib.ptr(ws[2]), _lib.ptr(part), _lib.ptr(ws[3]), b, h, w)
        s = part.sum(dim=0) / (3.0 * b * h * w)
        loss = torch.zeros((), device=dev)
        if 'l1' in loss_option:
            loss = loss + s[1]
        if 'l2' in loss_option:
            loss = loss + s[2]
        if 'ssim' in loss_option:
            loss = loss + (1 - s[0])
        return loss, s[2]


def train_pcnet(model, train_data, valid_data, cfg):
    """train_network.py:235-363 (without the visdom plots): `train_data` = dict(cam_scene [1,3,H,W], cam_train, prj_train),
    `cfg` with max_iters, batch_size, num_train, l2_reg, lr_drop_ratio, device.  Returns (model, valid_psnr, valid_rmse,
    valid_ssim) like the reference."""
    get = (lambda k, d=None: cfg[k] if k in cfg else d) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
    dev = torch.device(get('device', 'cuda'))
    tr = PCNetTrainer(model, train_data['cam_scene'], get('batch_size'), get('l2_reg', 1e-4), get('lr_drop_ratio', 0.2), dev)
    cam_train, prj_train = train_data['cam_train'], train_data['prj_train']
    valid_psnr = valid_rmse = valid_ssim = 0.0
    for it in range(get('max_iters')):
        idx = random.sample(range(get('num_train')), get('batch_size'))            # :295
        loss, l2 = tr.step(prj_train[idx], cam_train[idx])
        if get('verbose', False) and (it % 50 == 0 or it == get('max_iters') - 1):
            print(f'Iter:{it:5d} | Train Loss: {loss:.4f} | Train RMSE: {math.sqrt(l2 * 3):.4f}')
    if valid_data is not None:
        from . import metrics
        with torch.no_grad():
            infer = model(valid_data['prj_valid'].to(dev), train_data['cam_scene'].to(dev).expand(valid_data['prj_valid'].shape[0], -1, -1, -1))
        valid_psnr, valid_rmse, valid_ssim = (metrics.psnr(infer, valid_data['cam_valid']), metrics.rmse(infer, valid_data['cam_valid']),
                                              metrics.ssim(infer, valid_data['cam_valid']))
    return model, valid_psnr, valid_rmse, valid_ssim
