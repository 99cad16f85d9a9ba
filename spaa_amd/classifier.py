"""Frozen ImageNet classifier on HIP kernels behind the reference's `Classifier` contract.

Mirrors /root/reference/src/python/classifier.py:12-75: `Classifier(model_name, device, device_ids, fix_params,
sort_results)` and `classifier(im, crop_sz) -> (raw_score Tensor[B,1000], p_sorted ndarray, idx ndarray)`.
The network bodies are torchvision's (third-party, not in the reference tree; pinned torchvision==0.15.1): the
architecture is restated here as a layer table over the tap-list convolution kernel, with eval-mode BatchNorm
folded into the convolutions.  Pretrained weights cannot be downloaded in this environment (the reference fetches
them by URL, classifier.py:24-36): pass `state_dict=` (torchvision key names) or `weights_path=`.

`ClassifierEngine` is what the fused attack loop drives: forward = crop + area-resize + normalise -> net -> logits;
backward = input gradient only (all parameters frozen, classifier.py:41-44).
"""
import os
import weakref

import numpy as np
import torch

from . import _lib
from . import convplan as cp
from .models import to_nhwc4, to_nchw, USE_GATE_MASKS

FUSE_POOL = os.environ.get('SPAA_FUSE_POOL', '1') != '0'   # VGG-16, fp16 storage: the 2 x 2 max-pools in the epilogue of the convolution before them
FOLD_S2_F16 = int(os.environ.get('SPAA_FOLD_S2_F16', '0'))   # fp16 storage: ResNet's stride-2 input gradients with the four parity classes folded into N: 0 never (default: neutral in the loop, 238.7-239.3 it/s either way), 1 layer2.0 (49.5 -> 38 us per launch), 2 all three
# 1: ResNet-18's max-pool adjoint as the prologue of the stem's input gradient.  Measured SLOWER (profiles/r05_frontend.txt: stem_dgrad
# 190 -> 304 us for the 55 us launch it removes -- the patch formed by loads + VALU work in four dependent round trips per channel block
# where the LDS-DMA of the separate form costs no issue slots): off by default, kept with its bitwise test.
FUSE_POOL_ADJOINT = os.environ.get('SPAA_FUSE_POOL_ADJOINT', '0') == '1'
BODY_GATE_MASKS = os.environ.get('SPAA_BODY_MASKS', '1') != '0'   # 0: VGG-16 / Inception-v3 gate with the activation itself (A/B measurements)

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
INPUT_SZ = {'resnet18': (224, 224), 'vgg16': (224, 224), 'inception_v3': (299, 299)}


def center_crop_origin(h, w, size):
    """img_proc.py:126-132."""
    th, tw = size
    return int(round((h - th) / 2.)), int(round((w - tw) / 2.))


def _strip(sd):
    out = {}
    for k, v in sd.items():
        while k.startswith('module.'):
            k = k[len('module.'):]
        out[k] = v
    return out


class ResNet18Body:
    """torchvision.models.resnet18 (eval) forward + input-gradient on tapconv/maxpool/avgpool kernels."""

    def __init__(self, sd, batch, in_hw, dev, storage='f32'):
        sd = _strip(sd)
        self.B, self.dev, self.storage = batch, dev, storage
        hd = torch.float16 if storage == 'f16' else torch.float32   # activations / gradients; input image, features, logits fp32
        h, w = in_hw

        def folded(conv, bn):
            return cp.fold_bn(sd[conv + '.weight'], sd[bn + '.weight'], sd[bn + '.bias'], sd[bn + '.running_mean'],
                              sd[bn + '.running_var'])

        def zf(*shape):
            return torch.zeros(*shape, device=dev)

        def z(*shape):
            return torch.zeros(*shape, device=dev, dtype=hd)

        def osz(n, k, s, p):
            return (n + 2 * p - k) // s + 1

        wgt, b = folded('conv1', 'bn1')
        self.stem_f = cp.conv_fwd_plan(wgt, b, 2, 3, dev, 'stem')
        self.stem_d = cp.conv_dgrad_plan(wgt, 2, 3, dev, 'stem_dgrad')
        self.in_hw = (h, w)
        h1, w1 = osz(h, 7, 2, 3), osz(w, 7, 2, 3)
        self.c1 = z(batch, h1, w1, 64)
        h2, w2 = osz(h1, 3, 2, 1), osz(w1, 3, 2, 1)
        self.mp = z(batch, h2, w2, 64)
        self.mp_arg = torch.zeros(batch, h2, w2, 64, dtype=torch.uint8, device=dev)
        self.blocks = []
        cin, hh, ww = 64, h2, w2
        x_buf = self.mp
        for li, cout in enumerate((64, 128, 256, 512), start=1):
            for bi in range(2):
                p = f'layer{li}.{bi}'
                stride = 2 if (li > 1 and bi == 0) else 1
                ho, wo = osz(hh, 3, stride, 1), osz(ww, 3, stride, 1)
                blk = dict(name=p, stride=stride, x=x_buf)
                w1_, b1_ = folded(p + '.conv1', p + '.bn1')
                w2_, b2_ = folded(p + '.conv2', p + '.bn2')
                blk['f1'] = cp.conv_fwd_plan(w1_, b1_, stride, 1, dev, p + '.conv1')
                blk['d1'] = cp.conv_dgrad_plan(w1_, stride, 1, dev, p + '.conv1_dgrad',
                                               fold=True if (FOLD_S2_F16 and storage == 'f16' and stride == 2 and (cin <= 64 or FOLD_S2_F16 > 1)) else None)
                if stride == 2 and cin in (32, 64) and storage == 'f32' and batch * hh * ww >= 100000:
                    # layer2.0.conv1's input gradient (128 -> 64 channels, 28^2 -> 56^2) at benchmark batches: the patch-staged
                    # stride-2 kernel with its four parity classes in one launch (tools/lab/x6p_resnet.py: 79 -> 71 us)
                    d74 = cp.conv_dgrad_plan(w1_, stride, 1, dev, p + '.conv1_dgrad', fold=False)
                    if d74.x6p_ok():
                        d74.fixed_tile = 74
                        blk['d1'] = d74
                blk['f2'] = cp.conv_fwd_plan(w2_, b2_, 1, 1, dev, p + '.conv2')
                blk['d2'] = cp.conv_dgrad_plan(w2_, 1, 1, dev, p + '.conv2_dgrad')
                if p + '.downsample.0.weight' in sd:
                    wd, bd = folded(p + '.downsample.0', p + '.downsample.1')
                    blk['fd'] = cp.conv_fwd_plan(wd, bd, stride, 0, dev, p + '.downsample')
                    blk['dd'] = cp.conv_dgrad_plan(wd, stride, 0, dev, p + '.downsample_dgrad')
                    blk['idt'] = z(batch, ho, wo, cout)
                    blk['g_t'] = z(batch, hh, ww, cin)
                blk['o1'] = z(batch, ho, wo, cout)
                blk['out'] = z(batch, ho, wo, cout)
                # ReLU gates as byte masks (1 byte per 4 channels), written by the forward epilogues
                blk['m_o1'] = torch.zeros(batch, ho, wo, cout // 4, dtype=torch.uint8, device=dev)
                blk['m_out'] = torch.zeros(batch, ho, wo, cout // 4, dtype=torch.uint8, device=dev)
                blk['g_o1'] = z(batch, ho, wo, cout)
                blk['g_x'] = z(batch, hh, ww, cin)
                self.blocks.append(blk)
                x_buf, cin, hh, ww = blk['out'], cout, ho, wo
        self.feat_hw = hh * ww
        self.feat = zf(batch, 1, 1, 512)
        ncls = sd['fc.weight'].shape[0]
        self.ncls = ncls
        self.fc_f = cp.linear_fwd_plan(sd['fc.weight'], sd['fc.bias'], dev, 'fc')
        self.fc_d = cp.linear_dgrad_plan(sd['fc.weight'], dev, 'fc_dgrad')
        self.logits = zf(batch, 1, 1, ncls)
        self.g_feat = zf(batch, 1, 1, 512)
        self.g_last = z(batch, hh, ww, 512)
        self.g_c1 = z(batch, h1, w1, 64)
        self.g_in = zf(batch, h, w, 4)
        self.write_masks = True

    def forward(self, x4):
        R = _lib.ACT_RELU
        B = self.B
        self.stem_f.run(x4, self.c1, act=R)
        _, h1, w1, _ = self.c1.shape
        _, h2, w2, _ = self.mp.shape
        h16 = self.storage == 'f16'
        if h16:
            _lib.call('spaa_maxpool_fwd_f16', _lib.hptr(self.c1), _lib.hptr(self.mp), _lib.ptr(self.mp_arg), B, h1, w1, 64,
                      h2, w2, 3, 2, 1, 64, 0)
        else:
            _lib.call('spaa_maxpool3s2_fwd', _lib.ptr(self.c1), _lib.ptr(self.mp), _lib.ptr(self.mp_arg), B, h1, w1, 64,
                      h2, w2)
        masks = (USE_GATE_MASKS or h16) and getattr(self, 'write_masks', True)   # (write_masks False: a forward pass nobody differentiates)
        for blk in self.blocks:
            blk['f1'].run(blk['x'], blk['o1'], act=R, mask_out=blk['m_o1'] if masks else None)
            if 'fd' in blk:
                blk['fd'].run(blk['x'], blk['idt'])
                idt = blk['idt']
            else:
                idt = blk['x']
            blk['f2'].run(blk['o1'], blk['out'], add=idt, act=R, mask_out=blk['m_out'] if masks else None)
        last = self.blocks[-1]['out']
        if h16:
            _lib.call('spaa_avgpool_fwd_f16', _lib.hptr(last), _lib.ptr(self.feat), B, self.feat_hw, 512)
        else:
            _lib.call('spaa_avgpool_fwd', _lib.ptr(last), _lib.ptr(self.feat), B, self.feat_hw, 512)
        self.fc_f.run(self.feat, self.logits)
        return self.logits.view(B, self.ncls)

    def backward(self, g_logits):
        """g_logits [B,ncls] -> gradient w.r.t. the normalised input [B,h,w,4]."""
        B = self.B
        self.fc_d.run(g_logits.view(B, 1, 1, self.ncls), self.g_feat)
        last = self.blocks[-1]['out']
        h16 = self.storage == 'f16'
        if h16:
            _lib.call('spaa_avgpool_bwd_f16', _lib.ptr(self.g_feat), _lib.hptr(last), _lib.hptr(self.g_last), B,
                      self.feat_hw, 512)
        else:
            _lib.call('spaa_avgpool_bwd', _lib.ptr(self.g_feat), _lib.ptr(last), _lib.ptr(self.g_last), B, self.feat_hw,
                      512)
        gP = self.g_last
        for i in range(len(self.blocks) - 1, -1, -1):
            blk = self.blocks[i]
            # the block's input is the previous block's output; block 0's is the max-pool output (gated in maxpool_bwd)
            if USE_GATE_MASKS or h16:
                kw2, kw1 = dict(gate_bits=blk['m_o1']), dict(gate_bits=self.blocks[i - 1]['m_out'] if i > 0 else None)
            else:
                kw2, kw1 = dict(gate=blk['o1']), dict(gate=blk['x'] if i > 0 else None)
            blk['d2'].run(gP, blk['g_o1'], **kw2)
            if 'dd' in blk:
                blk['dd'].run(gP, blk['g_t'])
                blk['d1'].run(blk['g_o1'], blk['g_x'], add=blk['g_t'], **kw1)
            else:
                blk['d1'].run(blk['g_o1'], blk['g_x'], add=gP, **kw1)
            gP = blk['g_x']
        _, h1, w1, _ = self.c1.shape
        _, h2, w2, _ = self.mp.shape
        if h16:
            _lib.call('spaa_maxpool_bwd_f16', _lib.hptr(gP), _lib.ptr(self.mp_arg), 1, _lib.hptr(self.g_c1), B, h1, w1, 64,
                      h2, w2, 3, 2, 1, 64, 0)
        elif FUSE_POOL_ADJOINT and gP.is_contiguous() and gP.shape[3] == 64:
            # the pool's adjoint as the prologue of the stem's input gradient (csrc/tapconv_thinmf.hip, POOL): g_c1 -- 205 MB at batch
            # 64 -- is neither written nor read (measured slower, see FUSE_POOL_ADJOINT)
            self.stem_d.run(gP, self.g_in, pool_adjoint=(self.mp_arg, (h1, w1), True))
            return self.g_in
        else:
            _lib.call('spaa_maxpool3s2_bwd', _lib.ptr(gP), _lib.ptr(self.mp_arg), 1, _lib.ptr(self.g_c1),
                      B, h1, w1, 64, h2, w2)
        self.stem_d.run(self.g_c1, self.g_in)
        return self.g_in

    def refresh_masks(self):
        """Recompute the gate masks from the activation buffers (after a test has overwritten the activations)."""
        for blk in self.blocks:
            blk['m_o1'].copy_(_lib.pack_gate_mask(blk['o1'].float()))
            blk['m_out'].copy_(_lib.pack_gate_mask(blk['out'].float()))

    def flops_fwd(self):
        h, w = self.in_hw
        t = self.stem_f.flops(self.B, *self.c1.shape[1:3])
        for blk in self.blocks:
            ho, wo = blk['out'].shape[1:3]
            t += blk['f1'].flops(self.B, ho, wo) + blk['f2'].flops(self.B, ho, wo)
            if 'fd' in blk:
                t += blk['fd'].flops(self.B, ho, wo)
        return t + self.fc_f.flops(self.B, 1, 1)


VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


class VGG16Body:
    """torchvision.models.vgg16 (eval: dropout is the identity) forward + input-gradient."""

    def __init__(self, sd, batch, in_hw, dev, storage='f32'):
        sd = _strip(sd)
        self.B, self.dev, self.storage = batch, dev, storage
        hd = torch.float16 if storage == 'f16' else torch.float32   # activations / gradients; input image and logits fp32
        h, w = in_hw
        self.in_hw = (h, w)

        def zf(*shape):
            return torch.zeros(*shape, device=dev)

        def z(*shape):
            return torch.zeros(*shape, device=dev, dtype=hd)

        self.ops = []
        cin, idx = 3, 0
        for v in VGG16_CFG:
            if v == 'M':
                ho, wo = h // 2, w // 2
                self.ops.append(dict(kind='pool', hin=h, win=w, c=cin, out=z(batch, ho, wo, cin),
                                     arg=torch.zeros(batch, ho, wo, cin, dtype=torch.uint8, device=dev),
                                     g=z(batch, h, w, cin)))
                h, w = ho, wo
                idx += 1
            else:
                wt, bs = sd[f'features.{idx}.weight'], sd[f'features.{idx}.bias']
                self.ops.append(dict(kind='conv', f=cp.conv_fwd_plan(wt, bs, 1, 1, dev, f'features.{idx}'),
                                     d=cp.conv_dgrad_plan(wt, 1, 1, dev, f'features.{idx}_dgrad'),
                                     out=z(batch, h, w, v), g=zf(batch, h, w, 4) if cin == 3 else z(batch, h, w, cin)))
                cin = v
                idx += 2
        self.feat_hw = (h, w)
        # ReLU gates of the conv -> conv transitions as byte masks (1 byte per 4 channels, written by the forward epilogue): the input
        # gradient then reads 2 bits per element instead of the activation and stays on the branch-free epilogue
        # (profiles/r05_configs4_f16s_tapconv_layers.json: features.2_dgrad 689 us against 477 forward with the activation as gate)
        self.masks = BODY_GATE_MASKS and (USE_GATE_MASKS or storage == 'f16')
        self.write_masks = True       # (ClassifierEngine.forward(need_grad=False): a forward pass nobody differentiates skips them)
        # fp16 storage: conv -> ReLU -> 2 x 2 max-pool in ONE launch (csrc/tapconv_h16p.hip POOL); tests that read the convolution's own
        # activation switch it off
        self.fuse_pool = FUSE_POOL and storage == 'f16'
        for i, op in enumerate(self.ops[:-1]):
            if op['kind'] == 'conv' and self.ops[i + 1]['kind'] == 'conv' and self.masks:
                op['m'] = torch.zeros(*op['out'].shape[:3], op['out'].shape[3] // 4, dtype=torch.uint8, device=dev)
        self.pool7 = z(batch, 7, 7, 512)
        self.g_pool7 = z(batch, 7, 7, 512)
        self.g_feat = z(batch, h, w, 512)
        w1 = sd['classifier.0.weight']
        w1p = w1.view(w1.shape[0], 512, 49).permute(0, 2, 1).reshape(w1.shape[0], 49 * 512)  # NCHW flatten -> NHWC
        self.fc = [(cp.linear_fwd_plan(w1p, sd['classifier.0.bias'], dev, 'classifier.0'),
                    cp.linear_dgrad_plan(w1p, dev, 'classifier.0_dgrad')),
                   (cp.linear_fwd_plan(sd['classifier.3.weight'], sd['classifier.3.bias'], dev, 'classifier.3'),
                    cp.linear_dgrad_plan(sd['classifier.3.weight'], dev, 'classifier.3_dgrad')),
                   (cp.linear_fwd_plan(sd['classifier.6.weight'], sd['classifier.6.bias'], dev, 'classifier.6'),
                    cp.linear_dgrad_plan(sd['classifier.6.weight'], dev, 'classifier.6_dgrad'))]
        self.ncls = sd['classifier.6.weight'].shape[0]
        fcw = sd['classifier.0.weight'].shape[0]
        self.h1, self.h2 = z(batch, 1, 1, fcw), z(batch, 1, 1, fcw)
        self.logits = zf(batch, 1, 1, self.ncls)
        self.g_h1, self.g_h2 = z(batch, 1, 1, fcw), z(batch, 1, 1, fcw)

    def forward(self, x4):
        B, R = self.B, _lib.ACT_RELU
        t = x4
        fused_pool = False
        for i, op in enumerate(self.ops):
            if op['kind'] == 'conv':
                nxt = self.ops[i + 1] if i + 1 < len(self.ops) else None
                if self.fuse_pool and nxt is not None and nxt['kind'] == 'pool' and 'm' not in op:
                    # conv -> ReLU -> MaxPool2d(2, 2) as one launch where the patch-staged fp16 kernel serves the layer (its epilogue pools: the
                    # full-size activation is not written; ConvPlan.run falls back to conv + spaa_maxpool_fwd anywhere else)
                    op['f'].run(t, op['out'], act=R, pool=(nxt['out'], nxt['arg'], self.write_masks))
                    fused_pool = True
                else:
                    op['f'].run(t, op['out'], act=R, mask_out=op.get('m') if self.write_masks else None)
            elif fused_pool:
                fused_pool = False       # (pooled by the convolution's launch)
            else:
                _lib.call('spaa_maxpool_fwd_f16' if self.storage == 'f16' else 'spaa_maxpool_fwd', _lib.hptr(t),
                          _lib.hptr(op['out']), _lib.ptr(op['arg']), B, op['hin'], op['win'], op['c'], op['hin'] // 2,
                          op['win'] // 2, 2, 2, 0, op['c'], 0)
            op['inp'] = t
            t = op['out']
        fh, fw = self.feat_hw
        if (fh, fw) != (7, 7):
            if self.storage == 'f16':
                raise NotImplementedError('fp16-storage VGG-16 needs a 224x224 input (7x7 features: no adaptive pooling)')
            _lib.call('spaa_adaptive_avgpool_fwd', _lib.ptr(t), _lib.ptr(self.pool7), B, fh, fw, 512, 7, 7)
            t = self.pool7
        flat = t.view(B, 1, 1, 49 * 512)
        self.fc[0][0].run(flat, self.h1, act=R)
        self.fc[1][0].run(self.h1, self.h2, act=R)
        self.fc[2][0].run(self.h2, self.logits)
        return self.logits.view(B, self.ncls)

    def backward(self, g_logits):
        B = self.B
        self.fc[2][1].run(g_logits.view(B, 1, 1, self.ncls), self.g_h2, gate=self.h2)
        self.fc[1][1].run(self.g_h2, self.g_h1, gate=self.h1)
        self.fc[0][1].run(self.g_h1, self.g_pool7.view(B, 1, 1, 49 * 512))
        fh, fw = self.feat_hw
        g = self.g_pool7
        if (fh, fw) != (7, 7):
            _lib.call('spaa_adaptive_avgpool_bwd', _lib.ptr(g), None, _lib.ptr(self.g_feat), B, fh, fw, 512, 7, 7)
            g = self.g_feat
        # g is the gradient w.r.t. the last pool's output
        pending = None     # (fp16 storage: a pool whose adjoint runs as the prologue of the convolution's input gradient below it)
        for i in range(len(self.ops) - 1, -1, -1):
            op = self.ops[i]
            if op['kind'] == 'pool':
                if self.fuse_pool and i > 0 and self.ops[i - 1]['kind'] == 'conv':
                    pending = (g, op)       # (ConvPlan.run(unpool=...) falls back to spaa_maxpool_bwd where the fused form does not apply)
                    continue
                # input of a pool is a conv+ReLU output: gather + ReLU gate -> gradient w.r.t. that conv's pre-activation
                _lib.call('spaa_maxpool_bwd_f16' if self.storage == 'f16' else 'spaa_maxpool_bwd', _lib.hptr(g),
                          _lib.ptr(op['arg']), 1, _lib.hptr(op['g']), B, op['hin'], op['win'], op['c'], op['hin'] // 2,
                          op['win'] // 2, 2, 2, 0, op['c'], 0)
            else:
                prev = self.ops[i - 1] if i > 0 else None
                kw = {}
                if pending is not None:
                    g, pop = pending
                    kw, pending = dict(unpool=(pop['arg'], pop['g'])), None
                if prev is not None and prev['kind'] == 'conv' and 'm' in prev:
                    op['d'].run(g, op['g'], gate_bits=prev['m'], **kw)
                else:
                    gate = op['inp'] if (prev is not None and prev['kind'] == 'conv') else None
                    op['d'].run(g, op['g'], gate=gate, **kw)
            g = op['g']
        return g

    def refresh_masks(self):
        """Recompute the gate masks from the activation buffers (after a test has overwritten the activations)."""
        for op in self.ops:
            if 'm' in op:
                op['m'].copy_(_lib.pack_gate_mask(op['out'].float()))

    def flops_fwd(self):
        t = 0
        for op in self.ops:
            if op['kind'] == 'conv':
                t += op['f'].flops(self.B, *op['out'].shape[1:3])
        return t + sum(f.flops(self.B, 1, 1) for f, _ in self.fc)


def _inception_body(sd, batch, in_hw, dev, storage='f32'):
    from .inception import InceptionV3Body
    return InceptionV3Body(sd, batch, in_hw, dev, storage)


BODIES = {'resnet18': ResNet18Body, 'vgg16': VGG16Body, 'inception_v3': _inception_body}


class ClassifierEngine:
    """crop -> area resize -> normalise -> net, forward and input-gradient, for a fixed batch/geometry."""

    def __init__(self, name, state_dict, batch, im_hw, crop_sz, input_sz=None, device='cuda', storage='f32'):
        self.storage = storage
        if name not in BODIES:
            raise NotImplementedError(f'classifier body {name!r} is not implemented on HIP yet (have: {list(BODIES)})')
        self.name, self.B, self.dev = name, batch, torch.device(device)
        self.H, self.W = im_hw
        self.ch, self.cw = crop_sz
        self.cy0, self.cx0 = center_crop_origin(self.H, self.W, crop_sz)
        self.oh, self.ow = tuple(input_sz) if input_sz is not None else INPUT_SZ[name]
        self.body = BODIES[name](state_dict, batch, (self.oh, self.ow), self.dev, storage)
        self.pre = torch.zeros(batch, self.oh, self.ow, 4, device=self.dev)
        self.g_y = torch.zeros(batch, self.H, self.W, 4, device=self.dev)
        import ctypes as C
        self._mean = (C.c_float * 3)(*IMAGENET_MEAN)
        self._std = (C.c_float * 3)(*IMAGENET_STD)
        self.ncls = self.body.ncls
        self.owner = None   # weakref to the attack state this engine is leased to (Classifier.engine)
        self.version = 0    # bumped whenever the activation workspaces are overwritten

    def forward(self, y4, need_grad=True):
        """`need_grad=False`: nobody will call backward() on this pass (PerC-AL's second, decision-only forward pass on the quantised
        image, perc_al/__init__.py:220-238): the bodies that write ReLU-gate masks skip them."""
        _lib.check_dev(y4)
        assert y4.shape == (self.B, self.H, self.W, 4)
        self.version += 1
        self._grad_ready = bool(need_grad)
        if hasattr(self.body, 'write_masks'):
            self.body.write_masks = bool(need_grad)
        _lib.call('spaa_preproc_fwd', _lib.ptr(y4), _lib.ptr(self.pre), self.B, self.H, self.W, self.cy0, self.cx0,
                  self.ch, self.cw, self.oh, self.ow, self._mean, self._std)
        return self.body.forward(self.pre)

    def backward(self, g_logits):
        if not getattr(self, '_grad_ready', False):
            # (a need_grad=False pass overwrote the activations but left the ReLU-gate masks / pool arg-max bytes of the pass before it)
            raise RuntimeError('ClassifierEngine.backward(): the last forward() ran with need_grad=False (its gate masks were not '
                               'written); run forward(..., need_grad=True) first')
        g_pre = self.body.backward(g_logits)
        _lib.call('spaa_preproc_bwd', _lib.ptr(g_pre), _lib.ptr(self.g_y), self.B, self.H, self.W, self.cy0, self.cx0,
                  self.ch, self.cw, self.oh, self.ow, self._std)
        return self.g_y


def _classify_impl(clf, im, crop_sz):
    """spaa::classify.  Returns (logits [B,ncls], saved)."""
    b, _, h, w = im.shape
    with _lib.on_device(im.device):
        eng = clf.engine(b, (h, w), tuple(crop_sz))
        im4 = to_nhwc4(im)
        logits = eng.forward(im4).clone()
        saved = dict(eng=eng, version=eng.version, im4=im4, clf=weakref.ref(clf), key=(b, (h, w), tuple(crop_sz)))
        clf._last_saved = saved
        return logits, saved


def _classify_backward_impl(saved, g):
    eng = saved['eng']
    with _lib.on_device(g.device):
        if eng.version != saved['version']:  # workspaces reused by a later forward: recompute this call's activations
            if eng.owner is not None and eng.owner() is not None and saved['clf']() is not None:
                # ... and the engine has since been leased to an attack state: leave ITS workspaces alone, take a free engine
                eng = saved['eng'] = saved['clf']().engine(*saved['key'])
            eng.forward(saved['im4'])
            saved['version'] = eng.version
        return to_nchw(eng.backward(g.detach().float().contiguous()))


class Classifier(object):
    """classifier.py:12-75 on HIP.  Extra keyword arguments: `state_dict`/`weights_path` (no download here) and
    `input_sz` (tests use reduced sizes)."""

    def __init__(self, model_name, device, device_ids=(0,), fix_params=True, sort_results=True, state_dict=None,
                 weights_path=None, input_sz=None):
        self.name = model_name
        self.fix_params = fix_params
        self.device = torch.device(device)
        self.sort_results = sort_results
        if model_name not in INPUT_SZ:
            raise ValueError(f'unknown classifier {model_name!r}')
        self.input_sz = tuple(input_sz) if input_sz is not None else INPUT_SZ[model_name]
        if state_dict is None and weights_path is not None:
            state_dict = torch.load(weights_path, map_location='cpu')
        if state_dict is None:
            raise RuntimeError('no network access: pass state_dict= (torchvision key names) or weights_path= instead of '
                               'the pretrained-weights URL the reference downloads (classifier.py:24-36)')
        if not fix_params:
            raise NotImplementedError('spaa_amd classifiers are frozen (input gradients only)')
        self.state_dict = {k: v.detach().float().cpu() for k, v in _strip(state_dict).items()}
        self._engines = {}

    def engine(self, batch, im_hw, crop_sz, owner=None, storage='f32'):
        """Cached engine for a batch size / geometry; an engine leased to an `owner` (attack state) is not handed to
        anyone else while the owner lives (see PCNet.engine).  `storage`: 'f32' or 'f16' (fp16 activations in HBM)."""
        key = (batch, tuple(im_hw), tuple(crop_sz), storage)
        pool = self._engines.setdefault(key, [])
        for e in pool:
            if e.owner is None or e.owner() is None:
                break
        else:
            with _lib.on_device(self.device):
                e = ClassifierEngine(self.name, self.state_dict, batch, im_hw, crop_sz, self.input_sz, self.device, storage)
            pool.append(e)
        e.owner = weakref.ref(owner) if owner is not None else None
        return e

    def classify(self, im, crop_sz=(240, 240)):
        if im.dtype == torch.uint8:
            im = im.type(torch.float32) / 255
        while im.ndim < 4:
            im = im[None]
        if self.device.type != 'cuda':
            raise RuntimeError('spaa_amd.Classifier runs on the GPU only (no CPU fallback); got device=%s' % self.device)
        from . import ops
        raw_score = torch.ops.spaa.classify(im.to(self.device), ops.handle_of(self), int(crop_sz[0]), int(crop_sz[1]))
        # Compatibility outputs (classifier.py:64-72).  The fused attack loop does NOT use these: it takes top-1 and
        # its probability on device (spaa_decide); the full 1000-way sort is only done for API parity here.
        p = torch.softmax(raw_score.detach(), dim=1).cpu()
        if self.sort_results:
            p_sorted, idx = p.sort(descending=True)
        else:
            p_sorted, idx = p, torch.arange(p.shape[1]).repeat(p.shape[0], 1)
        return raw_score, p_sorted.numpy(), idx.numpy()

    def __call__(self, im, crop_sz):
        return self.classify(im, crop_sz)


def load_imagenet_labels(filename):
    """classifier.py:109-116 (the label file is a Python dict literal)."""
    import ast
    with open(filename) as f:
        labels = ast.literal_eval(f.read())
    return {k: v.split(',')[0] for k, v in labels.items()}
