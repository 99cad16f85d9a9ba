"""Inception-v3 (torchvision.models.inception_v3, eval, transform_input=True, aux head unused) as a small op graph over
the tap-list convolution kernel and the generic pooling kernels: forward and input-gradient.

The reference builds it with `models.inception_v3(init_weights=False, transform_input=True)` and feeds 299x299 crops
(/root/reference/src/python/classifier.py:29-33).  The architecture is third-party (torchvision==0.15.1); it is
restated here from its published definition.  BasicConv2d = conv(bias=False) + BatchNorm(eps=1e-3) + ReLU, with the
BatchNorm folded into the convolution; concatenations are channel windows of one NHWC buffer (no copy).

fp16-STORAGE mode (`storage='f16'`: activations and their gradients fp16 in HBM, fp32 accumulation; image, pooled features
and logits fp32): the fp16 matrix-core kernels take 32-channel slices, so the two layers whose width is not a multiple of 32
(`Conv2d_3b_1x1`: 80, `Mixed_5x.branch5x5_1`: 48) are built 96 / 64 wide with ZERO weights and biases in the extra channels
(their activations and gradients are exactly 0; the consumer's weights for them are zero too).
"""
import os

import torch

from . import _lib
from . import convplan as cp
from .models import USE_GATE_MASKS

PAD32_F32 = int(os.environ.get('SPAA_INCEPTION_PAD32', '1'))
FUSE_ENTRY = os.environ.get('SPAA_INCEPTION_FUSE_ENTRY', '1') != '0'
BODY_GATE_MASKS = os.environ.get('SPAA_BODY_MASKS', '1') != '0'   # 0: the activation itself as the ReLU gate (A/B measurements)

# layer table: name -> builder spec.  conv spec = (name, cout, (kh, kw), stride, (ph, pw))
A_ = lambda pf: dict(kind='A', pf=pf)  # noqa: E731
STEM = [('Conv2d_1a_3x3', 32, (3, 3), 2, (0, 0)), ('Conv2d_2a_3x3', 32, (3, 3), 1, (0, 0)),
        ('Conv2d_2b_3x3', 64, (3, 3), 1, (1, 1)), 'maxpool', ('Conv2d_3b_1x1', 80, (1, 1), 1, (0, 0)),
        ('Conv2d_4a_3x3', 192, (3, 3), 1, (0, 0)), 'maxpool']
BLOCKS = [('Mixed_5b', 'A', 32), ('Mixed_5c', 'A', 64), ('Mixed_5d', 'A', 64), ('Mixed_6a', 'B', None),
          ('Mixed_6b', 'C', 128), ('Mixed_6c', 'C', 160), ('Mixed_6d', 'C', 160), ('Mixed_6e', 'C', 192),
          ('Mixed_7a', 'D', None), ('Mixed_7b', 'E', None), ('Mixed_7c', 'E', None)]


def block_spec(kind, arg):
    """Branches of an Inception block: list of (branch ops, concat?) in torchvision's concatenation order.
    op = ('conv', suffix, cout, (kh,kw), stride, (ph,pw)) | ('avg',) | ('max',) | ('split', [ops_a], [ops_b])."""
    c = lambda s, co, k=(1, 1), st=1, p=(0, 0): ('conv', s, co, k, st, p)  # noqa: E731
    if kind == 'A':
        return [[c('branch1x1', 64)],
                [c('branch5x5_1', 48), c('branch5x5_2', 64, (5, 5), 1, (2, 2))],
                [c('branch3x3dbl_1', 64), c('branch3x3dbl_2', 96, (3, 3), 1, (1, 1)),
                 c('branch3x3dbl_3', 96, (3, 3), 1, (1, 1))],
                [('avg',), c('branch_pool', arg)]]
    if kind == 'B':
        return [[c('branch3x3', 384, (3, 3), 2)],
                [c('branch3x3dbl_1', 64), c('branch3x3dbl_2', 96, (3, 3), 1, (1, 1)), c('branch3x3dbl_3', 96, (3, 3), 2)],
                [('max',)]]
    if kind == 'C':
        c7 = arg
        return [[c('branch1x1', 192)],
                [c('branch7x7_1', c7), c('branch7x7_2', c7, (1, 7), 1, (0, 3)), c('branch7x7_3', 192, (7, 1), 1, (3, 0))],
                [c('branch7x7dbl_1', c7), c('branch7x7dbl_2', c7, (7, 1), 1, (3, 0)),
                 c('branch7x7dbl_3', c7, (1, 7), 1, (0, 3)), c('branch7x7dbl_4', c7, (7, 1), 1, (3, 0)),
                 c('branch7x7dbl_5', 192, (1, 7), 1, (0, 3))],
                [('avg',), c('branch_pool', 192)]]
    if kind == 'D':
        return [[c('branch3x3_1', 192), c('branch3x3_2', 320, (3, 3), 2)],
                [c('branch7x7x3_1', 192), c('branch7x7x3_2', 192, (1, 7), 1, (0, 3)),
                 c('branch7x7x3_3', 192, (7, 1), 1, (3, 0)), c('branch7x7x3_4', 192, (3, 3), 2)],
                [('max',)]]
    if kind == 'E':
        return [[c('branch1x1', 320)],
                [c('branch3x3_1', 384), ('split', [c('branch3x3_2a', 384, (1, 3), 1, (0, 1))],
                                         [c('branch3x3_2b', 384, (3, 1), 1, (1, 0))])],
                [c('branch3x3dbl_1', 448), c('branch3x3dbl_2', 384, (3, 3), 1, (1, 1)),
                 ('split', [c('branch3x3dbl_3a', 384, (1, 3), 1, (0, 1))], [c('branch3x3dbl_3b', 384, (3, 1), 1, (1, 0))])],
                [('avg',), c('branch_pool', 192)]]
    raise ValueError(kind)


class Ten:
    """A channel window [coff, coff+C) of an NHWC buffer, plus its gradient buffer of identical layout."""

    def __init__(self, buf, coff, c, gbuf=None, kind='act', mbuf=None):
        # mbuf: ReLU-gate bytes of the whole buffer (uint8 [B, H, W, C_buffer / 4], the `mask_out` format of include/spaa_hip.h) or None
        self.buf, self.coff, self.c, self.gbuf, self.kind, self.mbuf = buf, coff, c, gbuf, kind, mbuf
        self.consumers = []
        self.g_written = False

    @property
    def hw(self):
        return self.buf.shape[1], self.buf.shape[2]

    def whole(self):
        return self.coff == 0 and self.c == self.buf.shape[3]


def _osz(n, k, s, p):
    return (n + 2 * p - k) // s + 1


class InceptionV3Body:
    def __init__(self, sd, batch, in_hw, dev, storage='f32'):
        from .classifier import _strip
        sd = _strip(sd)
        self.B, self.dev, self.sd, self.storage = batch, dev, sd, storage
        self.h16 = storage == 'f16'
        hd = torch.float16 if self.h16 else torch.float32
        self.ops = []
        h, w = in_hw
        self.in_hw = (h, w)

        def zf(*shape):
            return torch.zeros(*shape, device=dev)

        def z(*shape):
            return torch.zeros(*shape, device=dev, dtype=hd)

        self.z = z
        # ReLU gates as byte masks written by the forward epilogues (and by spaa_gate_mask for the max-pool outputs): an input-gradient
        # launch reads 2 bits per element instead of the activation and stays on the branch-free epilogue (epilogue.hpp fast_epi_*)
        self.masks = BODY_GATE_MASKS and (USE_GATE_MASKS or self.h16)
        self.write_masks = True       # (ClassifierEngine.forward(need_grad=False) clears it for a pass nobody differentiates)
        self.zm = lambda *shape: torch.zeros(*shape, dtype=torch.uint8, device=dev) if self.masks else None
        # transform_input=True (torchvision): per-channel affine on the already-normalised image, folded into the
        # first convolution:  x' = a_c * x + b_c  =>  W' = W * a_c, bias' += sum_taps W * b_c  -- valid only where no
        # zero padding is involved: Conv2d_1a has padding 0, so the folding is exact.
        self.x_in = Ten(zf(batch, h, w, 4), 0, 3, zf(batch, h, w, 4), kind='input')
        t = self.x_in
        first = True
        for item in STEM:
            if item == 'maxpool':
                t = self.add_pool('max', t, 3, 2, 0)
            else:
                name, co, k, st, p = item
                t = self.add_conv(name, t, co, k, st, p, fold_input_affine=first)
                first = False
        for name, kind, arg in BLOCKS:
            t = self.add_block(name, kind, arg, t)
        self.feat = t
        fh, fw = t.hw
        self.feat_hw = fh * fw
        self.pooled = zf(batch, 1, 1, 2048)
        self.g_pooled = zf(batch, 1, 1, 2048)
        self.ncls = sd['fc.weight'].shape[0]
        self.fc_f = cp.linear_fwd_plan(sd['fc.weight'], sd['fc.bias'], dev, 'fc')
        self.fc_d = cp.linear_dgrad_plan(sd['fc.weight'], dev, 'fc_dgrad')
        self.logits = zf(batch, 1, 1, self.ncls)

    # ---- graph construction ------------------------------------------------------------------------------------
    def folded(self, name, fold_input_affine=False):
        sd = self.sd
        wgt, b = cp.fold_bn(sd[name + '.conv.weight'], sd[name + '.bn.weight'], sd[name + '.bn.bias'],
                            sd[name + '.bn.running_mean'], sd[name + '.bn.running_var'], eps=1e-3)
        if fold_input_affine:
            a = torch.tensor([0.229 / 0.5, 0.224 / 0.5, 0.225 / 0.5], dtype=torch.float64)
            c = torch.tensor([(0.485 - 0.5) / 0.5, (0.456 - 0.5) / 0.5, (0.406 - 0.5) / 0.5], dtype=torch.float64)
            wd = wgt.double()
            b = (b.double() + (wd * c.view(1, 3, 1, 1)).sum(dim=(1, 2, 3))).float()
            wgt = (wd * a.view(1, 3, 1, 1)).float()
        return wgt, b

    def add_conv(self, name, inp, cout, k, stride, pad, out=None, fold_input_affine=False):
        hin, win = inp.hw
        ho, wo = _osz(hin, k[0], stride, pad[0]), _osz(win, k[1], stride, pad[1])
        wgt, b = self.folded(name, fold_input_affine)
        # fp32 storage: Conv2d_3b_1x1's 80 channels are built 96 wide as well (SPAA_INCEPTION_PAD32: 0 never, 1 widths above 64 --
        # the default --, 2 every width): Conv2d_4a_3x3 (80 -> 192, unpadded 3x3 on 73 x 73) then reads whole 32-channel blocks and
        # runs on the Winograd kernel's pad-0 form instead of the register-staged tile
        pad32 = self.h16 or (PAD32_F32 == 2 and out is None) or (PAD32_F32 == 1 and out is None and cout > 64)
        if pad32 or (inp.kind != 'input' and inp.c != wgt.shape[1]):   # 32-channel slices: zero-padded widths (see the module docstring)
            cout_p = cout if (out is not None or not pad32) else -(-cout // 32) * 32
            cin_p = inp.c if inp.kind != 'input' else wgt.shape[1]
            if cout_p != cout or cin_p != wgt.shape[1]:
                wp = torch.zeros(cout_p, cin_p, *wgt.shape[2:], dtype=wgt.dtype)
                wp[:cout, :wgt.shape[1]] = wgt
                bp = torch.zeros(cout_p, dtype=b.dtype)
                bp[:cout] = b
                wgt, b, cout = wp, bp, cout_p
        if out is None:
            out = Ten(self.z(self.B, ho, wo, cout), 0, cout, self.z(self.B, ho, wo, cout), mbuf=self.zm(self.B, ho, wo, cout // 4))
        assert out.hw == (ho, wo) and out.c == cout
        op = dict(kind='conv', name=name, inp=inp, out=out,
                  f=cp.conv_fwd_plan(wgt, b, stride, pad, self.dev, name),
                  d=cp.conv_dgrad_plan(wgt, stride, pad, self.dev, name + '_dgrad'))
        inp.consumers.append(op)
        self.ops.append(op)
        return out

    def add_pool(self, mode, inp, k, s, p, out=None):
        assert inp.whole(), 'pool inputs are whole buffers'
        hin, win = inp.hw
        ho, wo = _osz(hin, k, s, p), _osz(win, k, s, p)
        if out is None:
            # (an average pool's output needs no gate of its own: where it is 0 every input of the window is 0 and gated itself)
            out = Ten(self.z(self.B, ho, wo, inp.c), 0, inp.c, self.z(self.B, ho, wo, inp.c), kind='pool' if mode == 'max' or not self.masks else 'avgpool',
                      mbuf=self.zm(self.B, ho, wo, inp.c // 4) if mode == 'max' else None)
        op = dict(kind=mode, inp=inp, out=out, k=k, s=s, p=p)
        if mode == 'max':
            op['arg'] = torch.zeros(self.B, ho, wo, inp.c, dtype=torch.uint8, device=self.dev)
        inp.consumers.append(op)
        self.ops.append(op)
        return out

    def add_block(self, name, kind, arg, inp):
        spec = block_spec(kind, arg)
        hin, win = inp.hw

        def out_channels(ops):
            last = ops[-1]
            if last[0] == 'conv':
                return last[2]
            if last[0] == 'split':
                return out_channels(last[1]) + out_channels(last[2])
            return inp.c  # max-pool branch passes the input channels through

        def spatial(ops, h, w):
            for o in ops:
                if o[0] == 'conv':
                    h, w = _osz(h, o[3][0], o[4], o[5][0]), _osz(w, o[3][1], o[4], o[5][1])
                elif o[0] == 'max':
                    h, w = _osz(h, 3, 2, 0), _osz(w, 3, 2, 0)
                elif o[0] == 'split':
                    h, w = spatial(o[1], h, w)
            return h, w

        ctot = sum(out_channels(b) for b in spec)
        ho, wo = spatial(spec[0], hin, win)
        cat = self.z(self.B, ho, wo, ctot)
        gcat = self.z(self.B, ho, wo, ctot)
        mcat = self.zm(self.B, ho, wo, ctot // 4)

        def run_chain(ops, t, coff):
            """Builds ops; the LAST op of the chain writes into the concat window starting at coff."""
            for i, o in enumerate(ops):
                is_last = i == len(ops) - 1
                if o[0] == 'conv':
                    dst = Ten(cat, coff, o[2], gcat, mbuf=mcat) if is_last else None
                    t = self.add_conv(f'{name}.{o[1]}', t, o[2], o[3], o[4], o[5], out=dst)
                elif o[0] == 'avg':
                    t = self.add_pool('avg', t, 3, 1, 1)
                elif o[0] == 'max':
                    dst = Ten(cat, coff, t.c, gcat, kind='pool', mbuf=mcat) if is_last else None
                    t = self.add_pool('max', t, 3, 2, 0, out=dst)
                elif o[0] == 'split':
                    assert is_last
                    ca = out_channels(o[1])
                    run_chain(o[1], t, coff)
                    run_chain(o[2], t, coff + ca)
            return t

        # The 1 x 1 convolutions that OPEN a multi-layer branch read the same block input (C blocks: branch7x7_1 and branch7x7dbl_1;
        # A: branch5x5_1, branch3x3dbl_1; D: branch3x3_1, branch7x7x3_1; E: branch3x3_1, branch3x3dbl_1): built as ONE convolution whose
        # output channels are the branches' intermediate tensors side by side in one buffer -- the input is read once instead of twice,
        # and the backward pass has one accumulating input-gradient launch over the concatenated K instead of two read-modify-write
        # passes over the block's input gradient (SPAA_INCEPTION_FUSE_ENTRY=0: separate launches, A/B measurements).
        entries = [bi for bi, br in enumerate(spec) if len(br) > 1 and br[0][0] == 'conv' and br[0][3] == (1, 1) and br[0][4] == 1]
        heads = {}
        if FUSE_ENTRY and len(entries) >= 2:
            ws, bs, widths = [], [], []
            for bi in entries:
                o = spec[bi][0]
                wgt, b = self.folded(f'{name}.{o[1]}')
                c = o[2]
                cp_ = -(-c // 32) * 32 if (self.h16 or PAD32_F32 == 2) else c      # (zero pad channels, as add_conv)
                if cp_ != c:
                    wgt = torch.cat([wgt, torch.zeros(cp_ - c, *wgt.shape[1:], dtype=wgt.dtype)], 0)
                    b = torch.cat([b, torch.zeros(cp_ - c, dtype=b.dtype)], 0)
                ws.append(wgt), bs.append(b), widths.append(cp_)
            ce = sum(widths)
            ebuf, gebuf, mebuf = self.z(self.B, hin, win, ce), self.z(self.B, hin, win, ce), self.zm(self.B, hin, win, ce // 4)
            eout = Ten(ebuf, 0, ce, gebuf, mbuf=mebuf)
            wcat, bcat = torch.cat(ws, 0), torch.cat(bs, 0)
            if inp.c != wcat.shape[1]:     # (an input with zero pad channels)
                wcat = torch.cat([wcat, torch.zeros(ce, inp.c - wcat.shape[1], 1, 1, dtype=wcat.dtype)], 1)
            op = dict(kind='conv', name=f'{name}.entry', inp=inp, out=eout, outs=[],
                      f=cp.conv_fwd_plan(wcat, bcat, 1, (0, 0), self.dev, f'{name}.entry'),
                      d=cp.conv_dgrad_plan(wcat, 1, (0, 0), self.dev, f'{name}.entry_dgrad'))
            inp.consumers.append(op)
            self.ops.append(op)
            eoff = 0
            for bi, wd in zip(entries, widths):
                heads[bi] = Ten(ebuf, eoff, wd, gebuf, mbuf=mebuf)
                op['outs'].append((f'{name}.{spec[bi][0][1]}', heads[bi]))
                eoff += wd
        coff = 0
        for bi, br in enumerate(spec):
            if bi in heads:
                run_chain(br[1:], heads[bi], coff)
            else:
                run_chain(br, inp, coff)
            coff += out_channels(br)
        return Ten(cat, 0, ctot, gcat, mbuf=mcat)

    # ---- execution ---------------------------------------------------------------------------------------------
    def forward(self, x4):
        B = self.B
        self.x_in.buf = x4
        wm = self.masks and self.write_masks
        for op in self.ops:
            i, o = op['inp'], op['out']
            if op['kind'] == 'conv':
                if i is self.x_in:
                    i.buf = x4
                op['f'].run(i.buf, o.buf, act=_lib.ACT_RELU, in_coff=i.coff, out_coff=o.coff, mask_out=o.mbuf if wm else None)
            elif op['kind'] == 'max':
                hin, win = i.hw
                ho, wo = o.hw
                _lib.call('spaa_maxpool_fwd_f16' if self.h16 else 'spaa_maxpool_fwd', _lib.hptr(i.buf), _lib.hptr(o.buf),
                          _lib.ptr(op['arg']), B, hin, win, i.c, ho, wo, op['k'], op['s'], op['p'], o.buf.shape[3], o.coff)
                if wm:   # the pooled values gate the layers that consume them like every other ReLU output
                    _lib.call('spaa_gate_mask', _lib.hptr(o.buf), int(self.h16), _lib.ptr(o.mbuf), B * ho * wo, i.c, o.buf.shape[3], o.coff)
            else:
                hin, win = i.hw
                ho, wo = o.hw
                _lib.call('spaa_avgpool2d_fwd_f16' if self.h16 else 'spaa_avgpool2d_fwd', _lib.hptr(i.buf), _lib.hptr(o.buf), B,
                          hin, win, i.c, ho, wo, op['k'], op['s'], op['p'], o.buf.shape[3], o.coff)
        _lib.call('spaa_avgpool_fwd_f16' if self.h16 else 'spaa_avgpool_fwd', _lib.hptr(self.feat.buf), _lib.ptr(self.pooled), B,
                  self.feat_hw, 2048)
        self.fc_f.run(self.pooled, self.logits)
        return self.logits.view(B, self.ncls)

    def backward(self, g_logits):
        B = self.B
        self.fc_d.run(g_logits.view(B, 1, 1, self.ncls), self.g_pooled)
        # gradient w.r.t. the pre-activations of the last concat (all four slices are ReLU outputs)
        _lib.call('spaa_avgpool_bwd_f16' if self.h16 else 'spaa_avgpool_bwd', _lib.ptr(self.g_pooled), _lib.hptr(self.feat.buf),
                  _lib.hptr(self.feat.gbuf), B, self.feat_hw, 2048)
        written = set()
        for op in reversed(self.ops):
            i, o = op['inp'], op['out']
            key = (id(i.gbuf), i.coff)
            last = i.consumers[0] is op          # processed last in reverse order -> applies the ReLU gate of `i`
            gated = last and i.kind not in ('input', 'avgpool')
            gate = i.buf if (gated and not self.masks) else None
            if op['kind'] == 'conv':
                add = i.gbuf if key in written else None
                op['d'].run(o.gbuf, i.gbuf, add=add, gate=gate, gate_bits=i.mbuf if (gated and self.masks) else None, in_coff=o.coff,
                            out_coff=i.coff, add_coff=i.coff, gate_coff=i.coff)
            else:
                assert key not in written, 'pool branches must be the first gradient contribution of their input'
                hin, win = i.hw
                ho, wo = o.hw
                if op['kind'] == 'max':
                    _lib.call('spaa_maxpool_bwd_f16' if self.h16 else 'spaa_maxpool_bwd', _lib.hptr(o.gbuf), _lib.ptr(op['arg']),
                              int(gated), _lib.hptr(i.gbuf), B, hin, win, i.c, ho, wo,
                              op['k'], op['s'], op['p'], o.gbuf.shape[3], o.coff)
                else:
                    assert not gated, 'avg-pool is never the only consumer in Inception-v3'
                    _lib.call('spaa_avgpool2d_bwd_f16' if self.h16 else 'spaa_avgpool2d_bwd', _lib.hptr(o.gbuf), _lib.hptr(i.gbuf), B,
                              hin, win, i.c, ho, wo, op['k'], op['s'], op['p'], o.gbuf.shape[3], o.coff)
            written.add(key)
        return self.x_in.gbuf

    def refresh_masks(self):
        """Recompute the gate masks from the activation buffers (after a test has overwritten the activations)."""
        if not self.masks:
            return
        for op in self.ops:
            o = op['out']
            if op['kind'] in ('conv', 'max') and o.mbuf is not None:
                o.mbuf[..., o.coff // 4:(o.coff + o.c) // 4] = _lib.pack_gate_mask(o.buf[..., o.coff:o.coff + o.c].float())

    def flops_fwd(self):
        t = 0
        for op in self.ops:
            if op['kind'] == 'conv':
                t += op['f'].flops(self.B, *op['out'].hw)
        return t + self.fc_f.flops(self.B, 1, 1)
