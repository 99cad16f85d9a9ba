"""TEST INFRASTRUCTURE: ReLU / clamp / max-pool gate bookkeeping between the HIP engines and the oracle.

The input gradient of a ReLU network is a discontinuous function of its activations' signs: a unit whose value is
within rounding of zero may be "on" in one fp32 implementation and "off" in another (different summation order), and
that one bit changes the gradient inside the unit's whole receptive field by O(1).  The parity tests therefore
  (1) count, per sample, the gates on which the HIP forward and the oracle forward disagree, and check that every such
      unit is within rounding of zero on BOTH sides;
  (2) assert the 1e-4 bar on every sample without a disagreeing gate;
  (3) re-run the HIP backward with the ORACLE's gates copied into the engine's activation buffers and assert the 1e-4
      bar on every sample: whatever exceeded it in (2) is then shown to be the gates and nothing else.
"""
import torch


def _nhwc(o, cpad=None):
    """oracle NCHW (or [B,C]) -> NHWC float tensor, channels zero-padded to `cpad`."""
    if o.ndim == 2:
        o = o[:, :, None, None]
    t = o.detach().float().permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad > t.shape[-1]:
        t = torch.cat([t, torch.zeros(*t.shape[:-1], cpad - t.shape[-1])], -1)
    return t


def _argmax_codes(idx, pooled_in, hin_win, k, s, p):
    """F.max_pool2d(return_indices=True) flat input indices [B,C,Ho,Wo] (+ the pooled tensor's input, a ReLU output) ->
    the HIP kernels' argmax bytes, NHWC uint8: window code ky*k+kx in bits 0-6, bit 7 = (maximum > 0)."""
    hin, win = hin_win
    b, c, ho, wo = idx.shape
    iy, ix = idx // win, idx % win
    oy = torch.arange(ho).view(1, 1, ho, 1)
    ox = torch.arange(wo).view(1, 1, 1, wo)
    code = (iy - (oy * s - p)) * k + (ix - (ox * s - p))
    assert int(code.min()) >= 0 and int(code.max()) < k * k
    pos = pooled_in.flatten(2).gather(2, idx.flatten(2)).view_as(idx) > 0
    return (code + 128 * pos).permute(0, 2, 3, 1).contiguous().to(torch.uint8)


def _window_values(codes, pooled_nhwc, k, s, p):
    """Value of the pooled input at the window position a byte code (bits 0-6 = ky * k + kx) names, per output element
    [B,Ho,Wo,C] (positions outside the input, which no correct code can name, read -inf)."""
    b, ho, wo, c = codes.shape
    hin, win = pooled_nhwc.shape[1:3]
    code = (codes & 127).long()
    iy = torch.arange(ho).view(1, ho, 1, 1) * s - p + code // k
    ix = torch.arange(wo).view(1, 1, wo, 1) * s - p + code % k
    ok = (iy >= 0) & (iy < hin) & (ix >= 0) & (ix < win)
    flat = (iy.clamp(0, hin - 1) * win + ix.clamp(0, win - 1))                     # [B,Ho,Wo,C]
    src = pooled_nhwc.reshape(b, hin * win, c)
    v = src.gather(1, flat.reshape(b, ho * wo, c)).reshape(b, ho, wo, c)
    return torch.where(ok, v, torch.full_like(v, float('-inf')))


class ArgmaxRef:
    """Oracle-side arg-max codes of a max-pool plus what `count_flips` needs to show that a disagreement is a TIE within
    rounding: the pooled input on both sides (NHWC) and the window geometry."""

    def __init__(self, codes, hip_in, orc_in, k, s, p):
        self.codes, self.hip_in, self.orc_in, self.k, self.s, self.p = codes, hip_in, orc_in, k, s, p
        self.shape = codes.shape

    def to(self, *a, **kw):   # (inject() copies the codes into the engine's byte buffer)
        return self.codes.to(*a, **kw)


def pcnet_pairs(eng, acts):
    """(name, kind, HIP buffer, oracle tensor in the HIP layout) for every gate of PCNetEngine.backward + select_grad."""
    m = dict(x1='X1', x2='X2', x3='X3', x4='X4', x5='X5', x6='X6', x7='X7', res1_s='S1', res2_s='S2', res3_s='S3',
             res4_s='S4')
    out = [(f'pcnet.{k}', 'relu', eng.a[v], _nhwc(acts[k])) for k, v in m.items()]
    out.append(('pcnet.ypre', 'clamp01', eng.a['Ypre'], _nhwc(acts['ypre'], 4)))
    return out


def resnet18_pairs(body, cacts):
    out = [('resnet.c1', 'relu', body.c1, _nhwc(cacts['c1']))]
    out.append(('resnet.maxpool', 'argmax', body.mp_arg,
                ArgmaxRef(_argmax_codes(cacts['mp_idx'], cacts['c1'], body.c1.shape[1:3], 3, 2, 1), body.c1, _nhwc(cacts['c1']),
                          3, 2, 1)))
    for blk in body.blocks:
        out.append((f'resnet.{blk["name"]}.o1', 'relu', blk['o1'], _nhwc(cacts[blk['name'] + '.o1'])))
        out.append((f'resnet.{blk["name"]}.out', 'relu', blk['out'], _nhwc(cacts[blk['name'] + '.out'])))
    return out


def vgg16_pairs(body, cacts):
    out, nc, npool = [], 0, 0
    for op in body.ops:
        if op['kind'] == 'conv':
            out.append((f'vgg.conv{nc}', 'relu', op['out'], _nhwc(cacts[f'conv{nc}'])))
            nc += 1
        else:
            out.append((f'vgg.pool{npool}', 'argmax', op['arg'],
                        ArgmaxRef(_argmax_codes(cacts[f'pool{npool}'], cacts[f'conv{nc - 1}'], (op['hin'], op['win']), 2, 2, 0),
                                  out[-1][2], out[-1][3], 2, 2, 0)))
            npool += 1
    out.append(('vgg.fc1', 'relu', body.h1, _nhwc(cacts['fc1'])))
    out.append(('vgg.fc2', 'relu', body.h2, _nhwc(cacts['fc2'])))
    return out


class _Recorded(list):
    """ReLU outputs in call order; `.pools` = the (input, kernel, stride, padding) of every F.max_pool2d call."""
    pools = None


class record_relu:
    """Context manager: every F.relu output produced inside (the oracle's forward) is collected in call order, and the
    input and geometry of every F.max_pool2d call (for the arg-max gates)."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.orig, self.orig_mp, self.outs = F, F.relu, F.max_pool2d, _Recorded()
        self.outs.pools = []

        def relu(t, *a, **k):
            o = self.orig(t, *a, **k)
            self.outs.append(o.detach())
            return o

        def max_pool2d(t, kernel_size, stride=None, padding=0, *a, **k):
            self.outs.pools.append((t.detach(), kernel_size, stride if stride is not None else kernel_size, padding))
            return self.orig_mp(t, kernel_size, stride, padding, *a, **k)

        F.relu, F.max_pool2d = relu, max_pool2d
        return self.outs

    def __exit__(self, *exc):
        self.F.relu, self.F.max_pool2d = self.orig, self.orig_mp
        return False


def inception_pairs(body, relu_outs):
    """Inception-v3: the HIP body's 94 convolutions' outputs (every one followed by a ReLU) are matched with the oracle's recorded
    ReLU outputs by shape and value (call orders differ inside the mixed blocks)."""
    out, used = [], set()
    h16 = getattr(body, 'h16', False)     # fp16 storage: matching to fp16 accuracy; two layers carry zero pad channels
    # (a fused branch-entry launch, spaa_amd/inception.py add_block, carries its logical outputs -- channel windows -- in 'outs')
    logical = [(nm, t) for o in body.ops if o['kind'] == 'conv' for nm, t in (o.get('outs') or [(o.get('name'), o['out'])])]
    for n, (lname, t) in enumerate(logical):
        op = dict(name=lname)
        hip = t.buf[..., t.coff:t.coff + t.c]
        h = hip.detach().float().cpu()
        best = None
        for i, r in enumerate(relu_outs):
            if i in used or r.ndim != 4 or (r.shape[0], r.shape[2], r.shape[3]) != tuple(h.shape[:3]):
                continue
            if r.shape[1] != h.shape[3] and not (r.shape[1] < h.shape[3] and -(-r.shape[1] // 32) * 32 == h.shape[3]):   # (zero pad channels)
                continue
            e = float((r.permute(0, 2, 3, 1) - h[..., :r.shape[1]]).abs().max()) / (float(r.abs().max()) + 1e-30)
            if best is None or e < best[1]:
                best = (i, e)
        assert best is not None and best[1] < (2e-2 if h16 else 1e-3), (n, op.get('name'), best)
        used.add(best[0])
        c = relu_outs[best[0]].shape[1]
        assert c == h.shape[3] or float(h[..., c:].abs().max()) == 0.0      # (pad channels are exactly zero)
        out.append((f'inception.{op.get("name", n)}', 'relu', hip[..., :c], _nhwc(relu_outs[best[0]])))
    # the max-pools (stem x 2, Mixed_6a, Mixed_7a), in call order on both sides
    import torch.nn.functional as F
    pools = getattr(relu_outs, 'pools', None) or []
    mops = [o for o in body.ops if o['kind'] == 'max']
    assert len(pools) in (0, len(mops)), (len(pools), len(mops))
    for n, (op, (inp, k, st, pd)) in enumerate(zip(mops, pools)):
        pooled, idx = F.max_pool2d(inp, k, st, pd, return_indices=True)
        i, o = op['inp'], op['out']
        c = inp.shape[1]
        # the pooled tensor itself gates the layers that consume it (its sign: maximum > 0)
        out.append((f'inception.maxpool{n}.out', 'relu', o.buf[..., o.coff:o.coff + c], _nhwc(pooled)))
        out.append((f'inception.maxpool{n}', 'argmax', op['arg'][..., :c] if op['arg'].shape[3] != c else op['arg'],
                    ArgmaxRef(_argmax_codes(idx, inp, inp.shape[2:], k, st, pd), i.buf[..., :c], _nhwc(inp), k, st, pd)))
    return out


# Measured maxima over a test session (printed by the parity tests into profiles/r03_parity.txt): the tolerances below are
# ~3x these, not a guess.  value: max over layers of |hip - oracle| / max|oracle|; near: max distance of a disagreeing unit
# from its threshold / layer scale; tie: max difference of the two candidates of a disagreeing arg-max / layer scale.
MEASURED = dict(value=0.0, near=0.0, tie=0.0)
NEAR_ZERO = 1.3e-6   # 3 x the largest seen (4.2e-7 of the layer's largest activation; ties 1.6e-7): profiles/r03_parity.txt
VALUE_TOL = 1.3e-5   # 3 x 4.1e-6


def count_flips(pairs, near_zero=NEAR_ZERO, value_tol=VALUE_TOL, measured=None):
    """Per-sample number of gates on which HIP and oracle disagree.  Asserts (a) the activations themselves agree to
    `value_tol` relative L-inf per layer, (b) every disagreeing ReLU/clamp unit is within `near_zero` x layer scale of the
    gate's threshold on both sides, (c) every disagreeing max-pool arg-max is a tie: the two candidate inputs are within
    `near_zero` x layer scale of each other on both sides (and a disagreeing "maximum > 0" bit is a maximum within
    `near_zero` of zero).  Returns (flips [B] int tensor, {layer: count}).  `measured`: where the largest quantities seen are
    recorded (default: the module's MEASURED, the fp32 record the tolerances above are derived from)."""
    flips, per_layer = None, {}
    MEASURED = measured if measured is not None else globals()['MEASURED']   # (fp16-storage runs keep their own record)
    for name, kind, hip, orc in pairs:
        h = hip.detach().cpu()
        assert h.shape == orc.shape, (name, h.shape, orc.shape)
        if kind == 'argmax':
            oc = orc.codes
            mism = h != oc
            if mism.any():
                hin, oin = orc.hip_in.detach().float().cpu(), orc.orc_in
                scale = float(oin.abs().max()) + 1e-30
                for src in (hin, oin):   # the value each side's choice has in THIS side's input: a tie on both sides
                    va = _window_values(h, src, orc.k, orc.s, orc.p)
                    vb = _window_values(oc, src, orc.k, orc.s, orc.p)
                    win = ((h & 127) != (oc & 127))
                    if win.any():
                        tie = float((va - vb).abs()[win].max()) / scale
                        MEASURED['tie'] = max(MEASURED['tie'], tie)
                        assert tie < near_zero, (name, 'arg-max disagreement that is not a tie', tie)
                    pos = ((h & 128) != (oc & 128))
                    if pos.any():
                        near = float(torch.maximum(va.abs(), vb.abs())[pos].max()) / scale
                        MEASURED['near'] = max(MEASURED['near'], near)
                        assert near < near_zero, (name, 'max > 0 bit differs away from zero', near)
        else:
            hf = h.float()
            scale = float(orc.abs().max()) + 1e-30
            verr = float((hf - orc).abs().max()) / scale
            MEASURED['value'] = max(MEASURED['value'], verr)
            assert verr <= value_tol, (name, verr)
            if kind == 'relu':
                mism = (hf > 0) != (orc > 0)
                if mism.any():
                    near = float(torch.maximum(hf.abs(), orc.abs())[mism].max()) / scale
                    MEASURED['near'] = max(MEASURED['near'], near)
                    assert near < near_zero, (name, near)
            else:  # 0 < v <= 1
                mism = ((hf > 0) & (hf <= 1)) != ((orc > 0) & (orc <= 1))
                if mism.any():
                    d0 = torch.maximum(hf.abs(), orc.abs())[mism]
                    d1 = torch.maximum((hf - 1).abs(), (orc - 1).abs())[mism]
                    near = float(torch.minimum(d0, d1).max()) / max(scale, 1.0)
                    MEASURED['near'] = max(MEASURED['near'], near)
                    assert near < near_zero, (name, near)
        n = mism.flatten(1).sum(dim=1)
        flips = n if flips is None else flips + n
        if int(n.sum()):
            per_layer[name] = int(n.sum())
    return flips, per_layer


def inject(pairs, engines=()):
    """Copy the oracle's activations / arg-maxes into the HIP engine's buffers (they agree to rounding: only the gates
    that sit within rounding of zero change), then let the engines rebuild the byte masks their backward passes read."""
    for name, kind, hip, orc in pairs:
        hip.copy_(orc.to(hip.dtype).to(hip.device))   # (`hip` may be a channel window of a concatenation buffer: a view)
    for e in engines:
        if hasattr(e, 'refresh_masks'):
            e.refresh_masks()
