"""fp16 storage: magnitude of the gradients a classifier body stores, per layer, for the loss scale AttackState uses (+-16 at one logit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import synthetic as syn, models as M
from spaa_amd.classifier import Classifier
DEV = 'cuda:0'
body = sys.argv[1] if len(sys.argv) > 1 else 'inception_v3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
csd = {'resnet18': syn.resnet18_state_dict, 'vgg16': syn.vgg16_state_dict, 'inception_v3': syn.inception_v3_state_dict}[body](2, logit_gain=20.0)
SMALL = len(sys.argv) > 3
clf = Classifier(body, DEV, state_dict=csd, input_sz=(107, 107) if SMALL else None)
IM, CROP = ((128, 128), (120, 120)) if SMALL else ((256, 256), (240, 240))
im = syn.scenes(8, 2, IM)
res = {}
for storage in ('f32', 'f16'):
    ce = clf.engine(2, IM, CROP, storage=storage)
    logits = ce.forward(M.to_nhwc4(im.to(DEV)))
    g = torch.zeros_like(logits)
    g[0, 3], g[1, 7] = -scale, -scale
    gx = ce.backward(g.contiguous()).float().clone()
    res[storage] = gx
    if storage == 'f16' and hasattr(ce.body, 'ops'):
        rows = []
        for op in ce.body.ops:
            if op['kind'] == 'conv' and 'out' in op and hasattr(op['out'], 'gbuf'):
                t = op['out']
                gb = t.gbuf[..., t.coff:t.coff + t.c].float().abs()
                nz = gb[gb > 0]
                rows.append((op['name'], float(gb.max()), float(nz.median()) if nz.numel() else 0.0, float((nz < 6.1e-5).float().mean()) if nz.numel() else 0.0))
        for r in rows[::6]:
            print(f'{r[0]:32s} max {r[1]:.2e} median {r[2]:.2e} subnormal fraction {r[3]:.2f}')
e = (res['f16'] - res['f32']).norm() / res['f32'].norm()
print(f'{body} scale {scale}: input-gradient rel L2 f16 vs f32 engine {float(e):.3e}; |g| max {float(res["f32"].abs().max()):.2e}')

# gate-aware at the classifier level: the fp32 engine's activations / arg-maxes copied into the fp16 engine before its backward
if hasattr(clf.engine(2, IM, CROP, storage='f16').body, 'ops') and body == 'inception_v3':
    c32 = clf.engine(2, IM, CROP, storage='f32')
    c16 = clf.engine(2, IM, CROP, storage='f16')
    x4 = M.to_nhwc4(im.to(DEV))
    l32 = c32.forward(x4)
    l16 = c16.forward(x4)
    g = torch.zeros_like(l32)
    g[0, 3], g[1, 7] = -scale, -scale
    g32 = c32.backward(g.contiguous()).float().clone()
    for o32, o16 in zip(c32.body.ops, c16.body.ops):
        t32, t16 = o32['out'], o16['out']
        a = t32.buf[..., t32.coff:t32.coff + t32.c]
        t16.buf[..., t16.coff:t16.coff + a.shape[-1]].copy_(a.half())
        if o32['kind'] == 'max':
            o16['arg'].copy_(o32['arg'])
    g16 = c16.backward(g.contiguous()).float().clone()
    print(f'with the fp32 engine\'s activations and arg-maxes: rel L2 {float((g16 - g32).norm() / g32.norm()):.3e}, rel Linf {float((g16 - g32).abs().max() / g32.abs().max()):.3e}')
