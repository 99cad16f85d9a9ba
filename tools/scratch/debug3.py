import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F, spaa_oracle as so
from spaa_amd import _lib as lib, synthetic as syn
from spaa_amd.classifier import Classifier
from spaa_amd.models import PCNet, WarpingNet, to_nchw, to_nhwc4
from spaa_amd.projector_based_attack import AttackState
DEV = 'cuda'
sz, crop, insz = (128, 128), (120, 120), (107, 107)
sd = syn.pcnet_state_dict(12, cam_sz=sz, mask='ones')
pc = PCNet(sd['mask'], WarpingNet(out_size=sz)); pc.load_state_dict(sd); pc = pc.to(DEV)
csd = syn.inception_v3_state_dict(4, logit_gain=20.0)
clf = Classifier('inception_v3', DEV, state_dict=csd, input_sz=insz)
scene = syn.scenes(13, 1, sz)
setup = dict(classifier_crop_sz=crop, prj_brightness=0.5, prj_im_sz=sz)
targets = [204, 291]
st = AttackState(pc, clf, targets, scene, 'camdE_caml2', setup, DEV)
st.forward_decide(True, 5, 0.9)
y = to_nchw(st.eng.a['Y']).cpu()
print('cam image: min/max', y.min().item(), y.max().item(), 'fraction == 1:', (y == 1).float().mean().item(), 'fraction == 0:', (y == 0).float().mean().item())
g_y = to_nchw(st.clf.backward(st.g_logits)).cpu()
# oracle with every ReLU output's gradient retained
kept = []
orig_relu = F.relu
def relu_keep(t, *a, **k):
    o = orig_relu(t, *a, **k)
    if o.requires_grad:
        o.retain_grad()
        kept.append(o)
    return o
F.relu = relu_keep
yc = y.clone().requires_grad_(True)
raw, p, idx = so.OracleClassifier('inception_v3', csd, input_sz=insz)(yc, crop)
F.relu = orig_relu
(-raw[torch.arange(2), torch.tensor(targets)]).mean().backward()
print('input grad rel L2', ((g_y - yc.grad).norm() / yc.grad.norm()).item())
body = st.clf.body
ops = [op for op in body.ops if op['kind'] == 'conv']
print(len(ops), 'HIP convs;', len(kept), 'oracle relus')
used = set()
for n, op in enumerate(ops):
    o = op['out']
    gb = o.gbuf[..., o.coff:o.coff + o.c].permute(0, 3, 1, 2).float().cpu()
    ab = o.buf[..., o.coff:o.coff + o.c].permute(0, 3, 1, 2).float().cpu()
    best = None
    for i, t in enumerate(kept):
        if i in used or tuple(t.shape) != tuple(ab.shape):
            continue
        e = (t.detach() - ab).abs().max().item() / (t.detach().abs().max().item() + 1e-30)
        if best is None or e < best[1]:
            best = (i, e)
    if best is None or best[1] > 1e-3:
        print('op', n, op.get('name', ''), tuple(ab.shape), 'no oracle match', best)
        continue
    used.add(best[0])
    t = kept[best[0]]
    gref = t.grad * (t.detach() > 0)
    ge = ((gb - gref).norm() / (gref.norm() + 1e-30)).item()
    if n in (79, 80, 81):
        tt = t.detach()
        mism = (ab > 0) != (tt > 0)
        print('   op', n, 'gate mismatches', int(mism.sum()), 'of', mism.numel(), 'max |act| at mismatches', float(torch.maximum(ab.abs(), tt.abs())[mism].max()) if mism.any() else 0.0,
              'act max', float(tt.abs().max()), 'frac act>0', float((tt > 0).float().mean()))
        gu = t.grad
        print('   ungated oracle grad norm', float(gu.norm()), 'gated', float(gref.norm()), 'HIP', float(gb.norm()), ' HIP vs oracle-gated-by-HIP-gates', float((gb - gu * (ab > 0)).norm() / gref.norm()))
        idx = (gb - gref).abs().flatten().argsort(descending=True)[:5]
        for i in idx:
            i = int(i)
            print('      elem', np.unravel_index(i, gb.shape), 'hip', float(gb.flatten()[i]), 'ref', float(gref.flatten()[i]), 'act hip/ref', float(ab.flatten()[i]), float(tt.flatten()[i]), 'ungated ref', float(gu.flatten()[i]))
    flag = '  <-----' if ge > 1e-3 else ''
    print(f'op {n:3d} {str(op.get("name", "")):28s} {str(tuple(ab.shape)):22s} act err {best[1]:.1e} grad rel L2 {ge:.2e}{flag}')
