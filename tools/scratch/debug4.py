import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F, spaa_oracle as so
from spaa_amd import _lib as lib, synthetic as syn, convplan as cp
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
st = AttackState(pc, clf, [204, 291], scene, 'camdE_caml2', setup, DEV)
st.forward_decide(True, 5, 0.9)
cp.PROFILE = []
st.clf.backward(st.g_logits)
torch.cuda.synchronize()
prof = list(cp.PROFILE)
cp.PROFILE = None
body = st.clf.body
ops = [op for op in body.ops if op['kind'] == 'conv']
op81, op80 = ops[81], ops[80]
print('op81', op81['name'], 'dgrad plan', op81['d'].name, 'tiles', [(p[0], p[5], p[1]) for p in prof if 'Mixed_7b.branch3x3dbl_2' in p[0]])
o81, o80 = op81['out'], op80['out']
g81 = o81.gbuf[..., o81.coff:o81.coff + o81.c].float()
a80 = o80.buf[..., o80.coff:o80.coff + o80.c].float()
g80 = o80.gbuf[..., o80.coff:o80.coff + o80.c].float()
name = op81['name']
wf, _ = cp.fold_bn(csd[name + '.conv.weight'], csd[name + '.bn.weight'], csd[name + '.bn.bias'], csd[name + '.bn.running_mean'],
                   csd[name + '.bn.running_var'], eps=1e-3)
ref = torch.nn.grad.conv2d_input((2, 448, 2, 2), wf.double(), g81.permute(0, 3, 1, 2).double().cpu(), 1, 1).float()
ref = ref * (a80.permute(0, 3, 1, 2).cpu() > 0)
got = g80.permute(0, 3, 1, 2).cpu()
d = (got - ref)
print('rel L2', (d.norm() / ref.norm()).item(), 'ref norm', ref.norm().item())
for c0 in range(0, 448, 64):
    print('channels', c0, c0 + 64, 'rel', (d[:, c0:c0 + 64].norm() / (ref[:, c0:c0 + 64].norm() + 1e-30)).item())
for b in range(2):
    for y in range(2):
        for x in range(2):
            print('pixel', b, y, x, 'rel', (d[b, :, y, x].norm() / (ref[b, :, y, x].norm() + 1e-30)).item())
gy = o81.gbuf
for tile in (0, 34, 234, 434, 36, 6):
    out = torch.zeros_like(o80.gbuf)
    cp.FORCE_TILE = tile
    op81['d'].run(gy, out, gate=o80.buf, in_coff=o81.coff, out_coff=o80.coff, gate_coff=o80.coff)
    cp.FORCE_TILE = 0
    got2 = out[..., o80.coff:o80.coff + o80.c].float().permute(0, 3, 1, 2).cpu()
    print('tile', tile, 'rel L2 vs torch', ((got2 - ref).norm() / ref.norm()).item())
print('cin_p', op81['d'].cin_p, 'cout', op81['d'].cout, 'coffs', o81.coff, o80.coff, 'buf shapes', tuple(o81.gbuf.shape), tuple(o80.gbuf.shape))
