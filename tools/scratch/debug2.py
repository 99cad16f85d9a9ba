import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F, spaa_oracle as so
from spaa_amd import _lib as lib, convplan as cp, synthetic as syn
from spaa_amd.classifier import Classifier
from spaa_amd.models import PCNet, WarpingNet, to_nchw, to_nhwc4
from spaa_amd.projector_based_attack import AttackState
from tapconv_emu import nhwc, nchw
DEV = 'cuda'
# (1) split-K determinism / mask equivalence
torch.manual_seed(234)
b, ci, co, h, w = 2, 64, 96, 19, 23
wt2 = torch.randn(48, co, 3, 3) / (co * 9) ** 0.5
dplan = cp.conv_dgrad_plan(wt2, 1, 1, DEV)
gy = nhwc(torch.randn(b, 48, h, w)).to(DEV)
act = torch.randn(b, h, w, co, device=DEV)
res = {}
for tile in (34, 234):
    cp.FORCE_TILE = tile
    outs = []
    for rep in range(3):
        g = torch.zeros(b, h, w, co, device=DEV)
        dplan.run(gy, g, gate=act)
        outs.append(g.clone())
    gm = torch.zeros(b, h, w, co, device=DEV)
    dplan.run(gy, gm, gate_bits=lib.pack_gate_mask(act))
    gn = torch.zeros(b, h, w, co, device=DEV)
    dplan.run(gy, gn)
    cp.FORCE_TILE = 0
    print('tile', tile, 'float-gate reruns equal:', torch.equal(outs[0], outs[1]), torch.equal(outs[1], outs[2]), 'mask == float:', torch.equal(gm, outs[0]),
          'ungated*gate == float-gated:', torch.equal(gn * (act > 0), outs[0]), 'max diff mask vs float', (gm - outs[0]).abs().max().item())
# (2) inception loop: where does the first iteration diverge?
sz, crop, insz = (128, 128), (120, 120), (107, 107)
sd = syn.pcnet_state_dict(12, cam_sz=sz, mask='ones')
pc = PCNet(sd['mask'], WarpingNet(out_size=sz)); pc.load_state_dict(sd); pc = pc.to(DEV)
for body, csd in (('inception_v3', syn.inception_v3_state_dict(4, logit_gain=20.0)), ('resnet18', syn.resnet18_state_dict(2, logit_gain=20.0))):
    clf = Classifier(body, DEV, state_dict=csd, input_sz=insz)
    scene = syn.scenes(13, 1, sz)
    setup = dict(classifier_crop_sz=crop, prj_brightness=0.5, prj_im_sz=sz)
    targets = [204, 291]
    st = AttackState(pc, clf, targets, scene, 'camdE_caml2', setup, DEV)
    st.forward_decide(True, 5, 0.9)
    y = to_nchw(st.eng.a['Y']).cpu()
    g_y = to_nchw(st.clf.backward(st.g_logits)).cpu()
    yc = y.clone().requires_grad_(True)
    raw, p, idx = so.OracleClassifier(body, csd, input_sz=insz)(yc, crop)
    (-raw[torch.arange(2), torch.tensor(targets)]).mean().backward()
    d = (g_y - yc.grad)
    print(body, 'g_logits', st.g_logits.abs().sum(1).tolist(), 'state', st.state.cpu().tolist(), 'adv grad at cam image: rel L2', (d.norm() / yc.grad.norm()).item(), 'rel Linf', (d.abs().max() / yc.grad.abs().max()).item(),
          'norms', g_y.norm().item(), yc.grad.norm().item(), 'logit err', (st.stats[:, 6].cpu() - raw[torch.arange(2), torch.tensor(targets)].detach()).abs().max().item())
