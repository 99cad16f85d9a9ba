import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.nn.functional as F, spaa_oracle as so
from spaa_amd import _lib, synthetic as syn
from spaa_amd.classifier import Classifier
from tapconv_emu import nhwc, nchw
DEV = 'cuda'
# (1) generic max-pool forward/backward vs torch, input = ReLU output
for (k, s, p, h, w, c) in [(3, 2, 0, 17, 19, 8), (3, 2, 1, 16, 16, 64), (2, 2, 0, 12, 14, 4)]:
    torch.manual_seed(k + h)
    pre = torch.randn(2, c, h, w, requires_grad=True)
    x = F.relu(pre)
    y = F.max_pool2d(x, k, s, p)
    gy = torch.randn_like(y)
    y.backward(gy)
    ho, wo = y.shape[2:]
    xin = nhwc(x.detach()).to(DEV)
    out = torch.zeros(2, ho, wo, c, device=DEV)
    arg = torch.zeros(2, ho, wo, c, dtype=torch.uint8, device=DEV)
    _lib.call('spaa_maxpool_fwd', _lib.ptr(xin), _lib.ptr(out), _lib.ptr(arg), 2, h, w, c, ho, wo, k, s, p, c, 0)
    gin = torch.zeros(2, h, w, c, device=DEV)
    _lib.call('spaa_maxpool_bwd', _lib.ptr(nhwc(gy).to(DEV)), _lib.ptr(arg), 1, _lib.ptr(gin), 2, h, w, c, ho, wo, k, s, p, c, 0)
    print('maxpool', (k, s, p), 'fwd', (nchw(out.cpu()) - y).abs().max().item(), 'bwd(relu gate)', (nchw(gin.cpu()) - pre.grad).abs().max().item())
# (2) bare Inception-v3 gradient
csd = syn.inception_v3_state_dict(4, logit_gain=20.0)
for (h, crop, insz, b) in [(128, (120, 120), (107, 107), 2)]:
    torch.manual_seed(5)
    im = torch.rand(b, 3, h, h, requires_grad=True)
    raw, p, idx = so.OracleClassifier('inception_v3', csd, input_sz=insz)(im, crop)
    r = torch.randn(b, 1000)
    (raw * r).sum().backward()
    clf = Classifier('inception_v3', DEV, state_dict=csd, input_sz=insz)
    im2 = im.detach().clone().to(DEV).requires_grad_(True)
    raw2, p2, idx2 = clf(im2, crop)
    (raw2 * r.to(DEV)).sum().backward()
    g, gr = im2.grad.cpu().double(), im.grad.double()
    print('inception grad rel L2', ((g - gr).norm() / gr.norm()).item(), 'rel Linf', ((g - gr).abs().max() / gr.abs().max()).item(),
          'frac > 1e-3', ((g - gr).abs() > 1e-3 * gr.abs().max()).float().mean().item(), 'logits', ((raw2.cpu() - raw).abs().max() / raw.abs().max()).item())
    # one-hot cotangent on the target logit only (what the attack back-propagates)
    im.grad = None; im2.grad = None
    raw, _, _ = so.OracleClassifier('inception_v3', csd, input_sz=insz)(im, crop)
    raw[:, 204].sum().backward()
    raw2, _, _ = clf(im2, crop)
    raw2[:, 204].sum().backward()
    g, gr = im2.grad.cpu().double(), im.grad.double()
    print('one-hot cotangent: rel L2', ((g - gr).norm() / gr.norm()).item(), 'frac > 1e-3', ((g - gr).abs() > 1e-3 * gr.abs().max()).float().mean().item())
