import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from spaa_amd import _lib as lib, convplan as cp
from tapconv_emu import nhwc, nchw
DEV = 'cuda'
def say(*a):
    print(*a, flush=True)
def _h(x): return x.half().float()
which = sys.argv[1]
torch.manual_seed(31)
cases = {'c1': (64, 96, 3, 1, 19, 23, 2), 'c2': (32, 64, 3, 2, 22, 18, 3), 'c6': (64, 64, 5, 1, 11, 10, 2)}
ci, co, k, s, h, w, b = cases[which]
x = _h(torch.randn(b, ci, h, w)); wt = _h(torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5); bias = torch.randn(co)
y = F.conv2d(x.double(), wt.double(), bias.double(), s, k // 2).float()
ho, wo = y.shape[2:]
plan = cp.conv_fwd_plan(wt, bias, s, k // 2, DEV)
out = torch.zeros(b, ho, wo, co, device=DEV, dtype=torch.float16)
say(which, 'fwd launch'); plan.run(nhwc(x).half().to(DEV), out, act=lib.ACT_RELU); torch.cuda.synchronize()
say(which, 'fwd ok', ((nchw(out.float().cpu(), co) - F.relu(y)).abs().max() / y.abs().max()).item())
gy = _h(torch.randn(b, co, ho, wo))
gx_ref = torch.nn.grad.conv2d_input((b, ci, h, w), wt.double(), gy.double(), s, k // 2).float()
dplan = cp.conv_dgrad_plan(wt, s, k // 2, DEV)
gx = torch.zeros(b, h, w, ci, device=DEV, dtype=torch.float16)
gin = nhwc(gy, dplan.cin_p).half().to(DEV)
say(which, 'dgrad launch, no gate', [(c['ntaps'], c['K']) for c in dplan.cls]); dplan.run(gin, gx); torch.cuda.synchronize()
say(which, 'dgrad ok', ((nchw(gx.float().cpu(), ci) - gx_ref).abs().max() / gx_ref.abs().max()).item())
act = torch.randn(b, h, w, ci, device=DEV)
m = lib.pack_gate_mask(act); torch.cuda.synchronize()
say(which, 'dgrad launch, gate bits', tuple(m.shape)); dplan.run(gin, gx, gate_bits=m); torch.cuda.synchronize()
say(which, 'dgrad+bits ok', ((nchw(gx.float().cpu(), ci) - gx_ref * (nchw(act.cpu(), ci) > 0)).abs().max() / gx_ref.abs().max()).item())
