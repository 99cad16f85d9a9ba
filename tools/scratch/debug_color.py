import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch, spaa_oracle as so
from spaa_amd import differential_color_functions as dcf
z = np.load(os.path.join(ROOT, 'tests/golden/color_kat.npz'))
a0, b0 = torch.from_numpy(z['rgb_a']), torch.from_numpy(z['rgb_b'])
# stage 1: dE/dLab
la, lb = so.rgb2lab_diff(a0).detach().requires_grad_(True), so.rgb2lab_diff(b0).detach().requires_grad_(True)
so.ciede2000_diff(la, lb).sum().backward()
ga, gb = la.detach().clone().cuda().requires_grad_(True), lb.detach().clone().cuda().requires_grad_(True)
dcf.ciede2000_diff(ga, gb).sum().backward()
for nm, o, r in (('dE/dlab1', ga.grad.cpu(), la.grad), ('dE/dlab2', gb.grad.cpu(), lb.grad)):
    fin = torch.isfinite(r)
    d = ((o - r).abs() * fin)
    d[~fin] = 0
    i = int(d.flatten().argmax()); n, c, y, x = np.unravel_index(i, d.shape)
    print(nm, 'max abs diff', d.max().item(), 'scale', r[fin].abs().max().item(), 'at', (n, c, y, x), 'ours', o[n, :, y, x].tolist(), 'ref', r[n, :, y, x].tolist(),
          'lab1', la[n, :, y, x].tolist(), 'lab2', lb[n, :, y, x].tolist())
# stage 2: rgb2lab backward
r = torch.randn(2, 3, 16, 16)
xc = a0.clone().requires_grad_(True)
(so.rgb2lab_diff(xc) * r).sum().backward()
xg = a0.clone().cuda().requires_grad_(True)
(dcf.rgb2lab_diff(xg) * r.cuda()).sum().backward()
d = (xg.grad.cpu() - xc.grad).abs()
i = int(d.flatten().argmax()); n, c, y, x = np.unravel_index(i, d.shape)
print('rgb2lab bwd max abs diff', d.max().item(), 'scale', xc.grad.abs().max().item(), 'at', (n, c, y, x), a0[n, :, y, x].tolist(), xg.grad.cpu()[n, :, y, x].tolist(), xc.grad[n, :, y, x].tolist())
