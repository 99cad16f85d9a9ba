"""Time every applicable tile on the transposed-conv layers of ShadingNet (B=64, 256x256 camera)."""
import sys, torch
sys.path.insert(0, '.')
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
torch.manual_seed(0)
def bench(plan, x, out, tiles, **kw):
    res = {}
    for t in tiles:
        cp.FORCE_TILE = t
        try:
            for _ in range(2):
                plan.run(x, out, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                plan.run(x, out, **kw)
            e1.record()
            torch.cuda.synchronize()
            res[t] = e0.elapsed_time(e1) / 8 * 1e3
        except Exception as e:
            res[t] = str(e)[:30]
        cp.FORCE_TILE = 0
    return res
tiles = [0, 1, 2, 4, 15, 16, 17, 18, 20, 21, 25, 26, 27, 30, 31, 32, 33, 34, 35, 36, 37, 39, 40, 42, 44, 48, 49, 50, 51, 52]
R = _lib.ACT_RELU
# transConv1: 128 -> 64, k3 s2 p1 op1, 64x64 -> 128x128, + residual
w = torch.randn(128, 64, 3, 3) / 34
p = cp.deconv_fwd_plan(w, torch.randn(64), 2, 1, DEV, 'tc1')
x = torch.relu(torch.randn(64, 64, 64, 128, device=DEV)); out = torch.zeros(64, 128, 128, 64, device=DEV); add = torch.randn(64, 128, 128, 64, device=DEV)
print('transConv1', {k: (round(v) if isinstance(v, float) else v) for k, v in bench(p, x, out, tiles, add=add, act=R).items()}, flush=True)
pd = cp.deconv_dgrad_plan(w, 2, 1, DEV, 'tc1d')
gx = torch.zeros(64, 64, 64, 128, device=DEV)
print('transConv1_dgrad', {k: (round(v) if isinstance(v, float) else v) for k, v in bench(pd, add, gx, tiles).items()}, flush=True)
del x, out, add, gx
# transConv2: 64 -> 32, k2 s2, 128x128 -> 256x256
w = torch.randn(64, 32, 2, 2) / 16
p = cp.deconv_fwd_plan(w, torch.randn(32), 2, 0, DEV, 'tc2')
x = torch.relu(torch.randn(64, 128, 128, 64, device=DEV)); out = torch.zeros(64, 256, 256, 32, device=DEV)
print('transConv2 (fold=%d)' % p.nfold, {k: (round(v) if isinstance(v, float) else v) for k, v in bench(p, x, out, tiles, act=R).items()}, flush=True)
p2 = cp.deconv_fwd_plan(w, torch.randn(32), 2, 0, DEV, 'tc2', fold=False)
print('transConv2 unfolded', {k: (round(v) if isinstance(v, float) else v) for k, v in bench(p2, x, out, tiles, act=R).items()}, flush=True)
pd = cp.deconv_dgrad_plan(w, 2, 0, DEV, 'tc2d')
print('transConv2_dgrad', {k: (round(v) if isinstance(v, float) else v) for k, v in bench(pd, out, x, tiles).items()}, flush=True)
