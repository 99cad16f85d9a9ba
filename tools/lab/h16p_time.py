"""fp16-storage 3x3 layers: patch-staged kernel (tile 68) against the implicit-GEMM h16 tiles, us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
torch.manual_seed(0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
SHAPES = [(512, 512, 7, 7, 64), (256, 256, 14, 14, 64), (128, 128, 28, 28, 64)] if len(sys.argv) > 1 else [(128, 256, 64, 64, 64), (256, 128, 64, 64, 64), (128, 128, 64, 64, 64), (64, 64, 128, 128, 64), (256, 256, 64, 64, 64), (128, 128, 56, 56, 64), (64, 64, 56, 56, 64)]
for ci, co, h, w, b in SHAPES:
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV)
    x = torch.randn(b, h, w, ci, device=DEV).half()
    out = torch.zeros(b, h, w, co, device=DEV, dtype=torch.float16)
    add = torch.randn(b, h, w, co, device=DEV).half()
    mask = torch.zeros(b, h, w, co // 4, dtype=torch.uint8, device=DEV)
    res = {}
    for tile in ((60, 61, 62, 64, 68) if len(sys.argv) > 1 else (60, 64, 65, 68)):
        cp.FORCE_TILE = tile
        res[tile] = (t(lambda: plan.run(x, out)), t(lambda: plan.run(x, out, add=add, act=_lib.ACT_RELU, mask_out=mask)))
    cp.FORCE_TILE = 0
    fl = 2 * b * h * w * 9 * ci * co
    print(f'{ci}->{co} {h}x{w} B{b}: ' + '  '.join(f'{k}: {v[0]:.0f}/{v[1]:.0f} us ({fl / v[0] / 1e6:.0f} TF)' for k, v in res.items()), flush=True)
