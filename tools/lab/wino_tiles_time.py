"""The 64 x 64 PCNet layers at batch 64 in fp32: Winograd tile 70 (16 x 32 x 128 / 64, one workgroup per CU), 71 (64-wide), 73 (8 x 32 x 64, four
waves, two workgroups per CU), with residual + ReLU + byte mask as in the loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
torch.manual_seed(0)
def t(fn, n=40):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, ci, co in [('conv3 64->128', 64, 128), ('conv4 128->256', 128, 256), ('conv4^T 256->128', 256, 128), ('conv3^T 128->64', 128, 64)]:
    b, h, w = 64, 64, 64
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV, name)
    x = torch.randn(b, h, w, ci, device=DEV)
    add = torch.randn(b, h, w, co, device=DEV)
    out = torch.zeros(b, h, w, co, device=DEV)
    mask = torch.zeros(b, h, w, co // 4, device=DEV, dtype=torch.uint8)
    run = lambda: plan.run(x, out, add=add, act=_lib.ACT_RELU, mask_out=mask)
    run(); t(run, 100)
    res = []
    for tile in (0, 70, 71, 73, 0):
        cp.FORCE_TILE = tile
        run()
        wn = plan.wino
        res.append((f'{tile} -> {getattr(wn, "last_tile", None) if wn is not None else None} / {getattr(plan, "last_tile", None)} plan {getattr(wn, "last_wino_plan", None)}', t(run)))
    cp.FORCE_TILE = 0
    fl = 2 * b * h * w * ci * co * 9
    print(f'{name}: ' + '\n    '.join(f'{k_}: {v:.1f} us ({fl / v / 1e6:.0f} TF)' for k_, v in res), flush=True)
