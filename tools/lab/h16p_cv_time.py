"""3 x 3 / stride-1 layers of the fp16-storage classifiers on small maps at batch 64 (ResNet-18 layer3 / layer4, VGG-16's last block, the
28 x 28 blocks for comparison): the patch-staged fp16 kernel's canvas / K-range form (csrc/tapconv_h16p.hip CV) under its own plan and
under forced (N tile, K ranges), against the implicit-GEMM fp16 tiles the rules picked before (SPAA_DEFAULT_DISABLE=h16pcv)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
torch.manual_seed(0)
def t(fn, n=100):
    for _ in range(40): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [('resnet layer3', 256, 256, 14, 14), ('resnet layer4', 512, 512, 7, 7), ('vgg block5', 512, 512, 14, 14), ('vgg block4', 512, 512, 28, 28),
          ('resnet layer2 (224)', 128, 128, 28, 28), ('inception 35x35', 96, 96, 35, 35), ('inception 8x8', 384, 384, 8, 8)]
for name, ci, co, h, w in shapes:
    b = 64
    wt = (torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5).half().float()
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV, name)
    x = torch.randn(b, h, w, ci, device=DEV).half()
    add = torch.randn(b, h, w, co, device=DEV).half()
    out = torch.zeros(b, h, w, co, device=DEV, dtype=torch.float16)
    mask = torch.zeros(b, h, w, co // 4, device=DEV, dtype=torch.uint8)
    run = lambda: plan.run(x, out, add=add, act=_lib.ACT_RELU, mask_out=mask)
    res = []
    cp.DEFAULT_DISABLE.add('h16pcv')
    run()
    t(run, 300)       # (clocks up)
    res.append((f'before: {cp.TILE_NAMES.get(plan.last_tile, plan.last_tile)}', t(run)))
    cp.DEFAULT_DISABLE.discard('h16pcv')
    for cv in [(0, 0), (64, 1), (64, 2), (64, 4), (64, 8), (128, 1), (128, 2), (128, 4), (128, 8)]:
        if cv[1] > 1 and ci // 32 < 2 * cv[1]:
            continue
        cp.H16P_CV = cv
        run()
        if plan.last_tile != 68 or not hasattr(plan, 'last_h16p_plan'):
            res.append((f'{cv}: tile {plan.last_tile}', t(run)))
            continue
        p = plan.last_h16p_plan
        res.append((f'{"plan" if cv == (0, 0) else cv}: bn {p[0]} ks {p[1]} canvas {p[2]} ({p[3]}x{p[4]}) wgs {p[5]}', t(run)))
    cp.H16P_CV = (0, 0)
    cp.DEFAULT_DISABLE.add('h16pcv')
    run()
    res.append((f'before (again): {cp.TILE_NAMES.get(plan.last_tile, plan.last_tile)}', t(run)))
    cp.DEFAULT_DISABLE.discard('h16pcv')
    fl = 2 * b * h * w * ci * co * 9
    print(f'{name} {ci}->{co} {h}x{w}: ' + '\n    '.join(f'{k_}: {v:.1f} us ({fl / v / 1e6:.0f} TF)' for k_, v in res), flush=True)
