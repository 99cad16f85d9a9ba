"""Small-image 3x3 / s1 layers (ResNet-18 layer1-4, VGG-16 14 x 14, Inception-v3 35 x 35) at batch 64: the Winograd kernel's canvas /
K-range forms against its image-aligned form and the direct bf16x6 tiles, us per launch (plain / residual + ReLU + byte mask)."""
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
SHAPES = [(256, 256, 14, 14, 64), (512, 512, 7, 7, 64), (128, 128, 28, 28, 64), (64, 64, 56, 56, 64), (64, 96, 35, 35, 64), (96, 96, 35, 35, 64),
          (512, 512, 14, 14, 64), (512, 512, 28, 28, 64)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for ci, co, h, w, b in SHAPES:
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV)
    x = torch.randn(b, h, w, ci, device=DEV)
    out = torch.zeros(b, h, w, co, device=DEV)
    add = torch.randn(b, h, w, co, device=DEV)
    mask = torch.zeros(b, h, w, co // 4, dtype=torch.uint8, device=DEV)
    res = {}
    def run(label, tile, nocanvas=0):
        cp.FORCE_TILE, cp.DEBUG_WINO_NOCANVAS = tile, nocanvas
        try:
            a = t(lambda: plan.run(x, out))
            pl = getattr(plan.wino, 'last_wino_plan', None) if tile % 100 in (70, 71, 73) else None
            bb = t(lambda: plan.run(x, out, add=add, act=_lib.ACT_RELU, mask_out=mask))
            used = plan.wino.last_tile if (tile % 100 in (70, 71, 73) and plan.wino is not None) else plan.last_tile
            res[label] = (a, bb, used, pl)
        except Exception as e:   # noqa
            res[label] = (float('nan'), float('nan'), -1, str(e)[:40])
        cp.FORCE_TILE, cp.DEBUG_WINO_NOCANVAS = 0, 0
    run('tuned', 0)
    for tl in (34, 234, 434, 248, 448, 948):
        run(f'direct{tl}', tl)
    run('wino_plain', 170, 1)
    run('wino64_plain', 171, 1)
    run('wino_auto', 70)
    run('wino_4wave_plain', 173, 1)
    run('wino_4wave_auto', 73)
    for ks in (2, 4):
        run(f'wino_4wave_ks{ks}', 73 + 100 * ks, 2)
    for ks in (1, 2, 4, 8):
        run(f'wino128_ks{ks}', 70 + 100 * ks, 2)
        run(f'wino64_ks{ks}', 71 + 100 * ks, 2)
    fl = 2 * b * h * w * 9 * ci * co
    print(f'== {ci}->{co} {h}x{w} B{b} ({fl / 1e9:.1f} GF)')
    for k, v in res.items():
        print(f'   {k:16s} {v[0]:7.1f} / {v[1]:7.1f} us  {fl / v[0] / 1e6:6.0f} TF  tile {v[2]} plan {v[3]}', flush=True)
