"""Stride-2 fractional layers at batch 64: the patch-staged bf16x6 kernel (tile 74, csrc/tapconv_x6p.hip), alone and with its fused
1 x 1 second source, against the tiles the tune table holds for them + the separate 1 x 1 launch; us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
torch.manual_seed(0)
B = 64
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def case(name, plan, x, out, epi, plan2=None, x2=None, w2=None, tiles=(0, 16, 39, 42)):
    res = {}
    for tile in tiles + (74,):
        cp.FORCE_TILE = tile
        try:
            res[tile] = t(lambda: plan.run(x, out, **epi))
            res[tile] = (res[tile], plan.last_tile)
        except Exception as e:   # noqa
            res[tile] = (float('nan'), str(e)[:30])
    cp.FORCE_TILE = 0
    line = f'{name}: ' + '  '.join(f'{k}: {v[0]:.0f} (ran {v[1]})' for k, v in res.items())
    if plan2 is not None:
        tmp = torch.zeros_like(out)
        t2 = t(lambda: plan2.run(x2, tmp))
        plan.attach_second_source(w2, None)
        cp.FORCE_TILE = 74
        epi2 = {k: v for k, v in epi.items() if k != 'add'}
        tf = t(lambda: plan.run(x, out, inp2=x2, **epi2))
        cp.FORCE_TILE = 0
        line += f' | separate 1x1: {t2:.0f}  fused (74 + second source, no residual read): {tf:.0f}'
    print(line, flush=True)
# transConv1 128 -> 64, 64^2 -> 128^2 (+ skipConv2 32 -> 64 at 128^2)
wt = torch.randn(128, 64, 3, 3) / 34
x5 = torch.relu(torch.randn(B, 64, 64, 128, device=DEV)); x6 = torch.zeros(B, 128, 128, 64, device=DEV); r2 = torch.randn(B, 128, 128, 64, device=DEV)
m6 = torch.zeros(B, 128, 128, 16, dtype=torch.uint8, device=DEV)
ws = torch.randn(64, 32, 1, 1) / 6; x1 = torch.relu(torch.randn(B, 128, 128, 32, device=DEV))
case('transConv1 +add+relu+mask', cp.deconv_fwd_plan(wt, torch.randn(64), 2, 1, DEV, 'transConv1', fold=False), x5, x6,
     dict(add=r2, act=_lib.ACT_RELU, mask_out=m6), cp.conv_fwd_plan(ws, torch.randn(64), 1, 0, DEV, 'skipConv2'), x1, ws)
# conv2_dgrad: 64 (64^2) -> 32 (128^2) (+ skipConv2_dgrad 64 -> 32 at 128^2)
w2c = torch.randn(64, 32, 3, 3) / 17
p2 = torch.randn(B, 64, 64, 64, device=DEV); p1 = torch.zeros(B, 128, 128, 32, device=DEV); t1 = torch.randn(B, 128, 128, 32, device=DEV)
m1 = (torch.rand(B, 128, 128, 8, device=DEV) * 16).to(torch.uint8)
p6 = torch.randn(B, 128, 128, 64, device=DEV)
case('conv2_dgrad +add+gatebits', cp.conv_dgrad_plan(w2c, 2, 1, DEV, 'conv2_dgrad', fold=False), p2, p1, dict(add=t1, gate_bits=m1),
     cp.conv_dgrad_plan(ws, 1, 0, DEV, 'skipConv2_dgrad'), p6, ws[:, :, 0, 0].t().contiguous(), tiles=(0, 37, 30, 41))
case('conv2_dgrad folded (tune)', cp.conv_dgrad_plan(w2c, 2, 1, DEV, 'conv2_dgrad', fold=True), p2, p1, dict(add=t1, gate_bits=m1), tiles=(0, 39))
# ResNet-18 stride-2 input gradients: layer2.0.conv1 (64 -> 128 at 56 -> 28), layer3.0.conv1, layer4.0.conv1
for ci, co, hw in ((128, 64, 28), (256, 128, 14), (512, 256, 7)):
    w = torch.randn(ci, co, 3, 3) / (ci * 2.25) ** .5
    g = torch.randn(B, hw, hw, ci, device=DEV); o = torch.zeros(B, 2 * hw, 2 * hw, co, device=DEV)
    mk = (torch.rand(B, 2 * hw, 2 * hw, co // 4, device=DEV) * 16).to(torch.uint8)
    case(f'resnet s2 dgrad {ci}->{co} {hw}->{2 * hw} +gatebits', cp.conv_dgrad_plan(w, 2, 1, DEV, f'l{ci}', fold=False), g, o, dict(gate_bits=mk), tiles=(0, 16, 34, 44))
