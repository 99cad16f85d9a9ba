"""Timing lab for the byte-heavy PCNet layers (B = 64, 256 x 256 geometry): tiles x epilogue variants, one process."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
B = 64


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def lab(name, plan, x, out, tiles, add=None, mask=None, gate_bits=None, act=_lib.ACT_RELU):
    for tile in tiles:
        cp.FORCE_TILE = tile
        res = []
        for label, kw in (('bare', {}), ('+add', dict(add=add)), ('+add+mask', dict(add=add, mask_out=mask)), ('+gatebits', dict(gate_bits=gate_bits)),
                          ('+add+gatebits', dict(add=add, gate_bits=gate_bits))):
            if any(v is None for v in kw.values()):
                continue
            try:
                res.append(f'{label} {timeit(lambda: plan.run(x, out, act=act if "gate_bits" not in kw else 0, **kw)):.0f}')
            except Exception as e:  # a tile that does not apply
                res.append(f'{label} n/a')
        print(f'{name} tile {tile}: ' + ' | '.join(res), flush=True)
    cp.FORCE_TILE = 0


torch.manual_seed(0)
# transConv1: 128 -> 64, k3 s2 p1 op1, 64x64 -> 128x128
wt = torch.randn(128, 64, 3, 3) / (128 * 9) ** .5
x5 = torch.relu(torch.randn(B, 64, 64, 128, device=DEV))
x6 = torch.zeros(B, 128, 128, 64, device=DEV)
r2 = torch.randn(B, 128, 128, 64, device=DEV)
m6 = torch.zeros(B, 128, 128, 16, dtype=torch.uint8, device=DEV)
for fold in (None, True):
    plan = cp.deconv_fwd_plan(wt, torch.randn(64), 2, 1, DEV, 'transConv1', fold=fold)
    lab(f'transConv1 fold={fold}', plan, x5, x6, [16, 44, 42, 34, 39] if not fold else [34, 39, 36, 25, 48], add=r2, mask=m6)
# its input gradient: conv 64 -> 128, 3x3 s2, 128x128 -> 64x64
p6 = torch.randn(B, 128, 128, 64, device=DEV)
p5 = torch.zeros(B, 64, 64, 128, device=DEV)
m5 = (torch.rand(B, 64, 64, 32, device=DEV) * 16).to(torch.uint8)
dplan = cp.deconv_dgrad_plan(wt, 2, 1, DEV, 'transConv1_dgrad')
lab('transConv1_dgrad', dplan, p6, p5, [48, 34, 39, 25], gate_bits=m5, act=0)
# conv2 dgrad: 64 (64x64) -> 32 (128x128), 3x3 s2
w2 = torch.randn(64, 32, 3, 3) / (32 * 9) ** .5
p2 = torch.randn(B, 64, 64, 64, device=DEV)
p1 = torch.zeros(B, 128, 128, 32, device=DEV)
t1 = torch.randn(B, 128, 128, 32, device=DEV)
m1 = (torch.rand(B, 128, 128, 8, device=DEV) * 16).to(torch.uint8)
for fold in (True, False):
    d2 = cp.conv_dgrad_plan(w2, 2, 1, DEV, 'conv2_dgrad', fold=fold)
    lab(f'conv2_dgrad fold={fold}', d2, p2, p1, [39, 34, 25, 48] if fold else [37, 30, 41, 53, 16, 18], add=t1, gate_bits=m1, act=0)
# skipConv2: 1x1 32 -> 64 at 128x128, and its gradient
ws = torch.randn(64, 32, 1, 1) / 32 ** .5
x1 = torch.relu(torch.randn(B, 128, 128, 32, device=DEV))
lab('skipConv2', cp.conv_fwd_plan(ws, torch.randn(64), 1, 0, DEV, 'skipConv2'), x1, x6, [16, 36, 27, 49], act=0)
lab('skipConv2_dgrad', cp.conv_dgrad_plan(ws, 1, 0, DEV, 'skipConv2_dgrad'), p6, p1, [53, 37, 30, 16], act=0)
