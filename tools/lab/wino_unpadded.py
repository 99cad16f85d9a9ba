"""Inception-v3's unpadded 3x3 layer Conv2d_4a (80 -> 192 at 73 x 73 -> 71 x 71, batch 64): its input gradient (192 -> 80, pad 2) on the
Winograd kernel against the tuned direct kernel: python tools/lab/wino_unpadded.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
B = 64


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


w = torch.randn(192, 80, 3, 3) / 27
d = cp.conv_dgrad_plan(w, 1, 0, DEV, 'Conv2d_4a_3x3_dgrad')
g = torch.randn(B, 71, 71, 192, device=DEV)
gx = torch.zeros(B, 73, 73, 80, device=DEV)
gate = torch.zeros(B, 73, 73, 20, dtype=torch.uint8, device=DEV).fill_(15)
res = {}
for t in (0, 70, 71, 73, 34):
    cp.FORCE_TILE = t
    res[t] = (timeit(lambda: d.run(g, gx, gate_bits=gate)), d.last_tile if t not in (70, 71, 73) else d.wino.last_tile)
cp.FORCE_TILE = 0
print('Conv2d_4a_3x3 input gradient (192 -> 80, 71^2 -> 73^2, batch 64):', {k: (round(v[0], 1), v[1]) for k, v in res.items()})
