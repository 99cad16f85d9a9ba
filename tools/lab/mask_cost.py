"""What the ReLU gate bytes cost in an epilogue: conv4_s (128 -> 256, 64 x 64, batch 64, Winograd) and layer1-type (64 -> 64, 56 x 56)
with and without `mask_out`, with and without a `gate_bits` read: python tools/lab/mask_cost.py"""
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


for ci, co, h in ((128, 256, 64), (64, 64, 56), (32, 64, 128)):
    w, b = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5), torch.randn(co)
    plan = cp.conv_fwd_plan(w, b, 1, 1, DEV, 'layer')
    x = torch.relu(torch.randn(B, h, h, ci, device=DEV))
    y = torch.zeros(B, h, h, co, device=DEV)
    m = torch.zeros(B, h, h, co // 4, dtype=torch.uint8, device=DEV)
    gb = torch.full((B, h, h, co // 4), 15, dtype=torch.uint8, device=DEV)
    t0 = timeit(lambda: plan.run(x, y, act=_lib.ACT_RELU))
    t1 = timeit(lambda: plan.run(x, y, act=_lib.ACT_RELU, mask_out=m))
    t2 = timeit(lambda: plan.run(x, y, gate_bits=gb))
    print(f'{ci} -> {co} at {h}^2 (tile {plan.wino.last_tile if plan.wino is not None else plan.last_tile}): bias + ReLU {t0:.1f} us, + mask_out {t1:.1f} us, gate_bits read instead {t2:.1f} us')
