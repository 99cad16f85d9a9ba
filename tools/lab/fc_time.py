"""The classifiers' last layer (512 / 2048 / 4096 -> 1000 at batch 64) and its input gradient: csrc/linear_small.hip against the
1 x 1 convolution tiles (SPAA_SMALL_LINEAR=0 form): python tools/lab/fc_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for k in (512, 2048, 4096):
    w, b = torch.randn(1000, k) / k ** 0.5, torch.randn(1000)
    x, g = torch.randn(64, 1, 1, k, device=DEV), torch.randn(64, 1, 1, 1000, device=DEV)
    y, gx = torch.zeros(64, 1, 1, 1000, device=DEV), torch.zeros(64, 1, 1, k, device=DEV)
    f, d = cp.linear_fwd_plan(w, b, DEV, 'fc'), cp.linear_dgrad_plan(w, DEV, 'fc_dgrad')
    row = []
    for small in (True, False):
        cp.SMALL_LINEAR = small
        row += [timeit(lambda: f.run(x, y)), timeit(lambda: d.run(g, gx))]
    print(f'{k} -> 1000 at batch 64: forward {row[0]:.1f} us (tiles: {row[2]:.1f}), input gradient {row[1]:.1f} us (tiles: {row[3]:.1f})')
