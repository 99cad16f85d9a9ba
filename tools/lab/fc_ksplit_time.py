"""VGG-16 classifier.0 (25088 -> 4096, batch 64, fp16 storage) on the implicit-GEMM fp16 tile under forced K-split factors: how the
205 MB weight stream's rate depends on the number of concurrent K ranges."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
torch.manual_seed(0)
def t(fn, n=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# a second big tensor to flush the 256 MB MALL between launches
flush = torch.zeros(96 * 1024 * 1024, device=DEV)
for k, n in [(25088, 4096), (4096, 4096)]:
    w = (torch.randn(n, k) / k ** 0.5)
    plan = cp.linear_fwd_plan(w, torch.randn(n), DEV, 'fc')
    x = torch.randn(64, 1, 1, k, device=DEV).half()
    out = torch.zeros(64, 1, 1, n, device=DEV, dtype=torch.float16)
    for ks in (1, 2, 4, 8, 16, 32):
        cp.FORCE_KSPLIT = ks
        def run():
            flush.add_(1.0)
            plan.run(x, out, act=_lib.ACT_RELU)
        def run0():
            flush.add_(1.0)
        tt = t(run) - t(run0)
        print(f'{k} -> {n}: K ranges {ks:2d} tile {cp.TILE_NAMES.get(plan.last_tile)}: {tt:6.1f} us  ({n * k * 2 / tt / 1e6:.2f} TB/s of weights)', flush=True)
    cp.FORCE_KSPLIT = 0
