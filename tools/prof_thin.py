"""Times the thin layers (conv6, conv1 dgrad, ResNet stem dgrad) under a given tile: python tools/prof_thin.py tile"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaa_amd import convplan as cp

tile = int(sys.argv[1])
torch.manual_seed(0)
B = 64
cases = [('conv6', 'f', 32, 3, 3, 1, 1, 256), ('conv1_dgrad', 'd', 3, 32, 3, 2, 1, 256), ('stem_dgrad', 'd', 3, 64, 7, 2, 3, 224)]
only = sys.argv[2] if len(sys.argv) > 2 else None
for name, kind, ci, co, k, s, p, hw in cases:
    if only and name != only:
        continue
    wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
    if kind == 'f':
        plan = cp.conv_fwd_plan(wt, torch.randn(co), s, p, 'cuda')
        x = torch.randn(B, hw, hw, ci, device='cuda')
        out = torch.zeros(B, hw, hw, 4, device='cuda')
    else:
        plan = cp.conv_dgrad_plan(wt, s, p, 'cuda')
        ho = (hw + 2 * p - k) // s + 1
        x = torch.randn(B, ho, ho, co, device='cuda')
        out = torch.zeros(B, hw, hw, 4, device='cuda')
    cp.FORCE_TILE = tile
    for _ in range(2):
        plan.run(x, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.run(x, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{name:12s} tile {tile}: {ms*1e3:8.1f} us  {plan.flops(B, out.shape[1], out.shape[2]) / ms / 1e9:6.1f} TFLOP/s', flush=True)
