"""Times the few-input-channel layers (conv6 dgrad, conv1, conv1_s) under a given tile: python tools/prof_smallcin.py tile"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaa_amd import convplan as cp

tile = int(sys.argv[1])
torch.manual_seed(0)
B = 64
for name, kind, ci, co, k, s, p, hw in [('conv6_dgrad', 'd', 32, 3, 3, 1, 1, 256), ('conv1', 'f', 3, 32, 3, 2, 1, 256),
                                        ('conv1_s', 'f', 6, 32, 3, 2, 1, 256)]:
    if kind == 'f':
        wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
        plan = cp.conv_fwd_plan(wt, torch.randn(co), s, p, 'cuda')
        x = torch.randn(B, hw, hw, (ci + 3) // 4 * 4, device='cuda')
        ho = (hw + 2 * p - k) // s + 1
        out = torch.zeros(B, ho, ho, co, device='cuda')
        gate = None
    else:  # gradient w.r.t. the 32-channel input of a 32 -> 3 conv: 4 (padded 3) channels in, 32 out, ReLU gate
        wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
        plan = cp.conv_dgrad_plan(wt, s, p, 'cuda')
        x = torch.randn(B, hw, hw, 4, device='cuda')
        out = torch.zeros(B, hw, hw, ci, device='cuda')
        gate = torch.randn(B, hw, hw, ci, device='cuda')
    cp.FORCE_TILE = tile
    for _ in range(2):
        plan.run(x, out, gate=gate)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.run(x, out, gate=gate)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{name:12s} tile {tile}: {ms*1e3:8.1f} us  {plan.flops(B, out.shape[1], out.shape[2]) / ms / 1e9:6.1f} TFLOP/s', flush=True)
