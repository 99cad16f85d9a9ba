"""Runs one tapconv layer shape repeatedly (for rocprofv3 --pmc): python tools/prof_conv.py cin cout k stride H W B tile [reps]"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaa_amd import convplan as cp

cin, cout, k, s, H, W, B, tile = [int(v) for v in sys.argv[1:9]]
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 5
torch.manual_seed(0)
wt = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
plan = cp.conv_fwd_plan(wt, torch.randn(cout), s, k // 2, 'cuda')
x = torch.randn(B, H, W, (cin + 3) // 4 * 4, device='cuda')
if os.environ.get('PROF_ZERO_INPUT'):
    x.zero_()  # same instruction stream, no operand toggling: separates clock/power effects from cycle counts
ho, wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
out = torch.zeros(B, ho, wo, (cout + 3) // 4 * 4, device='cuda')
cp.FORCE_TILE = tile
for _ in range(2):
    plan.run(x, out, act=1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    plan.run(x, out, act=1)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f'tile {tile}: {ms*1e3:.1f} us  {plan.flops(B, ho, wo) / ms / 1e9:.1f} TFLOP/s')
