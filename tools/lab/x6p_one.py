"""A few launches of transConv1 (128 -> 64, k3 s2, 64^2 -> 128^2, batch 64) on one tile for rocprofv3 --pmc passes
(PROF_SCRIPT=lab/x6p_one.py bash tools/pmc_conv.sh <tag> [tile] [fused 0/1])."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
B = 64
wt = torch.randn(128, 64, 3, 3) / 34
plan = cp.deconv_fwd_plan(wt, torch.randn(64), 2, 1, DEV, 'transConv1', fold=False)
x5 = torch.relu(torch.randn(B, 64, 64, 128, device=DEV)); x6 = torch.zeros(B, 128, 128, 64, device=DEV); r2 = torch.randn(B, 128, 128, 64, device=DEV)
m6 = torch.zeros(B, 128, 128, 16, dtype=torch.uint8, device=DEV)
cp.FORCE_TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 74
kw = dict(add=r2)
if len(sys.argv) > 2 and sys.argv[2] == '1':
    ws = torch.randn(64, 32, 1, 1) / 6
    plan.attach_second_source(ws, None)
    kw = dict(inp2=torch.relu(torch.randn(B, 128, 128, 32, device=DEV)))
for _ in range(4):
    plan.run(x5, x6, act=_lib.ACT_RELU, mask_out=m6, **kw)
torch.cuda.synchronize()
