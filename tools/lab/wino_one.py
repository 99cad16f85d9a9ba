"""A few launches of one Winograd-shaped layer for rocprofv3 --pmc passes: wino_one.py [tile] [dbg bits] [cin] [cout]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
ci = int(sys.argv[3]) if len(sys.argv) > 3 else 128
co = int(sys.argv[4]) if len(sys.argv) > 4 else 256
wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** .5
plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV)
x = torch.relu(torch.randn(64, 64, 64, ci, device=DEV))
out = torch.zeros(64, 64, 64, co, device=DEV)
cp.FORCE_TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 70
cp.DEBUG_WINO = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # (bits 2..: DBG << 2; 256: the round-2 kernel)
for _ in range(4):
    plan.run(x, out)
torch.cuda.synchronize()
