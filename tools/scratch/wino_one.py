import sys, torch
sys.path.insert(0, '.')
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
ci, co = 128, 256
wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** .5
plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV)
x = torch.relu(torch.randn(64, 64, 64, ci, device=DEV))
out = torch.zeros(64, 64, 64, co, device=DEV)
cp.FORCE_TILE = int(sys.argv[1]) if len(sys.argv) > 1 else 70
for _ in range(4):
    plan.run(x, out)
torch.cuda.synchronize()
