import sys, torch
sys.path.insert(0, '.')
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
for ci, co, hw in [(128, 64, 64), (64, 64, 56)]:
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** .5
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV)
    x = torch.relu(torch.randn(64, hw, hw, ci, device=DEV))
    out = torch.zeros(64, hw, hw, co, device=DEV)
    cp.FORCE_TILE = 70
    for dbg in [0, 1, 2, 3, 0, 1, 2, 3]:
        cp.DEBUG_WINO = dbg
        for _ in range(3):
            plan.run(x, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.run(x, out)
        e1.record()
        torch.cuda.synchronize()
        print(f'{ci}->{co} {hw}: kernel variant {3 ^ dbg}: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us', flush=True)
