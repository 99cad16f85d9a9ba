"""Timing-only ablations of the Winograd kernel (DBG template bits, wrong results): where does a conv4-shaped launch spend its time?"""
import os, sys, torch
sys.path.insert(0, '.')
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
NAMES = {0: 'baseline', 1: 'no epilogue', 2: 'no fold', 4: 'no V', 8: 'no barrier', 3: 'no epi+fold', 5: 'no epi+V', 7: 'no epi+fold+V',
         15: 'no epi+fold+V+barrier', 31: 'no epi+fold+V+barrier+DMA', 32: 'raw barrier', 33: 'raw barrier, no epilogue'}
for ci, co, res in [(128, 256, False), (128, 256, True), (256, 128, False)]:
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** .5
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV)
    x = torch.relu(torch.randn(64, 64, 64, ci, device=DEV))
    out = torch.zeros(64, 64, 64, co, device=DEV)
    add = torch.randn(64, 64, 64, co, device=DEV) if res else None
    mask = torch.zeros(64, 64, 64, co // 4, device=DEV, dtype=torch.uint8) if res else None
    cp.FORCE_TILE = 70
    order = [int(v) for v in os.environ.get('WINO_DBG', '0').split(',')]
    res_t = {k: [] for k in order}
    for rnd in range(3):
        for dbg in order:
            cp.DEBUG_WINO = dbg << 2
            for _ in range(2):
                plan.run(x, out, add=add, act=_lib.ACT_RELU, mask_out=mask)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                plan.run(x, out, add=add, act=_lib.ACT_RELU, mask_out=mask)
            e1.record()
            torch.cuda.synchronize()
            res_t[dbg].append(e0.elapsed_time(e1) / 10 * 1e3)
    for dbg in order:
        print(f'{ci}->{co} res={res} {NAMES[dbg]:28s}: ' + ' '.join(f'{t:6.0f}' for t in res_t[dbg]) + ' us', flush=True)
