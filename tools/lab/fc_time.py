"""fp16-storage fully connected layers at batch 64 (VGG-16 classifier: 25088 -> 4096 -> 4096 -> 1000): implicit-GEMM tiles, us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for k, n in [(25088, 4096), (4096, 4096), (4096, 1000)]:
    w = torch.randn(n, k) / k ** 0.5
    fwd = cp.linear_fwd_plan(w, torch.randn(n), DEV, 'fc')
    bwd = cp.linear_dgrad_plan(w, DEV, 'fc_dgrad')
    x = torch.randn(64, 1, 1, k, device=DEV).half()
    y = torch.zeros(64, 1, 1, n, device=DEV, dtype=torch.float16)
    gy = torch.randn(64, 1, 1, (n + 31) // 32 * 32, device=DEV).half()[..., :bwd.cin_p].contiguous() if bwd.cin_p != n else torch.randn(64, 1, 1, n, device=DEV).half()
    gx = torch.zeros(64, 1, 1, k, device=DEV, dtype=torch.float16)
    for name, plan, a, b in (('fwd', fwd, x, y), ('dgrad', bwd, gy, gx)):
        res = {}
        for tile in (0, 60, 61, 62, 63):
            cp.FORCE_TILE = tile
            try:
                plan.run(a, b)
                res[f'{tile}->{plan.last_tile}' if tile == 0 else tile] = t(lambda: plan.run(a, b))
            except Exception as e:
                res[tile] = str(e)[:30]
        cp.FORCE_TILE = 0
        print(f'{k}->{n} {name}: ' + '  '.join(f'{kk}: {v:.0f} us' if isinstance(v, float) else f'{kk}: {v}' for kk, v in res.items()), flush=True)
