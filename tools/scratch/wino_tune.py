"""Winograd (tile 70) vs the tuned direct kernel for every 3x3/s1 layer shape in tapconv_tune.json with Cout >= 96."""
import sys, json, math, torch
sys.path.insert(0, '.')
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
t = json.load(open('spaa_amd/tapconv_tune.json'))
res = {}
for k, v in sorted(t.items()):
    p = k.split('_')
    if not (p[2] == '9' and p[3] == '1' and p[4] == '1' and 'fold' not in k):
        continue
    ci, co, m = int(p[0]), int(p[1]), int(p[5])
    if ci % 32 or co < 64:
        continue
    hw = int(round(math.sqrt(m / 64)))
    if 64 * hw * hw != m:
        continue
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** .5
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV)
    x = torch.relu(torch.randn(64, hw, hw, ci, device=DEV))
    out = torch.zeros(64, hw, hw, co, device=DEV)
    r = {}
    for tile in (v, 70):
        saved = cp.TUNE.get(k)
        cp.TUNE[k] = tile
        for _ in range(2):
            plan.run(x, out, act=_lib.ACT_RELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            plan.run(x, out, act=_lib.ACT_RELU)
        e1.record()
        torch.cuda.synchronize()
        r[tile] = e0.elapsed_time(e1) / 10 * 1e3
        cp.TUNE[k] = saved
    print(f'{k}: tuned tile {v} {r[v]:.0f} us   winograd {r[70]:.0f} us   {"WINO" if r[70] < 0.97 * r[v] else "keep"}', flush=True)
    if r[70] < 0.97 * r[v]:
        res[k] = 70
    del plan, x, out
print(json.dumps(res))
