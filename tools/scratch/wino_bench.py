"""Winograd vs direct bf16x6 on the six big ShadingNet layers (B=64, 64x64) -- timing with HIP events."""
import sys, torch
sys.path.insert(0, '.')
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
torch.manual_seed(0)
for ci, co in [(128, 256), (256, 128), (64, 128), (128, 128), (256, 256)]:
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** .5
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV, name=f'c{ci}_{co}')
    x = torch.relu(torch.randn(64, 64, 64, ci, device=DEV))
    out = torch.zeros(64, 64, 64, co, device=DEV)
    res = {}
    for t in (34, 70):
        cp.FORCE_TILE = t
        for _ in range(3):
            plan.run(x, out, act=_lib.ACT_RELU)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.run(x, out, act=_lib.ACT_RELU)
        e1.record()
        torch.cuda.synchronize()
        res[t] = e0.elapsed_time(e1) / 20 * 1e3
        if t == 34:
            ref = out.clone()
    cp.FORCE_TILE = 0
    fl = 2 * 9 * ci * co * 64 * 64 * 64
    print(f'{ci}->{co}: direct {res[34]:.0f} us ({fl / res[34] / 1e6:.0f} TF)  winograd {res[70]:.0f} us ({fl / res[70] / 1e6:.0f} TF-equiv)  '
          f'maxdiff {(out - ref).abs().max().item():.2e} / {ref.abs().max().item():.2e}', flush=True)
