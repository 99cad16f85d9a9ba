"""Thin-output input gradients at full size: folded matrix-core kernel (tile 72) against the VALU kernels (28 / 29 / 47), fp32 and fp16 input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
torch.manual_seed(0)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, ci, co, k, s, pad, h, w, b in [('resnet stem', 3, 64, 7, 2, 3, 224, 224, 64), ('pcnet conv1', 3, 32, 3, 2, 1, 256, 256, 64),
                                         ('vgg conv1_1', 3, 64, 3, 1, 1, 224, 224, 64), ('inception 1a', 3, 32, 3, 2, 0, 299, 299, 64)]:
    wt = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    dplan = cp.conv_dgrad_plan(wt, s, pad, DEV)
    gx = torch.zeros(b, h, w, 4, device=DEV)
    for half in (False, True):
        gy = torch.randn(b, ho, wo, co, device=DEV)
        gy = gy.half() if half else gy
        res = {}
        for tile in (72, 28, 29, 47):
            cp.FORCE_TILE = tile
            try:
                dplan.run(gy, gx)
                if dplan.last_tile == tile:
                    res[tile] = t(lambda: dplan.run(gy, gx))
            except Exception as e:
                res[tile] = str(e)[:20]
        cp.FORCE_TILE = 0
        print(f'{name} {"f16" if half else "f32"}: ' + '  '.join(f'{k_}: {v:.0f} us' if isinstance(v, float) else f'{k_}: {v}' for k_, v in res.items()), flush=True)
