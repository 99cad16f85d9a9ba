"""First layers of the classifier bodies at batch 64: the 3-channel-image kernel (tile 76, csrc/tapconv_c3.hip) against the tile the
tune table / rules pick today, fp32 and fp16 output, with bias + ReLU (+ gate bytes where the body writes them)."""
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
for name, co, k, s, pad, h, w, b, masked in [('resnet stem', 64, 7, 2, 3, 224, 224, 64, False), ('vgg conv1_1', 64, 3, 1, 1, 224, 224, 64, True),
                                             ('inception 1a', 32, 3, 2, 0, 299, 299, 64, True)]:
    wt = torch.randn(co, 3, k, k) / (3 * k * k) ** 0.5
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    plan = cp.conv_fwd_plan(wt, torch.randn(co), s, pad, DEV, name)
    x = torch.rand(b, h, w, 4, device=DEV)
    x[..., 3] = 0
    for dt in (torch.float32, torch.float16):
        out = torch.zeros(b, ho, wo, co, device=DEV, dtype=dt)
        mask = torch.zeros(b, ho, wo, co // 4, device=DEV, dtype=torch.uint8) if masked else None
        res = {}
        for tile in (0, 76):
            cp.FORCE_TILE = tile
            plan.run(x, out, act=_lib.ACT_RELU, mask_out=mask)
            res[cp.TILE_NAMES.get(plan.last_tile, plan.last_tile)] = t(lambda: plan.run(x, out, act=_lib.ACT_RELU, mask_out=mask))
        cp.FORCE_TILE = 0
        gb = (out.numel() * out.element_size() + x.numel() * 4) / 1e9
        print(f'{name} -> {"f16" if dt == torch.float16 else "f32"} ({gb:.2f} GB of tensors): ' + '  '.join(f'{k_}: {v:.0f} us' for k_, v in res.items()), flush=True)
