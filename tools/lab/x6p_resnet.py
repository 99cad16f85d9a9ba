"""ResNet-18 layer2.0.conv1 input gradient (128 -> 64 channels, k3 s2: 28^2 -> 56^2, batch 64) on the patch-staged stride-2 kernel
(tile 74, unfolded classes) against the tuned choice: python tools/lab/x6p_resnet.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
B = 64
torch.manual_seed(0)


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, co, ci, hin in (('layer2.0.conv1_dgrad', 128, 64, 28), ('layer3.0.conv1_dgrad', 256, 128, 14), ('layer4.0.conv1_dgrad', 512, 256, 7)):
    w = torch.randn(co, ci, 3, 3) / (3 * ci ** 0.5)
    g = torch.randn(B, hin, hin, co, device=DEV)
    hout = 2 * hin
    ref_plan = cp.conv_dgrad_plan(w, 2, 1, DEV, name)
    out0 = torch.zeros(B, hout, hout, ci, device=DEV)
    t0 = timeit(lambda: ref_plan.run(g, out0))
    line = f'{name}: tuned choice (tile {ref_plan.last_tile}) {t0:.1f} us'
    p = cp.conv_dgrad_plan(w, 2, 1, DEV, name + '/x6p', fold=False)
    if p.x6p_ok() and ci in (32, 64):
        p.fixed_tile = 74
        out1 = torch.zeros_like(out0)
        t1 = timeit(lambda: p.run(g, out1))
        err = float((out1 - out0).abs().max() / out0.abs().max())
        line += f'; tile 74 {t1:.1f} us (max diff {err:.1e})'
    print(line)
