"""Phase stagger of the 8-wave Winograd kernel (csrc/tapconv_wino.hip, SPAA_WINO_STAGGER_US): the six 64 x 64 x (128 <-> 256) layers of
ShadingNetSPAA at batch 64 with every second first-round workgroup started N us late; us per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
torch.manual_seed(0)
B = 64


def t(fn, n=20):
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


for name, ci, co in (('conv4 (128->256, +residual, ReLU, mask)', 128, 256), ('conv5 shape (256->128, +residual, ReLU, mask)', 256, 128)):
    wt = torch.randn(co, ci, 3, 3) / (ci * 9) ** 0.5
    plan = cp.conv_fwd_plan(wt, torch.randn(co), 1, 1, DEV)
    x = torch.relu(torch.randn(B, 64, 64, ci, device=DEV))
    out = torch.zeros(B, 64, 64, co, device=DEV)
    add = torch.randn(B, 64, 64, co, device=DEV)
    mask = torch.zeros(B, 64, 64, co // 4, dtype=torch.uint8, device=DEV)
    gate = (torch.rand(B, 64, 64, co // 4, device=DEV) * 16).to(torch.uint8)
    cp.FORCE_TILE = 70
    line = []
    for us in (0, 10, 20, 30, 40, 50, 60, 80, 0):
        os.environ['SPAA_WINO_STAGGER_US'] = str(us)
        a = t(lambda: plan.run(x, out, add=add, act=_lib.ACT_RELU, mask_out=mask))
        b = t(lambda: plan.run(x, out))
        c = t(lambda: plan.run(x, out, gate_bits=gate))
        line.append(f'{us}: {a:.0f} / {b:.0f} / {c:.0f}')
    print(f'{name}, stagger us: residual+ReLU+mask / plain / gate bits:  ' + '   '.join(line), flush=True)
os.environ['SPAA_WINO_STAGGER_US'] = '0'
