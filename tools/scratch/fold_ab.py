"""k3/s2 fractional-stride layers: four parity classes as separate passes vs folded into the GEMM rows (zero weights where a
class has no tap): timing of both plans in one process."""
import sys, torch
sys.path.insert(0, '.')
from spaa_amd import convplan as cp, _lib
_lib.load()
DEV = 'cuda'
torch.manual_seed(0)
def t(plan, x, out, **kw):
    best = 1e9
    for rep in range(3):
        for _ in range(2):
            plan.run(x, out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            plan.run(x, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    return best
for name, ci, co, hw in [('conv2_dgrad', 32, 64, 128), ('layer2.0.conv1_dgrad', 64, 128, 56), ('layer3.0.conv1_dgrad', 128, 256, 28), ('layer4.0.conv1_dgrad', 256, 512, 14)]:
    # forward conv ci -> co, k3 s2 p1 at hw x hw; its input gradient consumes [.., co] at hw/2 and produces [.., ci] at hw
    w = torch.randn(co, ci, 3, 3) / (ci * 9) ** .5
    g = torch.randn(64, hw // 2, hw // 2, co, device=DEV)
    out1, out2 = torch.zeros(64, hw, hw, ci, device=DEV), torch.zeros(64, hw, hw, ci, device=DEV)
    add = torch.randn(64, hw, hw, ci, device=DEV)
    p_sep = cp.conv_dgrad_plan(w, 2, 1, DEV, fold=False)
    p_fold = cp.conv_dgrad_plan(w, 2, 1, DEV, fold=True)
    a, b = t(p_sep, g, out1, add=add), t(p_fold, g, out2, add=add)
    print(f'{name}: separate {a:.0f} us  folded {b:.0f} us   maxdiff {(out1 - out2).abs().max().item():.2e} / {out1.abs().max().item():.2e}', flush=True)
w = torch.randn(128, 64, 3, 3) / 34
x = torch.relu(torch.randn(64, 64, 64, 128, device=DEV)); out1 = torch.zeros(64, 128, 128, 64, device=DEV); out2 = torch.zeros_like(out1)
add = torch.randn(64, 128, 128, 64, device=DEV)
p_sep = cp.deconv_fwd_plan(w, torch.randn(64), 2, 1, DEV, fold=False)
p_fold = cp.deconv_fwd_plan(w, p_sep.bias, 2, 1, DEV, fold=True)
a, b = t(p_sep, x, out1, add=add, act=_lib.ACT_RELU), t(p_fold, x, out2, add=add, act=_lib.ACT_RELU)
print(f'transConv1: separate {a:.0f} us  folded {b:.0f} us   maxdiff {(out1 - out2).abs().max().item():.2e}', flush=True)
w = torch.randn(64, 32, 3, 3) / 17
g = torch.randn(64, 64, 64, 64, device=DEV)
out2 = torch.zeros(64, 128, 128, 32, device=DEV); add = torch.randn(64, 128, 128, 32, device=DEV)
p_fold = cp.conv_dgrad_plan(w, 2, 1, DEV, fold=True)
for tile in (25, 26, 27, 33, 34, 35, 36, 39, 40, 42, 44, 48, 49, 50, 52, 54):
    cp.FORCE_TILE = tile
    print('conv2_dgrad folded tile', tile, f'{t(p_fold, g, out2, add=add):.0f} us', flush=True)
cp.FORCE_TILE = 0
