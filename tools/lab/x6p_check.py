"""Full-size check of tile 74 (BN = 32 form) against the implicit-GEMM tile on the same data, per image."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp, _lib
DEV = torch.device('cuda:0')
torch.manual_seed(0)
B = 64
w2c = torch.randn(64, 32, 3, 3) / 17
plan = cp.conv_dgrad_plan(w2c, 2, 1, DEV, 'conv2_dgrad', fold=False)
for trial in range(3):
    scale = [1.0, 1e-6, 1.0][trial]
    p2 = torch.randn(B, 64, 64, 64, device=DEV) * scale
    t1 = torch.randn(B, 128, 128, 32, device=DEV) * scale
    m1 = (torch.rand(B, 128, 128, 8, device=DEV) * 16).to(torch.uint8)
    outs = {}
    for tile in (30, 74, 74):
        cp.FORCE_TILE = tile
        o = torch.zeros(B, 128, 128, 32, device=DEV)
        plan.run(p2, o, add=t1, gate_bits=m1)
        torch.cuda.synchronize()
        outs.setdefault(tile, []).append(o)
    cp.FORCE_TILE = 0
    a, b, c = outs[30][0], outs[74][0], outs[74][1]
    d = (a - b).abs().flatten(1).max(dim=1).values / a.abs().max()
    print(f'trial {trial} scale {scale}: bitwise run-to-run {torch.equal(b, c)}; per-image rel diff vs tile 30: max {d.max().item():.2e}; images above 1e-5: {[i for i, v in enumerate(d.tolist()) if v > 1e-5]}')
    if d.max() > 1e-5:
        i = int(d.argmax())
        e = (a[i] - b[i]).abs()
        pos = (e > 1e-5 * a.abs().max()).nonzero()
        print('   first bad positions (y, x, c):', pos[:12].tolist(), ' count', pos.shape[0], ' rows', sorted(set(pos[:, 0].tolist()))[:20], 'cols', sorted(set(pos[:, 1].tolist()))[:40])
