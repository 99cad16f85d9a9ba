"""thinmf kernel, fp32 stem shape: with / without the operand split (make THINMF_ABLATE=1; timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spaa_amd import convplan as cp
DEV = torch.device('cuda:0')
wt = torch.randn(64, 3, 7, 7) / 12
dplan = cp.conv_dgrad_plan(wt, 2, 3, DEV)
gy = torch.randn(64, 112, 112, 64, device=DEV)
gx = torch.zeros(64, 224, 224, 4, device=DEV)
cp.FORCE_TILE = 72
for dbg in (0, 1, 0, 1):
    cp.DEBUG_THINMF = dbg
    for _ in range(3): dplan.run(gy, gx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dplan.run(gy, gx)
    e1.record(); torch.cuda.synchronize()
    print('no split' if dbg else 'full    ', round(e0.elapsed_time(e1) / 20 * 1e3), 'us', flush=True)
