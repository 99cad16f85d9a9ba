import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import synthetic as syn, models as M, _lib
_lib.load()
DEV = 'cuda'
cam = prj = (256, 256)
sd = syn.pcnet_state_dict(0, cam_sz=cam, mask='ones')
pc = M.PCNet(sd['mask'], M.WarpingNet(out_size=cam)); pc.load_state_dict(sd); pc = pc.to(DEV)
for cap in (0, 400, 484, 576, 784, 1024):
    M.TILED_BOX_CAP = cap
    eng = M.PCNetEngine(pc, 64, prj)
    if cap == 0:
        eng.tiled = None
    x = torch.rand(64, 256, 256, 4, device=DEV); eng._x, eng._clamp = x, 1
    g = torch.randn(64, 256, 256, 4, device=DEV)
    for _ in range(3): eng.warp_backward(g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): eng.warp_backward(g)
    e1.record(); torch.cuda.synchronize()
    nd = int((eng.tiled[2][:, 2] < 0).sum()) if eng.tiled is not None else -1
    print(f'cap {cap}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; direct tiles {nd} of 256; box_cap {eng.tiled[3] if eng.tiled else None}', flush=True)
    del eng
