import sys, os, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from spaa_amd import synthetic as syn
from spaa_amd import models as m_, _lib
_lib.load()
DEV = 'cuda'
cam = (256, 256); b = 64
sd = syn.pcnet_state_dict(4, cam_sz=cam, mask='ones')
pc = m_.PCNet(sd['mask'], m_.WarpingNet(out_size=cam)); pc.load_state_dict(sd); pc = pc.to(DEV)
eng = m_.PCNetEngine(pc, b, cam)
t = eng.tail; a = eng.a; m = eng.m; g = eng.g
gP = torch.randn(b, 256, 256, 4, device=DEV)
a['X6'].normal_()
def run(name, fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, f'{e0.elapsed_time(e1) / 10 * 1e3:.0f} us', flush=True)
fw = lambda: _lib.call('spaa_shading_tail_fwd', _lib.ptr(a['X6']), _lib.ptr(t['w2s']), _lib.ptr(t['b2']), _lib.ptr(t['w6']), _lib.ptr(t['b6']), _lib.ptr(a['R1']), _lib.ptr(a['Y']), _lib.ptr(a['Ypre']), _lib.ptr(m['X7']), b, 128, 128)
bw = lambda: _lib.call('spaa_shading_head_bwd', _lib.ptr(gP), _lib.ptr(t['w6t']), _lib.ptr(t['w2ts']), _lib.ptr(m['X7']), _lib.ptr(m['X6']), _lib.ptr(g['P6']), b, 128, 128)
run('fwd dbg=' + os.environ.get('SPAA_TAIL_DBG', '0'), fw)
run('bwd dbg=' + os.environ.get('SPAA_TAIL_DBG', '0'), bw)
