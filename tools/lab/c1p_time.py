"""Times the fused conv1_s + conv1 launch (csrc/conv1pair.hip) at batch 64, 256 x 256, beside the warp kernel with and without
the 8-channel concatenation and the two separate smallcin launches: python tools/lab/c1p_time.py [f16]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import _lib, models as M, synthetic as syn
_lib.load()
DEV = 'cuda'
B, sz = 64, (256, 256)
storage = 'f16' if len(sys.argv) > 1 and sys.argv[1] == 'f16' else 'f32'
sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='ones')
pc = M.PCNet(sd['mask'], M.WarpingNet(out_size=sz))
pc.load_state_dict(sd)
pc = pc.to(DEV)
eng = M.PCNetEngine(pc, B, sz, storage)
eng.set_scene(M.to_nhwc4(syn.scenes(1, 1, sz).repeat(B, 1, 1, 1).to(DEV)))
x = M.to_nhwc4(torch.rand(B, 3, *sz).to(DEV))
a, f, m = eng.a, eng.f, eng.m
wp, b1, bs = eng.pair1


def timeit(fn, n=20):
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


def pair():
    _lib.call('spaa_conv1_pair_fwd', _lib.ptr(a['xw']), _lib.ptr(eng.scene), _lib.ptr(wp), _lib.ptr(b1), _lib.ptr(bs), _lib.ptr(a['S1']),
              _lib.ptr(a['X1']), M.C_ptr(m['S1']), M.C_ptr(m['X1']), B, sz[0], sz[1], int(storage == 'f16'))


def sep():
    f['conv1_s'].run(a['cat8'], a['S1'], act=_lib.ACT_RELU, mask_out=m['S1'])
    f['conv1'].run(a['xw'], a['X1'], add=a['S1'], act=_lib.ACT_RELU, mask_out=m['X1'])


keep = eng.pair1
t_w1 = timeit(lambda: eng.warp(x))
eng.pair1 = None
t_w0 = timeit(lambda: eng.warp(x))
eng.pair1 = keep
mb = (B * sz[0] * sz[1] * 32 + B * (sz[0] // 2) * (sz[1] // 2) * (2 * 32 * (2 if storage == 'f16' else 4) + 16)) / 1e6
t_p, t_s = timeit(pair), timeit(sep)
print(f'{storage}: warp with cat8 {t_w0:.1f} us, without {t_w1:.1f} us; conv1_s + conv1 separate {t_s:.1f} us, fused {t_p:.1f} us '
      f'({mb:.0f} MB: {mb / t_p / 1e3:.2f} TB/s)')
