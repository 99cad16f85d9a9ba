"""CPU: the C-ABI library loads and exports every symbol include/spaa_hip.h declares; host logic fails loudly
without a GPU (no CPU fallback anywhere in the product path)."""
import ctypes
import os
import re

import pytest
import torch

from spaa_amd import _lib, synthetic as syn
from spaa_amd.sharding import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'spaa_hip.h')).read()
    return sorted(set(re.findall(r'\b(spaa_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/spaa_hip.h but not exported by libspaa_hip.so'
    assert set(names) == set(_lib.EXPORTS)
    assert b'gfx950' in lib.spaa_version()


def test_argument_counts_match_header():
    hdr = open(os.path.join(ROOT, 'include', 'spaa_hip.h')).read()
    for name, argtypes in _lib._SIGNATURES.items():
        m = re.search(r'int\s+' + name + r'\s*\(([^;]*?)\)\s*;', hdr, re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(',') if a.strip()])
        assert nargs == len(argtypes), (name, nargs, len(argtypes))


def test_struct_layout_matches_c():
    # spaa_tapclass_t: 6 x int32 + int64 = 32 bytes
    assert ctypes.sizeof(_lib.TapClass) == 32
    assert _lib.TapConv.cls.offset % 8 == 0
    assert _lib.TapConv.in2.offset == _lib.TapConv.cls.offset + 4 * 32      # (the second-source fields follow the classes)
    assert ctypes.sizeof(_lib.TapConv) == _lib.TapConv.in2.offset + 32
    # the compiled struct itself (layout probes exported by the library)
    lib = _lib.load()
    assert lib.spaa_tapconv_sizeof() == ctypes.sizeof(_lib.TapConv)
    for i, f in enumerate(('out', 'weights', 'taps', 'gate2', 'mask_out', 'tap_range', 'splitk_ws', 'io_dtype', 'nclass', 'cls', 'in2', 'w2_split')):
        assert lib.spaa_tapconv_offsetof(i) == getattr(_lib.TapConv, f).offset, f


def test_no_cpu_fallback():
    from spaa_amd.models import PCNet, WarpingNet, to_nhwc4
    from spaa_amd.classifier import Classifier
    from spaa_amd.projector_based_attack import spaa
    sd = syn.pcnet_state_dict(0, cam_sz=(64, 64))
    pc = PCNet(sd['mask'], WarpingNet(out_size=(64, 64)))
    pc.load_state_dict(sd)
    assert len(pc.state_dict()) == 46 and sum(p.numel() for p in pc.parameters()) == 1259435
    with pytest.raises(RuntimeError):
        to_nhwc4(torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError):
        Classifier('resnet18', 'cpu')  # no weights, no download
    clf = Classifier('resnet18', 'cpu', state_dict=syn.resnet18_state_dict(2))
    setup = dict(classifier_crop_sz=(60, 60), prj_brightness=0.5, prj_im_sz=(64, 64))
    with pytest.raises(RuntimeError):
        spaa(pc, clf, None, [1], True, syn.scenes(1, 1, (64, 64)), 5, 'caml2', 'cpu', setup)
    with pytest.raises(TypeError):   # neither a spaa_amd.Classifier nor a callable
        spaa(pc, object(), None, [1], True, syn.scenes(1, 1, (64, 64)), 5, 'caml2', 'cuda', setup)
    with pytest.raises(RuntimeError):  # a foreign callable takes the autograd route, which is GPU-only as well
        spaa(pc, lambda im, cp: None, None, [1], True, syn.scenes(1, 1, (64, 64)), 5, 'caml2', 'cpu', setup)
    with pytest.raises(TypeError):   # PCNet must be ours
        spaa(torch.nn.Identity(), clf, None, [1], True, syn.scenes(1, 1, (64, 64)), 5, 'caml2', 'cuda', setup)


def test_shard_range_partitions():
    for n in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_torch_library_ops_are_registered():
    """The PyTorch-ROCm custom ops of namespace `spaa` (spaa_amd/ops.py): registered with schemas, CUDA(HIP)-only — a CPU
    tensor is a dispatcher error, not a silent fallback — and with fake (meta) implementations for shape inference."""
    from spaa_amd import ops
    for n in ops.OPS:
        assert hasattr(torch.ops.spaa, n), n
    assert str(torch.ops.spaa.rgb2lab.default._schema) == 'spaa::rgb2lab(Tensor rgb4) -> Tensor'
    assert 'Tensor lab1, Tensor lab2' in str(torch.ops.spaa.ciede2000.default._schema)
    assert 'Int handle' in str(torch.ops.spaa.pcnet_forward.default._schema)
    with pytest.raises(NotImplementedError):
        torch.ops.spaa.rgb2lab(torch.zeros(1, 2, 2, 4))
    with pytest.raises(NotImplementedError):
        torch.ops.spaa.ciede2000(torch.zeros(1, 2, 2, 4), torch.zeros(1, 2, 2, 4))
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(2, 3, 8, 10, device='cuda')
        x4 = torch.ops.spaa.nchw_to_nhwc4(x)
        assert x4.shape == (2, 8, 10, 4)
        assert torch.ops.spaa.ciede2000(x4, x4).shape == (2, 8, 10) and torch.ops.spaa.rgb2lab(x4).shape == x4.shape
        assert torch.ops.spaa.nhwc4_to_nchw(x4).shape == (2, 3, 8, 10)


def _wino_plan(b, h, w, cin, cout, tile=70, ksplit=0, force_canvas=False):
    d = _lib.TapConv()
    d.reserved0 = (1 << 29) if force_canvas else 0
    d.Hin = d.Hout = d.Hm = h
    d.Win = d.Wout = d.Wm = w
    d.Cin, d.Cout, d.B, d.s_in, d.s_out, d.nclass = cin, cout, b, 1, 1, 1
    d.in_cstride, d.out_cstride = cin, cout
    d.cls[0].ntaps, d.cls[0].K, d.cls[0].Kpad = 16, 16 * cin, 16 * cin + 128
    d.tile, d.ksplit = tile, ksplit
    wp = (ctypes.c_int32 * 8)()
    assert _lib.load().spaa_tapconv_wino_plan(ctypes.byref(d), wp) == 0
    return dict(zip(('bn', 'ksplit', 'canvas', 'gy', 'gx', 'nwg', 'kb_per', 'ncanvas'), wp))


@pytest.mark.parametrize('tile', [70, 73])
@pytest.mark.parametrize('b,h,w,cin,cout', [(64, 14, 14, 256, 256), (64, 7, 7, 512, 512), (64, 35, 35, 96, 96), (5, 14, 14, 64, 64),
                                             (64, 17, 17, 128, 128), (3, 7, 9, 512, 512), (64, 14, 14, 512, 512)])
def test_winograd_canvas_plan_covers_every_pixel_once(b, h, w, cin, cout, tile):
    """csrc/tapconv_wino.hip, canvas form (small images laid out on virtual canvases, host-side plan through
    spaa_tapconv_wino_plan): the kernel's index arithmetic restated here -- division by the period as a multiplication by
    ceil(2^20 / period) -- maps the workgroup regions onto every output pixel exactly once, and the 3 x 3 neighbourhood of every
    output pixel onto the same image's pixels or onto the zero padding (never onto a neighbouring image)."""
    import numpy as np
    pl = _wino_plan(b, h, w, cin, cout, tile=tile, force_canvas=True)
    rh = 8 if tile == 73 else 16          # region height: four-wave workgroups own 4 x 16 tiles
    if tile == 73 and pl['canvas'] == 0:
        pytest.skip('8-row regions: the image-aligned form already has no fewer regions')
    assert pl['canvas'] == 1 and pl['ksplit'] * pl['kb_per'] >= cin // 32 > (pl['ksplit'] - 1) * pl['kb_per']
    gy, gx, nc = pl['gy'], pl['gx'], pl['ncanvas']
    py, px = h + 1, w + 1
    assert gy * gx * nc >= b and gy * py - 1 <= 4095 and gx * px - 1 <= 4095 and py <= 255 and px <= 255
    my, mx = ((1 << 20) + py - 1) // py, ((1 << 20) + px - 1) // px
    wg_y, wg_x = (gy * py - 1 + rh - 1) // rh, (gx * px - 1 + 31) // 32
    n_tiles = (cout + pl['bn'] - 1) // pl['bn']
    assert pl['nwg'] == nc * wg_y * wg_x * n_tiles * pl['ksplit']

    def canvas_pixel(cv, vy, vx):
        """-> (valid, image, iy, ix): the device lambda `canvas_pixel`, on int arrays (uint32 wrap-around for negative v)"""
        sy = ((vy.astype(np.int64) & 0xffffffff) * my & 0xffffffff) >> 20
        sx = ((vx.astype(np.int64) & 0xffffffff) * mx & 0xffffffff) >> 20
        iy, ix = vy - sy * py, vx - sx * px
        im = cv * gy * gx + sy * gx + sx
        ok = (vy >= 0) & (vx >= 0) & (iy < h) & (ix < w) & (sy < gy) & (sx < gx) & (im < b)
        return ok, im, iy, ix

    count = np.zeros((b, h, w), dtype=np.int64)
    for cv in range(nc):
        vy, vx = np.meshgrid(np.arange(rh * wg_y), np.arange(32 * wg_x), indexing='ij')
        ok, im, iy, ix = canvas_pixel(cv, vy, vx)
        assert (iy[ok] >= 0).all() and (ix[ok] >= 0).all()
        np.add.at(count, (im[ok], iy[ok], ix[ok]), 1)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok2, im2, iy2, ix2 = canvas_pixel(cv, vy + dy, vx + dx)
                inside = ok & (iy + dy >= 0) & (iy + dy < h) & (ix + dx >= 0) & (ix + dx < w)
                assert (ok2[ok] == inside[ok]).all()                      # a tap is read iff it lies inside the SAME image
                sel = ok & inside
                assert (im2[sel] == im[sel]).all() and (iy2[sel] == iy[sel] + dy).all() and (ix2[sel] == ix[sel] + dx).all()
    assert (count == 1).all()


def _h16p_plan(b, h, w, cin, cout, ksplit=0, bn=0, force_canvas=False):
    d = _lib.TapConv()
    d.Hin = d.Hout = d.Hm = h
    d.Win = d.Wout = d.Wm = w
    d.Cin, d.Cout, d.B, d.s_in, d.s_out, d.nclass = cin, cout, b, 1, 1, 1
    d.in_cstride, d.out_cstride = cin, cout
    d.cls[0].ntaps, d.cls[0].K, d.cls[0].Kpad = 9, 9 * cin, 9 * cin
    d.tile, d.ksplit, d.reserved1 = 68, ksplit, 4 | {0: 0, 64: 1, 128: 2}[bn] | (8 if force_canvas else 0)
    wp = (ctypes.c_int32 * 8)()
    rc = _lib.load().spaa_tapconv_h16p_plan(ctypes.byref(d), wp)
    return rc, dict(zip(('bn', 'ksplit', 'canvas', 'gy', 'gx', 'nwg', 'kb_per', 'ncanvas'), wp))


@pytest.mark.parametrize('b,h,w,cin,cout', [(64, 14, 14, 256, 256), (64, 7, 7, 512, 512), (5, 14, 14, 64, 64), (64, 17, 17, 128, 128),
                                             (3, 7, 9, 512, 512), (64, 14, 14, 512, 512), (9, 20, 33, 96, 192)])
def test_h16p_canvas_plan_covers_every_pixel_once(b, h, w, cin, cout):
    """csrc/tapconv_h16p.hip, canvas / K-range form of the patch-staged fp16 kernel (host-side plan through spaa_tapconv_h16p_plan): the
    kernel's index arithmetic restated here maps the 16 x 32-pixel regions onto every output pixel exactly once, the 3 x 3
    neighbourhood of every output pixel onto the same image's pixels or onto the zero padding, and the K ranges onto every
    32-channel block exactly once."""
    import numpy as np
    rc, pl = _h16p_plan(b, h, w, cin, cout, force_canvas=True)
    assert rc == 0 and pl['bn'] in (64, 128) and pl['ksplit'] >= 1
    nkb = cin // 32
    assert pl['ksplit'] * pl['kb_per'] >= nkb > (pl['ksplit'] - 1) * pl['kb_per']
    gy, gx, nc = pl['gy'], pl['gx'], pl['ncanvas']
    n_tiles = (cout + pl['bn'] - 1) // pl['bn']
    if not pl['canvas']:
        assert pl['nwg'] == b * ((h + 15) // 16) * ((w + 31) // 32) * n_tiles * pl['ksplit']
        return
    py, px = h + 1, w + 1
    assert gy * gx * nc >= b and gy * py - 1 <= 4095 and gx * px - 1 <= 4095 and py <= 255 and px <= 255
    my, mx = ((1 << 20) + py - 1) // py, ((1 << 20) + px - 1) // px
    wg_y, wg_x = (gy * py - 1 + 15) // 16, (gx * px - 1 + 31) // 32
    assert pl['nwg'] == nc * wg_y * wg_x * n_tiles * pl['ksplit']
    assert pl['nwg'] // (n_tiles * pl['ksplit']) < b * ((h + 15) // 16) * ((w + 31) // 32)     # fewer regions than the image-aligned form

    def canvas_pixel(cv, vy, vx):
        sy = ((vy.astype(np.int64) & 0xffffffff) * my & 0xffffffff) >> 20
        sx = ((vx.astype(np.int64) & 0xffffffff) * mx & 0xffffffff) >> 20
        iy, ix = vy - sy * py, vx - sx * px
        im = cv * gy * gx + sy * gx + sx
        ok = (vy >= 0) & (vx >= 0) & (iy < h) & (ix < w) & (sy < gy) & (sx < gx) & (im < b)
        return ok, im, iy, ix

    count = np.zeros((b, h, w), dtype=np.int64)
    for cv in range(nc):
        vy, vx = np.meshgrid(np.arange(16 * wg_y), np.arange(32 * wg_x), indexing='ij')
        ok, im, iy, ix = canvas_pixel(cv, vy, vx)
        assert (iy[ok] >= 0).all() and (ix[ok] >= 0).all()
        np.add.at(count, (im[ok], iy[ok], ix[ok]), 1)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok2, im2, iy2, ix2 = canvas_pixel(cv, vy + dy, vx + dx)
                inside = ok & (iy + dy >= 0) & (iy + dy < h) & (ix + dx >= 0) & (ix + dx < w)
                assert (ok2[ok] == inside[ok]).all()                      # a tap is read iff it lies inside the SAME image
                sel = ok & inside
                assert (im2[sel] == im[sel]).all() and (iy2[sel] == iy[sel] + dy).all() and (ix2[sel] == ix[sel] + dx).all()
    assert (count == 1).all()


def test_h16p_plan_honours_forced_choices():
    """Explicit K ranges / N tiles of the patch-staged fp16 kernel's plan are honoured and clamped; a K split needs whole channel quads."""
    rc, p = _h16p_plan(64, 14, 14, 256, 256, ksplit=3, bn=64)
    assert rc == 0 and p['bn'] == 64 and p['ksplit'] == 3 and p['kb_per'] == 3
    assert _h16p_plan(64, 14, 14, 256, 256, ksplit=1)[1]['ksplit'] == 1
    assert _h16p_plan(64, 14, 14, 64, 64, ksplit=16)[1]['ksplit'] == 2
    assert _h16p_plan(64, 7, 7, 512, 512, bn=128)[1]['bn'] == 128
    assert _h16p_plan(8, 14, 14, 256, 126, ksplit=4)[0] != 0
    rc, p = _h16p_plan(8, 14, 14, 256, 126)
    assert rc == 0 and p['ksplit'] == 1
    # the shapes this form was built for fill the chip: ResNet-18 layer3 / layer4 and VGG-16's last block at batch 64
    for shp in [(64, 14, 14, 256, 256), (64, 7, 7, 512, 512), (64, 14, 14, 512, 512)]:
        rc, p = _h16p_plan(*shp)
        assert rc == 0 and p['canvas'] == 1 and 100 <= p['nwg'] <= 512, (shp, p)


def test_winograd_plan_keeps_the_measured_choices():
    """Layer shapes whose Winograd launches were tuned in rounds 2 / 3 keep their image-aligned regions and N tiles; explicit K
    ranges are honoured and clamped."""
    assert _wino_plan(64, 64, 64, 128, 256) == dict(bn=128, ksplit=1, canvas=0, gy=1, gx=1, nwg=1024, kb_per=4, ncanvas=64)
    assert _wino_plan(64, 64, 64, 128, 64)['bn'] == 64 and _wino_plan(64, 56, 56, 64, 64)['nwg'] == 512
    assert _wino_plan(64, 28, 28, 128, 128) == dict(bn=64, ksplit=1, canvas=0, gy=1, gx=1, nwg=256, kb_per=4, ncanvas=64)
    assert _wino_plan(8, 64, 64, 128, 256) == dict(bn=64, ksplit=1, canvas=0, gy=1, gx=1, nwg=256, kb_per=4, ncanvas=8)
    p = _wino_plan(64, 14, 14, 256, 256, tile=71, ksplit=3)
    assert p['bn'] == 64 and p['ksplit'] == 3 and p['kb_per'] == 3
    assert _wino_plan(64, 14, 14, 256, 256, ksplit=1)['ksplit'] == 1
    assert _wino_plan(64, 14, 14, 64, 64, ksplit=16)['ksplit'] == 2      # (two 32-channel blocks: at most two ranges)


def test_winograd_plan_refuses_an_inadmissible_forced_split():
    """A forced K-range count with no admissible candidate (Cout % 4 != 0: the second pass stores channel quads) is an error
    from the plan query -- not an all-zero plan with rc 0 (the launcher then divided by zero on the host)."""
    d = _lib.TapConv()
    d.Hin = d.Hout = d.Hm = d.Win = d.Wout = d.Wm = 14
    d.Cin, d.Cout, d.B, d.s_in, d.s_out, d.nclass = 256, 126, 8, 1, 1, 1
    d.in_cstride, d.out_cstride = 256, 126
    d.cls[0].ntaps, d.cls[0].K, d.cls[0].Kpad = 16, 16 * 256, 16 * 256 + 128
    d.tile, d.ksplit = 70, 4
    wp = (ctypes.c_int32 * 8)()
    assert _lib.load().spaa_tapconv_wino_plan(ctypes.byref(d), wp) != 0
    d.ksplit = 0                        # chosen by the plan: falls back to no split
    assert _lib.load().spaa_tapconv_wino_plan(ctypes.byref(d), wp) == 0 and wp[1] == 1 and wp[0] in (64, 128)
