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
    assert ctypes.sizeof(_lib.TapConv) == _lib.TapConv.cls.offset + 4 * 32
    # the compiled struct itself (layout probes exported by the library)
    lib = _lib.load()
    assert lib.spaa_tapconv_sizeof() == ctypes.sizeof(_lib.TapConv)
    for i, f in enumerate(('out', 'weights', 'taps', 'gate2', 'mask_out', 'tap_range', 'splitk_ws', 'io_dtype', 'nclass', 'cls')):
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
