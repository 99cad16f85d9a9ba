"""ctypes binding of libspaa_hip.so (include/spaa_hip.h).

The product path has no CPU fallback: if the shared library is missing or a kernel launch fails, this module
raises.  PyTorch is used only as plumbing (device memory, current stream).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libspaa_hip.so')

MAX_CLASSES = 4
MAX_TAPS = 64
ACT_NONE, ACT_RELU, ACT_RELU_CLAMP1, ACT_LEAKY01 = 0, 1, 2, 3
IO_IN_F16, IO_OUT_F16 = 1, 2
GATE_NONE, GATE_POS, GATE_POS_LE1, GATE_MUL = 0, 1, 2, 3


class TapClass(C.Structure):
    _fields_ = [('oy0', C.c_int32), ('ox0', C.c_int32), ('ntaps', C.c_int32), ('tap_off', C.c_int32),
                ('K', C.c_int32), ('Kpad', C.c_int32), ('w_off', C.c_int64)]


class TapConv(C.Structure):
    _fields_ = [
        ('inp', C.c_void_p), ('Hin', C.c_int32), ('Win', C.c_int32), ('Cin', C.c_int32), ('in_cstride', C.c_int32),
        ('in_coff', C.c_int32),
        ('out', C.c_void_p), ('Hout', C.c_int32), ('Wout', C.c_int32), ('Cout', C.c_int32), ('out_cstride', C.c_int32),
        ('out_coff', C.c_int32),
        ('B', C.c_int32), ('Hm', C.c_int32), ('Wm', C.c_int32), ('s_in', C.c_int32), ('s_out', C.c_int32),
        ('weights', C.c_void_p), ('w_split', C.c_void_p), ('w_half', C.c_void_p), ('taps', C.c_void_p), ('bias', C.c_void_p),
        ('add', C.c_void_p), ('add_cstride', C.c_int32), ('add_coff', C.c_int32),
        ('gate', C.c_void_p), ('gate_cstride', C.c_int32), ('gate_coff', C.c_int32), ('gate_mode', C.c_int32),
        ('act', C.c_int32), ('tile', C.c_int32),
        ('aux_out', C.c_void_p),
        ('gate2', C.c_void_p), ('gate2_cstride', C.c_int32), ('gate2_coff', C.c_int32),
        ('mask_out', C.c_void_p), ('gate_bits', C.c_void_p), ('gate2_bits', C.c_void_p),
        ('tap_range', C.c_int32 * 4),
        ('splitk_ws', C.c_void_p), ('ksplit', C.c_int32), ('nfold', C.c_int32), ('reserved0', C.c_int32),
        ('io_dtype', C.c_int32), ('reserved1', C.c_int32), ('nclass', C.c_int32),
        ('cls', TapClass * MAX_CLASSES),
        ('in2', C.c_void_p), ('in2_cstride', C.c_int32), ('in2_coff', C.c_int32), ('Cin2', C.c_int32), ('reserved2', C.c_int32),
        ('w2_split', C.c_void_p),
    ]


_i, _f, _p, _l, _d = C.c_int, C.c_float, C.c_void_p, C.c_int64, C.c_double

# name -> argument types (all return int)
_SIGNATURES = {
    'spaa_tapconv_f32': [C.POINTER(TapConv), _p],
    'spaa_tapconv_wgrad': [C.POINTER(TapConv), _p, _p, _p, _p, _i, _p],
    'spaa_nchw_to_nhwc4': [_p, _p, _i, _i, _i, _i, _p],
    'spaa_nhwc4_to_nchw': [_p, _p, _i, _i, _i, _i, _p],
    'spaa_warp_coarse_grid': [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p],
    'spaa_warp_finish_grid': [_p, _p, _p, _i, _p],
    'spaa_warp_fwd': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'spaa_linear_small': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'spaa_conv1_pair_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'spaa_conv1_pair_bwd_f16': [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_s2f_h16': [_p, _i, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p],
    'spaa_s2f_x6': [_p, _i, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p],
    'spaa_fs2_h16': [_p, _i, _i, _p, _p, _i, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p],
    'spaa_warp_bwd': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'spaa_warp_taps': [_p, _i, _i, _i, _i, _p, _p, _p],
    'spaa_warp_bwd_gather': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'spaa_warp_bwd_tiled': [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    'spaa_warp_bwd_tiled_sumsq': [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _f, _p, _p, _p, _p],
    'spaa_warp_fwd_taps': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'spaa_warp_bwd_grid': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    'spaa_warp_finish_grid_bwd': [_p, _p, _p, _p, _p, _i, _p],
    'spaa_warp_coarse_grid_bwd': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p],
    'spaa_shading_tail_fwd': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_head_bwd': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_tail_fwd_f16': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_head_bwd_select': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_head_bwd_select_f16': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_head_bwd_f16': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_tail_fwd_g': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_tail_fwd_f16_g': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_head_bwd_select_g': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_shading_head_bwd_select_f16_g': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_relu_gate': [_p, _p, _p, _l, _p],
    'spaa_adam_step': [_p, _p, _p, _p, _l, _f, _d, _d, _f, _f, _i, _p],
    'spaa_rgb2lab': [_p, _p, _i, _p],
    'spaa_ciede2000': [_p, _p, _p, _i, _p],
    'spaa_rgb2lab_bwd': [_p, _p, _p, _i, _p],
    'spaa_ciede2000_bwd': [_p, _p, _p, _p, _p, _i, _p],
    'spaa_stealth_loss_fwd_bwd': [_p, _p, _p, _f, _f, _f, _p, _p, _p, _i, _i, _p],
    'spaa_img_dists': [_p, _p, _p, _i, _p],
    'spaa_train_loss_fwd_bwd': [_p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_ssim': [_p, _p, _p, _p, _i, _i, _i, _p],
    'spaa_add_nhwc4': [_p, _p, _p, _i, _p],
    'spaa_ce_grad': [_p, _i, _p, _f, _p, _i, _p],
    'spaa_masked_step': [_p, _p, _p, _p, _i, _i, _f, _i, _i, _p],
    'spaa_scale_by_map': [_p, _p, _p, _p, _i, _i, _p],
    'spaa_perc_clamp_quant': [_p, _p, _p, _p, _i, _i, _p],
    'spaa_perc_decide': [_p, _i, _p, _i, _f, _p, _i, _i, _p, _f, _f, _p, _p, _i, _p],
    'spaa_track_where': [_p, _p, _p, _i, _i, _p],
    'spaa_preproc_fwd': [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _p],
    'spaa_preproc_bwd': [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), _p],
    'spaa_maxpool3s2_fwd': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'spaa_maxpool3s2_bwd': [_p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    'spaa_maxpool_fwd': [_p, _p, _p] + [_i] * 11 + [_p],
    'spaa_maxpool_bwd': [_p, _p, _i, _p] + [_i] * 11 + [_p],
    'spaa_maxpool_fwd_f16': [_p, _p, _p] + [_i] * 11 + [_p],
    'spaa_gate_mask': [_p, _i, _p, _l, _i, _i, _i, _p],
    'spaa_maxpool_bwd_f16': [_p, _p, _i, _p] + [_i] * 11 + [_p],
    'spaa_avgpool_fwd_f16': [_p, _p, _i, _i, _i, _p],
    'spaa_avgpool_bwd_f16': [_p, _p, _p, _i, _i, _i, _p],
    'spaa_avgpool2d_fwd': [_p, _p] + [_i] * 11 + [_p],
    'spaa_avgpool2d_bwd': [_p, _p] + [_i] * 11 + [_p],
    'spaa_avgpool2d_fwd_f16': [_p, _p] + [_i] * 11 + [_p],
    'spaa_avgpool2d_bwd_f16': [_p, _p] + [_i] * 11 + [_p],
    'spaa_adaptive_avgpool_fwd': [_p, _p] + [_i] * 6 + [_p],
    'spaa_adaptive_avgpool_bwd': [_p, _p, _p] + [_i] * 6 + [_p],
    'spaa_avgpool_fwd': [_p, _p, _i, _i, _i, _p],
    'spaa_avgpool_bwd': [_p, _p, _p, _i, _i, _i, _p],
    'spaa_decide': [_p, _i, _p, _i, _p, _i, _i, _p, _f, _f, _f, _f, _f, _f, _p, _p, _p, _i, _p],
    'spaa_select_grad': [_p, _p, _p, _p, _p, _i, _i, _p],
    'spaa_prjl2_fwd': [_p, _f, _p, _i, _i, _p],
    'spaa_grad_sumsq': [_p, _p, _f, _f, _p, _p, _i, _i, _p],
    'spaa_step_and_track': [_p, _p, _p, _p, _f, _f, _p, _p, _p, _i, _i, _i, _p],
    'spaa_step_and_track_n': [_p, _p, _p, _i, _p, _f, _f, _p, _p, _p, _i, _i, _i, _p, _p],
    'spaa_zero': [_p, _l, _p],
}

EXPORTS = sorted(list(_SIGNATURES) + ['spaa_version', 'spaa_tapconv_sizeof', 'spaa_tapconv_offsetof', 'spaa_tapconv_wino_plan', 'spaa_tapconv_h16p_plan'])

_lib = None


def load():
    """Loads libspaa_hip.so; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'or `make -C spaa_amd/csrc`. spaa_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.spaa_version.restype = C.c_char_p
    lib.spaa_tapconv_sizeof.restype = C.c_int
    lib.spaa_tapconv_offsetof.argtypes = [C.c_int]
    lib.spaa_tapconv_offsetof.restype = C.c_int
    lib.spaa_tapconv_wino_plan.argtypes = [C.POINTER(TapConv), C.POINTER(C.c_int32)]   # (host-side query: no stream)
    lib.spaa_tapconv_wino_plan.restype = C.c_int
    lib.spaa_tapconv_h16p_plan.argtypes = [C.POINTER(TapConv), C.POINTER(C.c_int32)]   # (host-side query: no stream)
    lib.spaa_tapconv_h16p_plan.restype = C.c_int
    _lib = lib
    return lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _same_device(t):
    """Kernels are launched on torch's CURRENT device and stream with raw pointers: a tensor that lives on another GPU
    would be a wild pointer there (a GPU memory fault, not an exception), so it is refused here.  The Python entry
    points (`spaa`, `PCNet.forward`, `Classifier.classify`, ...) switch to their tensors' device with `on_device`."""
    if t.is_cuda and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f'tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: wrap the '
                           'call in `with torch.cuda.device(tensor.device)` (spaa_amd launches on the current device)')


def on_device(dev):
    """Context manager: make `dev` (a cuda torch.device / tensor device) the current device for the launches inside."""
    dev = torch.device(dev)
    if dev.type != 'cuda':
        raise RuntimeError(f'spaa_amd runs on the GPU only (no CPU fallback); got device={dev}')
    return torch.cuda.device(dev)


def ptr(t):
    if t is None:
        return None
    _same_device(t)
    return C.c_void_p(t.data_ptr())


def hptr(t):
    """Raw pointer of an activation tensor that is fp32 or, in fp16-storage mode, fp16."""
    if t is None:
        return None
    _same_device(t)
    if t.dtype not in (torch.float32, torch.float16) or not t.is_contiguous():
        raise ValueError(f'expected a contiguous fp32/fp16 tensor, got {t.dtype}')
    return C.c_void_p(t.data_ptr())


def check_dev(*tensors, half_ok=False):
    """Kernel operands: contiguous float32 GPU tensors (float16 as well where the caller runs in fp16-storage mode)."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda or not t.is_contiguous() or not (t.dtype == torch.float32 or (half_ok and t.dtype == torch.float16)):
            raise ValueError('spaa_amd kernels need contiguous float32 tensors on the GPU '
                             f'(got device={t.device}, dtype={t.dtype}, contiguous={t.is_contiguous()})')
        _same_device(t)


def check_mask(*tensors):
    """ReLU-gate masks: contiguous uint8 tensors on the current GPU."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda or not t.is_contiguous() or t.dtype != torch.uint8:
            raise ValueError(f'gate masks must be contiguous uint8 GPU tensors (got {t.device}, {t.dtype})')
        _same_device(t)


def pack_gate_mask(act):
    """The mask a `mask_out` launch writes for the activation `act` [..., C] (C % 4 == 0): uint8 [..., C/4], bit e of byte q =
    (act[..., 4q + e] > 0).  Index plumbing with torch ops; used when gates come from somewhere else than a launch
    (parity tests that exchange gates, tools)."""
    c = act.shape[-1]
    assert c % 4 == 0
    bits = (act > 0).view(*act.shape[:-1], c // 4, 4).to(torch.uint8)
    wgt = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=act.device)
    return (bits * wgt).sum(dim=-1).to(torch.uint8).contiguous()


PROFILE = None  # bench.py's instrumented pass: a list that receives (entry point, start event, end event) per launch


def call(name, *args):
    """Invoke an entry point on torch's current device and stream; non-zero return -> RuntimeError."""
    lib = load()
    if PROFILE is None:
        rc = getattr(lib, name)(*args, _stream())
    else:  # HIP events on the launch stream around this one entry point
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args, _stream())
        e1.record()
        PROFILE.append((name, e0, e1))
    if rc != 0:
        raise RuntimeError(f'{name} failed with HIP error {rc}')
