"""`torch.library` registration of the HIP entry points (namespace `spaa`): the PyTorch-ROCm custom ops behind the
reference-compatible Python API (BASELINE.json north_star: "driven from Python via PyTorch-ROCm custom ops so
spaa_attack()/PCNet.forward() stay API-compatible").

Every op is registered for the CUDA(HIP) dispatch key only — a CPU tensor is a dispatcher error, there is no CPU kernel
— with a fake (meta) implementation for shape inference and, where the reference differentiates through the function,
an autograd formula whose backward is itself a registered op.  The implementations call the C-ABI library through
`spaa_amd._lib` (ctypes stays the binding of `include/spaa_hip.h`; it is also what the tests drive directly).

  spaa::nchw_to_nhwc4, spaa::nhwc4_to_nchw      layout at the NCHW boundary
  spaa::rgb2lab (+ _backward)                   perc_al/differential_color_functions.py:39-64
  spaa::ciede2000 (+ _backward)                 perc_al/differential_color_functions.py:109-180
  spaa::stealth_loss                            projector_based_attack.py:279-287 with its gradient (fused)
  spaa::warp (+ _backward)                      models.py:163-185  (WarpingNet.forward on a prebuilt fine grid)
  spaa::pcnet_forward (+ _backward)             models.py:335-346  (engine looked up by an integer handle)
  spaa::classify (+ _backward)                  classifier.py:59-60 (preprocessing + network body -> logits)
"""
import torch
from torch import Tensor

from . import _lib

_HANDLES = {}   # integer handle -> (weakly held) module: custom ops take tensors and plain scalars only


def handle_of(obj):
    import weakref
    h = id(obj)
    if h not in _HANDLES:
        _HANDLES[h] = weakref.ref(obj, lambda _r, h=h: _HANDLES.pop(h, None))
    return h


def _obj(h):
    r = _HANDLES.get(h)
    o = r() if r is not None else None
    if o is None:
        raise RuntimeError(f'spaa op called with a stale module handle {h}')
    return o


def _npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError('spaa ops run on the GPU only (no CPU kernel is registered)')


# ---------------------------------------------------------------------------------------------------------------------
@torch.library.custom_op('spaa::nchw_to_nhwc4', mutates_args=(), device_types='cuda')
def nchw_to_nhwc4(x: Tensor, clamp01: bool = False) -> Tensor:
    x = x.detach().float().contiguous()
    b, c, h, w = x.shape
    if c != 3:
        raise RuntimeError('spaa::nchw_to_nhwc4 expects [B,3,H,W]')
    with _lib.on_device(x.device):
        out = torch.zeros(b, h, w, 4, device=x.device)
        _lib.call('spaa_nchw_to_nhwc4', _lib.ptr(x), _lib.ptr(out), b, h, w, int(clamp01))
    return out


@nchw_to_nhwc4.register_fake
def _(x, clamp01=False):
    return x.new_empty(x.shape[0], x.shape[2], x.shape[3], 4, dtype=torch.float32)


@torch.library.custom_op('spaa::nhwc4_to_nchw', mutates_args=(), device_types='cuda')
def nhwc4_to_nchw(x4: Tensor, clamp01: bool = False) -> Tensor:
    b, h, w, _ = x4.shape
    with _lib.on_device(x4.device):
        out = torch.empty(b, 3, h, w, device=x4.device)
        _lib.call('spaa_nhwc4_to_nchw', _lib.ptr(x4), _lib.ptr(out), b, h, w, int(clamp01))
    return out


@nhwc4_to_nchw.register_fake
def _(x4, clamp01=False):
    return x4.new_empty(x4.shape[0], 3, x4.shape[1], x4.shape[2])


# ---- colour ---------------------------------------------------------------------------------------------------------
@torch.library.custom_op('spaa::rgb2lab', mutates_args=(), device_types='cuda')
def rgb2lab(rgb4: Tensor) -> Tensor:
    """rgb4: NHWC4 sRGB in [0,1] -> NHWC4 Lab."""
    with _lib.on_device(rgb4.device):
        lab = torch.zeros_like(rgb4)
        _lib.call('spaa_rgb2lab', _lib.ptr(rgb4), _lib.ptr(lab), _npix(rgb4))
    return lab


@rgb2lab.register_fake
def _(rgb4):
    return torch.empty_like(rgb4)


@torch.library.custom_op('spaa::rgb2lab_backward', mutates_args=(), device_types='cuda')
def rgb2lab_backward(rgb4: Tensor, g_lab4: Tensor) -> Tensor:
    with _lib.on_device(rgb4.device):
        g = torch.zeros_like(rgb4)
        _lib.call('spaa_rgb2lab_bwd', _lib.ptr(rgb4), _lib.ptr(g_lab4.contiguous()), _lib.ptr(g), _npix(rgb4))
    return g


@rgb2lab_backward.register_fake
def _(rgb4, g_lab4):
    return torch.empty_like(rgb4)


def _rgb2lab_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])


def _rgb2lab_bwd(ctx, g):
    return rgb2lab_backward(ctx.saved_tensors[0], g)


rgb2lab.register_autograd(_rgb2lab_bwd, setup_context=_rgb2lab_setup)


@torch.library.custom_op('spaa::ciede2000', mutates_args=(), device_types='cuda')
def ciede2000(lab1: Tensor, lab2: Tensor) -> Tensor:
    """Two NHWC4 Lab images -> [B,H,W] CIEDE2000 map (the reference's constants: `aHP - 39`, ...)."""
    with _lib.on_device(lab1.device):
        de = torch.zeros(lab1.shape[:3], device=lab1.device)
        _lib.call('spaa_ciede2000', _lib.ptr(lab1), _lib.ptr(lab2), _lib.ptr(de), _npix(lab1))
    return de


@ciede2000.register_fake
def _(lab1, lab2):
    return lab1.new_empty(lab1.shape[:3])


@torch.library.custom_op('spaa::ciede2000_backward', mutates_args=(), device_types='cuda')
def ciede2000_backward(lab1: Tensor, lab2: Tensor, g_de: Tensor) -> tuple[Tensor, Tensor]:
    with _lib.on_device(lab1.device):
        g1, g2 = torch.zeros_like(lab1), torch.zeros_like(lab2)
        _lib.call('spaa_ciede2000_bwd', _lib.ptr(lab1), _lib.ptr(lab2), _lib.ptr(g_de.float().contiguous()), _lib.ptr(g1),
                  _lib.ptr(g2), _npix(lab1))
    return g1, g2


@ciede2000_backward.register_fake
def _(lab1, lab2, g_de):
    return torch.empty_like(lab1), torch.empty_like(lab2)


def _de_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _de_bwd(ctx, g):
    g1, g2 = ciede2000_backward(ctx.saved_tensors[0], ctx.saved_tensors[1], g)
    return g1, g2


ciede2000.register_autograd(_de_bwd, setup_context=_de_setup)


@torch.library.custom_op('spaa::stealth_loss', mutates_args=(), device_types='cuda')
def stealth_loss(y4: Tensor, scene4: Tensor, scene_lab4: Tensor, caml2_w: float, camdE_w: float, gscale: float) \
        -> tuple[Tensor, Tensor]:
    """Fused camera-side stealth terms: returns (per-sample sums [B,3] = (sum caml2_px, sum dE_px, sum dE_px^2),
    gradient [B,H,W,4] = gscale * d(caml2_w * caml2 + camdE_w * dE)/dy)."""
    b, h, w, _ = y4.shape
    with _lib.on_device(y4.device):
        nblk = (h * w + 255) // 256
        part = torch.zeros(b, nblk, 3, device=y4.device)
        g = torch.zeros_like(y4)
        _lib.call('spaa_stealth_loss_fwd_bwd', _lib.ptr(y4), _lib.ptr(scene4), _lib.ptr(scene_lab4), float(caml2_w),
                  float(camdE_w), float(gscale), _lib.ptr(g), None, _lib.ptr(part), b, h * w)
    return part.sum(dim=1), g


@stealth_loss.register_fake
def _(y4, scene4, scene_lab4, caml2_w, camdE_w, gscale):
    return y4.new_empty(y4.shape[0], 3), torch.empty_like(y4)


# ---- warp / PCNet / classifier: engine-backed ops (module looked up by handle) ------------------------------------------
@torch.library.custom_op('spaa::pcnet_forward', mutates_args=(), device_types='cuda')
def pcnet_forward(x: Tensor, s: Tensor, handle: int) -> Tensor:
    """PCNet.forward(x, s) (models.py:335-346) on NCHW tensors; `handle` = ops.handle_of(pcnet)."""
    from . import models
    return models._pcnet_forward_impl(_obj(handle), x, s)[0]


@pcnet_forward.register_fake
def _(x, s, handle):
    hc, wc = _obj(handle).warping_net.out_size
    return x.new_empty(x.shape[0], 3, hc, wc)


@torch.library.custom_op('spaa::classify', mutates_args=(), device_types='cuda')
def classify(im: Tensor, handle: int, crop_h: int, crop_w: int) -> Tensor:
    """normalize(resize(center_crop(im))) -> network -> logits (classifier.py:59-60); `handle` = ops.handle_of(classifier)."""
    from . import classifier
    return classifier._classify_impl(_obj(handle), im, (crop_h, crop_w))[0]


@classify.register_fake
def _(im, handle, crop_h, crop_w):
    return im.new_empty(im.shape[0], 1000)


@torch.library.custom_op('spaa::warp', mutates_args=(), device_types='cuda')
def warp(x: Tensor, handle: int) -> Tensor:
    """WarpingNet.forward(x) (models.py:163-185); `handle` = ops.handle_of(warping_net)."""
    from . import models
    return models._warp_forward_impl(_obj(handle), x)[0]


@warp.register_fake
def _(x, handle):
    hc, wc = _obj(handle).out_size
    return x.new_empty(x.shape[0], 3, hc, wc)


# (the engine-backed implementations leave what their input-gradient pass needs on the module — `_last_saved` — and the
# setup_context hooks below pick it up straight after the forward, on the same thread)
def _pcnet_bwd(ctx, gy):
    from . import models
    return models._pcnet_backward_impl(ctx.saved, gy), None, None


def _warp_bwd(ctx, gy):
    from . import models
    return models._warp_backward_impl(ctx.saved, gy), None


def _classify_bwd(ctx, g):
    from . import classifier
    return classifier._classify_backward_impl(ctx.saved, g), None, None, None


pcnet_forward.register_autograd(_pcnet_bwd, setup_context=lambda ctx, inputs, output: setattr(ctx, 'saved', _obj(inputs[2])._last_saved))
warp.register_autograd(_warp_bwd, setup_context=lambda ctx, inputs, output: setattr(ctx, 'saved', _obj(inputs[1])._last_saved))
classify.register_autograd(_classify_bwd, setup_context=lambda ctx, inputs, output: setattr(ctx, 'saved', _obj(inputs[1])._last_saved))
# layout ops are each other's adjoint (pad lane dropped / zero-filled); the clamp is only used outside autograd
nchw_to_nhwc4.register_autograd(lambda ctx, g: (nhwc4_to_nchw(g.contiguous()), None))
nhwc4_to_nchw.register_autograd(lambda ctx, g: (nchw_to_nhwc4(g.contiguous()), None))


OPS = ('warp', 'nchw_to_nhwc4', 'nhwc4_to_nchw', 'rgb2lab', 'rgb2lab_backward', 'ciede2000', 'ciede2000_backward', 'stealth_loss',
       'pcnet_forward', 'classify')
