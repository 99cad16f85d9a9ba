"""PCNet (WarpingNet + ShadingNetSPAA) on hand-written HIP kernels, behind the reference's module interface.

Mirrors /root/reference/src/python/models.py:98-185 (WarpingNet), :214-303 (ShadingNetSPAA), :305-346 (PCNet):
same constructor arguments, parameter/buffer names (state_dict of 44 parameters + `mask`, `warping_net.ctrl_pts`)
and `forward(x, s)` semantics.  The modules only *hold* parameters; all arithmetic runs in libspaa_hip.so through
`PCNetEngine` (no PyTorch compute ops on the path, no CPU fallback).

`PCNet.forward(x, s)` is differentiable w.r.t. `x` (torch.autograd.Function around the HIP forward / input-gradient
passes).  Parameters are treated as frozen, as the attack does (projector_based_attack.py:62-67): no weight
gradients are produced.
"""
import copy
import os
import weakref

import torch
import torch.nn as nn

from . import _lib
from . import convplan as cp
from .synthetic import uniform_ctrl_pts


USE_GATE_MASKS = os.environ.get('SPAA_GATE_MASKS', '1') != '0'   # 0: fp32 activations as ReLU gates (A/B measurements)
FUSE_TAIL = os.environ.get('SPAA_FUSE_TAIL', '1') != '0'         # 0: transConv2 / conv6 as separate launches (A/B measurements)
FUSE_SELECT = os.environ.get('SPAA_FUSE_SELECT', '1') != '0'     # 0: spaa_select_grad as its own launch (A/B measurements)
FUSE_SKIP2 = int(os.environ.get('SPAA_FUSE_SKIP2', '31'))        # bits: 1 transConv1 + skipConv2, 2 conv2^T + skipConv2^T, 4 conv2_s^T on the same kernel, 8 conv5 + skipConv3 and conv3^T + skipConv3^T, 16 conv1_s + conv1 (csrc/conv1pair.hip); 0: separate launches (A/B measurements)
FUSE_SKIP2_MIN_PIXELS = int(os.environ.get('SPAA_FUSE_SKIP2_MIN', '16384'))   # B x H/4 x W/4 from which the fused kernel's 4 x 32-pixel regions fill the chip


def C_ptr(t):
    """Raw device pointer of a non-float tensor (index arrays)."""
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def _strip(sd):
    out = {}
    for k, v in sd.items():
        while k.startswith('module.'):
            k = k[len('module.'):]
        out[k] = v
    return out


class _ParamHolder(nn.Module):
    """nn.Conv2d-like holder: parameters only (names/shapes as torch's), never called."""

    def __init__(self, wshape, bshape):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(*wshape))
        self.bias = nn.Parameter(torch.zeros(*bshape))

    def forward(self, *a):  # pragma: no cover
        raise RuntimeError('spaa_amd modules hold parameters only; compute runs in PCNetEngine (HIP)')


def _conv(ci, co, k):
    m = _ParamHolder((co, ci, k, k), (co,))
    fan_in = ci * k * k
    nn.init.kaiming_normal_(m.weight)
    nn.init.uniform_(m.bias, -fan_in ** -0.5, fan_in ** -0.5)
    return m


def _deconv(ci, co, k):
    m = _ParamHolder((ci, co, k, k), (co,))
    bound = (co * k * k) ** -0.5
    nn.init.uniform_(m.weight, -bound, bound)
    nn.init.uniform_(m.bias, -bound, bound)
    return m


class _Identity(nn.Module):
    def forward(self, x):  # pragma: no cover
        return x


class WarpingNet(nn.Module):
    """models.py:98-185.  Parameters: affine_mat [1,2,3], theta [1,38,2], grid_refine_net.{0,2,4,6}."""

    def __init__(self, grid_shape=(6, 6), out_size=(256, 256), with_refine=True):
        super().__init__()
        self.grid_shape = grid_shape
        self.out_size = tuple(out_size)
        self.with_refine = with_refine
        self.name = self.__class__.__name__ if with_refine else self.__class__.__name__ + '_without_refine'
        self.register_buffer('fine_grid', None)
        self.affine_mat = nn.Parameter(torch.Tensor([1, 0, 0, 0, 1, 0]).view(-1, 2, 3))
        self.nctrl = grid_shape[0] * grid_shape[1]
        self.nparam = self.nctrl + 2
        self.register_buffer('ctrl_pts', uniform_ctrl_pts(grid_shape))
        self.theta = nn.Parameter(torch.ones(1, self.nparam, 2) * 1e-3)
        if with_refine:
            mods = [_conv(2, 32, 3), _Identity(), _conv(32, 64, 3), _Identity(), _deconv(64, 32, 2), _Identity(),
                    _deconv(32, 2, 2), _Identity()]
            for i in (0, 2):
                nn.init.normal_(mods[i].weight, 0, 1e-4)
            self.grid_refine_net = nn.Sequential(*mods)
        else:
            self.grid_refine_net = None

    def set_affine(self, affine_vec):
        self.affine_mat.data = torch.Tensor(affine_vec).view(-1, 2, 3).to(self.affine_mat.device)

    def build_fine_grid(self, in_size):
        """The sampling grid [Hout, Wout, 4] = (gx, gy, 0, 0) of models.py:168-178, built once by HIP kernels."""
        dev = self.affine_mat.device
        if dev.type != 'cuda':
            raise RuntimeError('spaa_amd.WarpingNet needs its parameters on the GPU (no CPU fallback)')
        hi, wi = in_size
        ho, wo = self.out_size
        t = self.nctrl
        assert self.theta.shape[1] == t + 2, 'only the reduced TPS form (T+2 parameters) is used by the reference'
        coarse = torch.zeros(1, ho, wo, 4, device=dev)
        aff = self.affine_mat.detach().float().contiguous().view(-1)
        theta = self.theta.detach().float().contiguous().view(-1)
        ctrl = self.ctrl_pts.detach().float().contiguous().view(-1)
        _lib.call('spaa_warp_coarse_grid', _lib.ptr(aff), _lib.ptr(theta), _lib.ptr(ctrl), t, hi, wi, ho, wo,
                  _lib.ptr(coarse))
        refine = None
        if self.with_refine:
            if ho % 4 or wo % 4:
                raise ValueError('grid refinement net needs an output size divisible by 4 (as in the reference)')
            g = self.grid_refine_net
            p0 = cp.conv_fwd_plan(g[0].weight, g[0].bias, 2, 1, dev, 'refine0')
            p2 = cp.conv_fwd_plan(g[2].weight, g[2].bias, 2, 1, dev, 'refine2')
            p4 = cp.deconv_fwd_plan(g[4].weight, g[4].bias, 2, 0, dev, 'refine4')
            p6 = cp.deconv_fwd_plan(g[6].weight, g[6].bias, 2, 0, dev, 'refine6')
            r0 = torch.zeros(1, ho // 2, wo // 2, 32, device=dev)
            r2 = torch.zeros(1, ho // 4, wo // 4, 64, device=dev)
            r4 = torch.zeros(1, ho // 2, wo // 2, 32, device=dev)
            refine = torch.zeros(1, ho, wo, 4, device=dev)
            p0.run(coarse, r0, act=_lib.ACT_RELU)
            p2.run(r0, r2, act=_lib.ACT_RELU)
            p4.run(r2, r4, act=_lib.ACT_RELU)
            p6.run(r4, refine, act=_lib.ACT_LEAKY01)
        fine = torch.zeros(ho, wo, 4, device=dev)
        _lib.call('spaa_warp_finish_grid', _lib.ptr(coarse), _lib.ptr(refine), _lib.ptr(fine), ho * wo)
        return fine

    def simplify(self, x):
        """models.py:149-161: cache the fine grid (stored [1,H,W,2] like the reference's buffer)."""
        self.fine_grid = self.build_fine_grid(x.shape[-2:])[None, :, :, :2].contiguous()

    def forward(self, x):
        """models.py:163-185: warp a [B,3,H,W] image (HIP grid_sample, differentiable w.r.t. x): the registered custom op
        `spaa::warp` (spaa_amd/ops.py)."""
        from . import ops
        return torch.ops.spaa.warp(x, ops.handle_of(self))


class ShadingNetSPAA(nn.Module):
    """models.py:214-303 (parameters only; compute in PCNetEngine)."""

    def __init__(self, use_rough=True):
        super().__init__()
        self.use_rough = use_rough
        self.name = self.__class__.__name__ if use_rough else self.__class__.__name__ + '_no_rough'
        self.conv1 = _conv(3, 32, 3)
        self.conv2 = _conv(32, 64, 3)
        self.conv3 = _conv(64, 128, 3)
        self.conv4 = _conv(128, 256, 3)
        self.conv5 = _conv(256, 128, 3)
        nch = 6 if use_rough else 3
        self.conv1_s = _conv(nch, 32, 3)
        self.conv2_s = _conv(32, 64, 3)
        self.conv3_s = _conv(64, 128, 3)
        self.conv4_s = _conv(128, 256, 3)
        self.transConv1 = _deconv(128, 64, 3)
        self.transConv2 = _deconv(64, 32, 2)
        self.conv6 = _conv(32, 3, 3)
        self.skipConv1 = nn.Sequential(_conv(3, 3, 1), _Identity(), _conv(3, 3, 3), _Identity(), _conv(3, 3, 3),
                                       _Identity())
        self.skipConv2 = _conv(32, 64, 1)
        self.skipConv3 = _conv(64, 128, 3)
        for n in ('res1_s', 'res2_s', 'res3_s', 'res4_s'):
            self.register_buffer(n, None)


class PCNet(nn.Module):
    """models.py:305-346."""

    def __init__(self, mask=None, warping_net=None, shading_net=None, fix_shading_net=False, use_mask=True,
                 use_rough=True):
        super().__init__()
        self.name = self.__class__.__name__
        self.use_mask = use_mask
        self.use_rough = use_rough
        if not use_mask:
            self.name += '_no_mask'
        if not use_rough:
            self.name += '_no_rough'

        def unwrap(m):
            return copy.deepcopy(m.module if hasattr(m, 'module') else m)

        if warping_net is None:
            out_size = tuple(mask.shape[-2:]) if mask is not None else (256, 256)
            self.warping_net = WarpingNet(out_size=out_size)
        else:
            self.warping_net = unwrap(warping_net)
        self.shading_net = unwrap(shading_net) if shading_net is not None else ShadingNetSPAA(use_rough)
        if use_mask:
            self.register_buffer('mask', mask.clone().float())
        for p in self.shading_net.parameters():
            p.requires_grad = not fix_shading_net
        self._engines = {}

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(_strip(state_dict), strict)
        self._engines = {}
        return r

    def engine(self, batch, prj_size, owner=None, storage='f32'):
        """Engine (packed weights + workspaces) for a batch size / projector size; `storage` = 'f32' (default) or 'f16'
        (fp16-storage mode, BASELINE.json configs[4]: activations and gradients fp16 in HBM, images and accumulation fp32).  Engines are cached, but an engine
        handed to an `owner` (an AttackState) is that owner's alone for as long as the owner lives: two attacks built
        from one PCNet never share activation workspaces.  Without an owner (the autograd path) any free engine is
        returned; `_PCNetFn.backward` detects a workspace that has been reused since its forward and recomputes it."""
        key = (batch, tuple(prj_size), self.shading_net.conv1.weight.device, storage)
        pool = self._engines.setdefault(key, [])
        for e in pool:
            if e.owner is None or e.owner() is None:
                break
        else:
            with _lib.on_device(key[2]):
                e = PCNetEngine(self, batch, prj_size, storage)
            pool.append(e)
        e.owner = weakref.ref(owner) if owner is not None else None
        return e

    def invalidate(self):
        """Call after changing parameters in place (packed weights are cached)."""
        self._engines = {}

    def forward(self, x, s):
        """models.py:335-346 through the registered custom op `spaa::pcnet_forward` (differentiable w.r.t. x; gradients w.r.t.
        the parameters are PCNetTrainer's: spaa_amd/train_network.py)."""
        if isinstance(s, torch.Tensor) and s.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError('spaa_amd.PCNet.forward is differentiable w.r.t. the projector image x only: detach the '
                                      'scene `s` (a gradient w.r.t. it would silently be missing)')
        from . import ops
        return torch.ops.spaa.pcnet_forward(x, s, ops.handle_of(self))


# ----------------------------------------------------------------------------------------------------------------
def to_nhwc4(x, clamp01=False):
    """[B,3,H,W] (any float layout) -> NHWC4 on the GPU."""
    x = x.detach().float().contiguous()
    if not x.is_cuda:
        raise RuntimeError('spaa_amd needs CUDA(HIP) tensors; there is no CPU fallback')
    b, c, h, w = x.shape
    assert c == 3
    out = torch.zeros(b, h, w, 4, device=x.device)
    _lib.call('spaa_nchw_to_nhwc4', _lib.ptr(x), _lib.ptr(out), b, h, w, int(clamp01))
    return out


def to_nchw(x4, clamp01=False):
    b, h, w, _ = x4.shape
    out = torch.empty(b, 3, h, w, device=x4.device)
    _lib.call('spaa_nhwc4_to_nchw', _lib.ptr(x4), _lib.ptr(out), b, h, w, int(clamp01))
    return out


TILED_WARP_BWD = os.environ.get('SPAA_TILED_WARP_BWD', '1') != '0'   # LDS-staged grid_sample adjoint (0: the untiled gather)
TAP_TABLE_FWD = os.environ.get('SPAA_TAP_TABLE_FWD', '1') != '0'     # grid_sample forward from the per-attack tap table, 32 x 8 tiles (0: the grid kernel)
GATE_BYTE_Y = os.environ.get('SPAA_GATE_BYTE_Y', '1') != '0'         # fused tail / head: the output's clamp gate as one byte per pixel, no pre-clamp tensor in HBM (0: Ypre written and read)
S2F_H16 = os.environ.get('SPAA_S2F_H16', '1') != '0'                 # fp16 storage: the stride-2 forward forms on csrc/s2f_h16.hip (0: the patch-staged kernel's stride-2 form)
S2F_X6 = os.environ.get('SPAA_S2F_X6', '1') != '0'                   # fp32: conv2 / conv2_s on csrc/s2f_x6.hip (0: the implicit-GEMM bf16x6 tile)
FS2_H16 = os.environ.get('SPAA_FS2_H16', '1') != '0'                 # fp16 storage: the fractional-stride 3 x 3 layers on csrc/fs2_h16.hip (0: the patch-staged kernel's folded form)
FUSE_C1BWD = os.environ.get('SPAA_FUSE_C1BWD', '1') != '0'           # fp16 storage: the input gradients of conv1 / conv1_s as one launch (0: two thin-output launches)
FUSE_SUMSQ = os.environ.get('SPAA_FUSE_SUMSQ', '1') != '0'           # spaa_grad_sumsq as the epilogue of the tiled grid_sample adjoint (0: its own launch)


def transposed_taps(grid, prj_size, cam_size, mask=None, want_table=False):
    """Transposed sampling structure for the deterministic backward of grid_sample (index plumbing, once per grid):
    the 4 bilinear taps of every camera pixel (spaa_warp_taps), sorted by the projector pixel they read (CSR).
    Returns (tap_off [Hp*Wp+1], tap_order [4*Hc*Wc], tap_weight x mask [4*Hc*Wc]) for spaa_warp_bwd_gather."""
    dev = grid.device
    (hp, wp), (hc, wc) = prj_size, cam_size
    hwc, hwp = hc * wc, hp * wp
    tap_src = torch.zeros(hwc * 4, dtype=torch.int32, device=dev)
    tap_w = torch.zeros(hwc * 4, device=dev)
    _lib.call('spaa_warp_taps', _lib.ptr(grid), hp, wp, hc, wc, C_ptr(tap_src), _lib.ptr(tap_w))
    order = torch.argsort(tap_src, stable=True)
    bounds = torch.searchsorted(tap_src[order].to(torch.int64), torch.arange(hwp + 1, dtype=torch.int64, device=dev))
    # bilinear weight x mask of the camera pixel the tap belongs to (entry 4*campix + tap)
    tap_wm = (tap_w.view(hwc, 4) * mask.view(hwc, 1)).reshape(-1).contiguous() if mask is not None else tap_w
    if want_table:   # (+ the per-camera-pixel table itself: spaa_warp_fwd_taps reads it, the same weights as the backward pass)
        return bounds.to(torch.int32).contiguous(), order.to(torch.int32).contiguous(), tap_wm, tap_src
    return bounds.to(torch.int32).contiguous(), order.to(torch.int32).contiguous(), tap_wm


TILED_BOX_CAP = int(os.environ.get('SPAA_TILED_BOX_CAP', '576'))   # camera pixels per (16 x 16 projector tile, image) staged in LDS: 4 images x 9 KiB -> 4 workgroups per CU


def tiled_taps(tap_off, tap_order, tap_wm, prj_size, cam_size):
    """Per-tile structure for spaa_warp_bwd_tiled (index plumbing, once per grid): the bounding box in the camera image of
    the tap-list entries of every 16 x 16 tile of projector pixels, each entry's index inside its tile's box, and the weights
    in entry order.  Returns (lidx, w_e, tbox, box_cap) or None when a box would not fit (a wild warp: the untiled gather
    stays in charge).  Reports nothing else: `tbox[:, 2] < 0` marks the tiles gathered from global memory."""
    dev = tap_off.device
    (hp, wp), (hc, wc) = prj_size, cam_size
    ts = 16
    ntx, nty = (wp + ts - 1) // ts, (hp + ts - 1) // ts
    n_ent = int(tap_off[-1])
    if n_ent == 0:
        return None
    order = tap_order[:n_ent].long()
    counts = (tap_off[1:] - tap_off[:-1]).long()
    sp = torch.repeat_interleave(torch.arange(hp * wp, device=dev), counts)          # projector pixel of every entry
    tile = (sp // wp // ts) * ntx + (sp % wp) // ts
    cpix = order >> 2
    cy, cx = cpix // wc, cpix % wc
    big = 1 << 30
    y0 = torch.full((ntx * nty,), big, device=dev, dtype=torch.long).scatter_reduce(0, tile, cy, 'amin')
    x0 = torch.full((ntx * nty,), big, device=dev, dtype=torch.long).scatter_reduce(0, tile, cx, 'amin')
    y1 = torch.full((ntx * nty,), -1, device=dev, dtype=torch.long).scatter_reduce(0, tile, cy, 'amax')
    x1 = torch.full((ntx * nty,), -1, device=dev, dtype=torch.long).scatter_reduce(0, tile, cx, 'amax')
    empty = y1 < 0
    ch, cw = torch.where(empty, 0, y1 - y0 + 1), torch.where(empty, 0, x1 - x0 + 1)
    y0, x0 = torch.where(empty, 0, y0), torch.where(empty, 0, x0)
    # a tile whose box does not fit (clamped grids pile the camera pixels of a whole border strip onto the projector's border
    # pixels) is gathered from global memory: rows = -1, its entries carry the camera pixel itself
    direct = ch * cw > TILED_BOX_CAP
    if bool(direct.float().mean() > 0.5):
        return None
    cap = int(torch.where(direct, 0, ch * cw).max())
    lidx = torch.where(direct[tile], cpix, (cy - y0[tile]) * cw[tile] + (cx - x0[tile])).to(torch.int32).contiguous()
    w_e = tap_wm[order].contiguous()
    tbox = torch.stack([y0, x0, torch.where(direct, -1, ch), cw], 1).to(torch.int32).contiguous()
    return lidx, w_e, tbox, max(cap, 1)


FS2_PAIRS = ((0, 0), (0, 1), (0, 2), (0, 3), (1, 1), (1, 3), (2, 2), (2, 3), (3, 3))   # (operand 2 r + q, class 2 cy + cx): csrc/fs2_h16.hip


def _fs2_lanes(wt):
    """[N, K] matrix -> MFMA A operands [K / 32][N / 16][64 lanes][8]: lane = (row & 15, 8-value chunk g), element e = W[chan(row)][32 ks + 8 g + e].
    MFMA row 16 rb + 4 gr + er carries output channel 32 (rb >> 1) + 8 gr + 4 (rb & 1) + er: a lane of the kernel (which holds rows 4 g ..
    4 g + 3 of every row block) then owns eight consecutive channels per pair of row blocks -- 16-byte stores (csrc/fs2_h16.hip)."""
    n, k = wt.shape
    r = torch.arange(n)
    rb, gr, er = r // 16, (r % 16) // 4, r % 4
    chan = 32 * (rb // 2) + 8 * gr + 4 * (rb % 2) + er
    wp = wt[chan]                                                          # row r of the GEMM = channel chan[r]
    v = wp.reshape(n // 16, 16, k // 32, 4, 8).permute(2, 0, 3, 1, 4)      # [ks][rb][g][row][e]
    return v.reshape(k // 32, n // 16, 64, 8)


def pack_fs2(w_eff, w2=None):
    """Weight images of spaa_fs2_h16 (include/spaa_hip.h).  `w_eff` [3, 3, N, K]: the layer as out[2 y' - 1 + ky] += W[ky][kx][n][k] in[y'][k]
    -- ConvTranspose2d(K, N, 3, 2, 1, 1): weight.permute(2, 3, 1, 0); input gradient of Conv2d(N, K, 3, 2, 1): weight.permute(2, 3, 1, 0) too.
    Returns (w_img fp16 [K/32][9][N/16][64][8], w2_img fp16 [K2/32][N/16][64][8] or None)."""
    w_eff = w_eff.detach().float().cpu()
    imgs = []
    for rq, cl in FS2_PAIRS:
        r, q, cy, cx = rq >> 1, rq & 1, cl >> 1, cl & 1
        ky = 1 if cy == 0 else (2 if r == 0 else 0)
        kx = 1 if cx == 0 else (2 if q == 0 else 0)
        imgs.append(_fs2_lanes(w_eff[ky, kx]))
    w_img = torch.stack(imgs, 1).half().contiguous()                          # [ks][9][rb][64][8]
    w2_img = _fs2_lanes(w2.detach().float().cpu()).half().contiguous() if w2 is not None else None
    return w_img, w2_img


def pack_s2f(w_eff):
    """Weight image of spaa_s2f_h16 (include/spaa_hip.h).  `w_eff` [3, 3, N, K]: out[y][n] = sum W[ky][kx][n][k] in[2 y - 1 + ky][2 x - 1 + kx][k]
    -- Conv2d(K, N, 3, 2, 1): weight.permute(2, 3, 0, 1); input gradient of ConvTranspose2d(N, K, 3, 2, 1, 1): weight.permute(2, 3, 0, 1) too.
    Returns fp16 [K/32][9][N/16][64][8]."""
    w_eff = w_eff.detach().float().cpu()
    return torch.stack([_fs2_lanes(w_eff[t // 3, t % 3]) for t in range(9)], 1).half().contiguous()


def pack_s2f_x6(w_eff):
    """Weight image of spaa_s2f_x6 (include/spaa_hip.h): `w_eff` [3, 3, N, K] as for pack_s2f, every entry split exactly into three bf16 planes
    (cp.split_planes).  Returns int16 (bf16 bits) [K/32][9][3][N/16][64][8]."""
    w_eff = w_eff.detach().float().cpu()
    # K index 8 g + e of a 32-channel step = channel 4 g + e (e < 4) / 16 + 4 g + e - 4: the kernel's two 16-byte loads per lane are bytes
    # [16 g, 16 g + 16) of the pixel's first and second 64 bytes (whole 64-byte segments per load instruction)
    kk = torch.arange(w_eff.shape[3])
    ge, e = (kk % 32) // 8, kk % 8
    perm = (kk // 32) * 32 + torch.where(e < 4, 4 * ge + e, 16 + 4 * ge + e - 4)
    w_eff = w_eff[..., perm]
    img = torch.stack([_fs2_lanes(w_eff[t // 3, t % 3]) for t in range(9)], 1)          # [ks][9][rb][64][8] fp32
    return cp.split_planes(img).permute(1, 2, 0, 3, 4, 5).contiguous()                    # [3][ks][9][...] -> [ks][9][3][rb][64][8]


def pack_pair1_bwd(w_conv1, w_conv1_s):
    """MFMA A operands of spaa_conv1_pair_bwd_f16 (include/spaa_hip.h): [2 sources][4 operands (r, q)][64 lanes][8] fp16.  Lane = (row
    4 (2 cy + cx) + c, 8-channel chunk g); element e = weight[n = 8 g + e][c (+ 3: conv1_s's rough input channels)][ky][kx] of the tap
    through which output-parity class (cy, cx) reads operand in[y + r][x + q] (k3 / s2 / p1: ky = 1 for even rows, 2 / 0 for odd rows)."""
    w1, ws = w_conv1.detach().float().cpu(), w_conv1_s.detach().float().cpu()
    assert tuple(w1.shape) == (32, 3, 3, 3) and tuple(ws.shape) == (32, 6, 3, 3)
    img = torch.zeros(2, 4, 64, 8)
    for src, w, c0 in ((0, w1, 0), (1, ws, 3)):
        for rq in range(4):
            r, q = rq >> 1, rq & 1
            for lane in range(64):
                row, g = lane & 15, lane >> 4
                cl, c = row >> 2, row & 3
                cy, cx = cl >> 1, cl & 1
                if c < 3 and r <= cy and q <= cx:
                    ky = 1 if cy == 0 else (2 if r == 0 else 0)
                    kx = 1 if cx == 0 else (2 if q == 0 else 0)
                    img[src, rq, lane] = w[8 * g:8 * g + 8, c0 + c, ky, kx]
    return img.half().contiguous()


class _Activations(dict):
    """The engine's activation workspaces by name; 'X7' is recomputed on demand when the fused tail kept it in LDS."""

    def __init__(self, eng):
        super().__init__()
        self._eng = eng

    def __getitem__(self, k):
        if k == 'X7':
            self._eng._materialize_x7()
        elif k == 'Ypre':
            self._eng._materialize_ypre()
        return dict.__getitem__(self, k)


class PCNetEngine:
    """Packed weights, sampling grid and workspaces of one PCNet for a fixed batch size; HIP forward and
    input-gradient passes over NHWC4 tensors."""

    def __init__(self, pcnet, batch, prj_size, storage='f32', fuse_skip2=None):
        """`fuse_skip2`: None = where it applies (fp32 storage, enough pixels), False = never (the training step: weight gradients
        and weight refreshes work on the separate layers' plans)."""
        if storage not in ('f32', 'f16'):
            raise ValueError("storage must be 'f32' or 'f16'")
        self.storage = storage
        hd = torch.float16 if storage == 'f16' else torch.float32   # activations / their gradients; images stay fp32
        wn, sn = pcnet.warping_net, pcnet.shading_net
        dev = sn.conv1.weight.device
        if dev.type != 'cuda':
            raise RuntimeError('PCNetEngine needs the model on the GPU (no CPU fallback)')
        self.dev, self.B = dev, batch
        # models.py:342-345: with use_rough the surface branch sees cat([s, x_w * s]) and depends on the projector image;
        # without it the branch sees the scene alone: a constant of the attack, computed once in set_scene()
        self.rough = bool(pcnet.use_rough)
        if sn.conv1_s.weight.shape[1] != (6 if self.rough else 3):
            raise ValueError(f'shading_net.conv1_s takes {sn.conv1_s.weight.shape[1]} channels: use_rough={self.rough} needs {6 if self.rough else 3}')
        self.Hp, self.Wp = prj_size
        self.Hc, self.Wc = wn.out_size
        if self.Hc % 4 or self.Wc % 4:
            raise ValueError('camera size must be divisible by 4')
        fg = getattr(wn, 'fine_grid', None)
        if fg is not None:
            self.grid = torch.zeros(self.Hc, self.Wc, 4, device=dev)
            self.grid[..., :2] = fg[0]
        else:
            self.grid = wn.build_fine_grid(prj_size)
        self.mask = pcnet.mask.detach().float().contiguous().view(-1).to(dev) if pcnet.use_mask else None
        if self.mask is not None:
            assert self.mask.numel() == self.Hc * self.Wc
        self.tap_off, self.tap_order, self.tap_wm, self.tap_src = transposed_taps(self.grid, prj_size, (self.Hc, self.Wc), self.mask, want_table=True)
        self.tiled = tiled_taps(self.tap_off, self.tap_order, self.tap_wm, prj_size, (self.Hc, self.Wc)) if TILED_WARP_BWD else None
        f, d = {}, {}
        for nm, st in (('conv1', 2), ('conv2', 2), ('conv3', 1), ('conv4', 1), ('conv5', 1), ('conv1_s', 2),
                       ('conv2_s', 2), ('conv3_s', 1), ('conv4_s', 1), ('conv6', 1), ('skipConv3', 1)):
            m = getattr(sn, nm)
            f[nm] = cp.conv_fwd_plan(m.weight, m.bias, st, 1, dev, nm)
            if nm.endswith('_s') and not self.rough:
                continue   # (no gradient through a branch that does not depend on the projector image)
            d[nm] = cp.conv_dgrad_plan(m.weight, st, 1, dev, nm + '_dgrad', in_ch=(3, 6) if nm == 'conv1_s' else None)
        f['skipConv2'] = cp.conv_fwd_plan(sn.skipConv2.weight, sn.skipConv2.bias, 1, 0, dev, 'skipConv2')
        d['skipConv2'] = cp.conv_dgrad_plan(sn.skipConv2.weight, 1, 0, dev, 'skipConv2_dgrad')
        # (fp16 storage: the four parity classes folded into the GEMM columns -- the patch-staged fp16 kernel reads the input once;
        # in fp32 the folded form is slower than the four classes: 16 instead of 9 (class, tap) products)
        f['transConv1'] = cp.deconv_fwd_plan(sn.transConv1.weight, sn.transConv1.bias, 2, 1, dev, 'transConv1',
                                             fold=True if storage == 'f16' else None)
        d['transConv1'] = cp.deconv_dgrad_plan(sn.transConv1.weight, 2, 1, dev, 'transConv1_dgrad')
        # fp32 storage: `transConv1(x5) + skipConv2(x1)` (models.py:293,299) as ONE launch of the patch-staged stride-2 kernel with
        # the 1 x 1 convolution as its second source (csrc/tapconv_x6p.hip: no R2 tensor, no separate launch), its mirror image in
        # the backward pass (`conv2^T(g2) + skipConv2^T(g6)`), and the surface branch's conv2_s input gradient on the same kernel
        self.fuse_skip2 = False
        if FUSE_SKIP2 and fuse_skip2 is not False and storage == 'f32' and USE_GATE_MASKS and batch * (self.Hc // 4) * (self.Wc // 4) >= FUSE_SKIP2_MIN_PIXELS:
            tc = cp.deconv_fwd_plan(sn.transConv1.weight, sn.transConv1.bias, 2, 1, dev, 'transConv1+skipConv2', fold=False)
            c2 = cp.conv_dgrad_plan(sn.conv2.weight, 2, 1, dev, 'conv2_dgrad+skipConv2_dgrad', fold=False)
            if tc.x6p_ok() and c2.x6p_ok() and tuple(sn.skipConv2.weight.shape) == (tc.cout, c2.cout, 1, 1) and c2.cout in (32, 64) and tc.cout in (32, 64):
                tc.attach_second_source(sn.skipConv2.weight, sn.skipConv2.bias)
                c2.attach_second_source(sn.skipConv2.weight.detach()[:, :, 0, 0].t().contiguous(), None)
                if FUSE_SKIP2 & 1:
                    f['transConv1x'] = tc
                if FUSE_SKIP2 & 2:
                    d['conv2x'] = c2
                if self.rough and FUSE_SKIP2 & 4:
                    d['conv2_s'] = cp.conv_dgrad_plan(sn.conv2_s.weight, 2, 1, dev, 'conv2_s_dgrad', fold=False)
                    d['conv2_s'].fixed_tile = 74
                self.fuse_skip2 = True
        # fp16 storage: the same two fusions on the patch-staged fp16 kernel's folded form (csrc/tapconv_h16p.hip, second source of a
        # folded stride-2 transposed layer: skipConv2 101 us + skipConv2^T 64 us of separate launches and the R2 / t1 round trips)
        if FUSE_SKIP2 and fuse_skip2 is not False and storage == 'f16' and batch * (self.Hc // 4) * (self.Wc // 4) >= FUSE_SKIP2_MIN_PIXELS:
            tc = cp.deconv_fwd_plan(sn.transConv1.weight, sn.transConv1.bias, 2, 1, dev, 'transConv1+skipConv2', fold=True)
            c2 = cp.conv_dgrad_plan(sn.conv2.weight, 2, 1, dev, 'conv2_dgrad+skipConv2_dgrad', fold=True)
            if (tc.nfold == 4 and c2.nfold == 4 and tuple(sn.skipConv2.weight.shape) == (tc.cout, c2.cout, 1, 1) and c2.cout in (32, 64)
                    and tc.cout in (32, 64)):
                tc.attach_second_source_h16(sn.skipConv2.weight, sn.skipConv2.bias)
                c2.attach_second_source_h16(sn.skipConv2.weight.detach()[:, :, 0, 0].t().contiguous(), None)
                if FUSE_SKIP2 & 1:
                    f['transConv1x'] = tc
                if FUSE_SKIP2 & 2:
                    d['conv2x'] = c2
                self.fuse_skip2 = True
        # round 6, fp16 storage: the three fractional-stride layers (transConv1 + skipConv2, conv2^T + skipConv2^T, conv2_s^T) on the
        # persistent per-input-pixel kernel: exactly the nine real (class, tap) products, all weights resident in LDS (csrc/fs2_h16.hip)
        self.fs2 = None
        small = batch * self.Hc * self.Wc * 32 < 2 ** 31      # (the persistent stride-2 kernels address every tensor with 32-bit byte offsets; the largest -- X6 in fp16, X1 in fp32 -- holds 32 bytes per camera pixel)
        if (FS2_H16 and small and storage == 'f16' and self.fuse_skip2 and 'transConv1x' in f and 'conv2x' in d and self.rough
                and tuple(sn.transConv1.weight.shape) == (128, 64, 3, 3) and tuple(sn.conv2.weight.shape) == (64, 32, 3, 3)
                and tuple(sn.conv2_s.weight.shape) == (64, 32, 3, 3) and tuple(sn.skipConv2.weight.shape) == (64, 32, 1, 1)):
            sk = sn.skipConv2.weight.detach()[:, :, 0, 0]
            t_img, t_img2 = pack_fs2(sn.transConv1.weight.permute(2, 3, 1, 0), sk)
            c_img, c_img2 = pack_fs2(sn.conv2.weight.permute(2, 3, 1, 0), sk.t())
            s_img, _ = pack_fs2(sn.conv2_s.weight.permute(2, 3, 1, 0))
            self.fs2 = dict(tc=(t_img.to(dev), t_img2.to(dev), (sn.transConv1.bias.detach().float() + sn.skipConv2.bias.detach().float()).contiguous().to(dev)),
                            c2=(c_img.to(dev), c_img2.to(dev)), c2s=(s_img.to(dev),))
            # ... and the three stride-2 FORWARD forms (conv2, conv2_s, transConv1's input gradient) on its sibling (csrc/s2f_h16.hip)
            if S2F_H16 and self.Hc % 4 == 0 and self.Wc % 4 == 0:
                self.fs2.update(f2=(pack_s2f(sn.conv2.weight.permute(2, 3, 0, 1)).to(dev), sn.conv2.bias.detach().float().contiguous().to(dev)),
                                f2s=(pack_s2f(sn.conv2_s.weight.permute(2, 3, 0, 1)).to(dev), sn.conv2_s.bias.detach().float().contiguous().to(dev)),
                                tcd=(pack_s2f(sn.transConv1.weight.permute(2, 3, 0, 1)).to(dev),))
        # fp32: conv2 / conv2_s (32 -> 64, stride 2) on the persistent weights-in-LDS bf16x6 kernel (csrc/s2f_x6.hip).  Frozen weights only
        # (its weight images are packed here, once: the training step, which refreshes the separate layers' plans, passes fuse_skip2=False)
        self.s2fx = None
        if (S2F_X6 and small and fuse_skip2 is not False and storage == 'f32' and self.Hc % 4 == 0 and self.Wc % 4 == 0 and tuple(sn.conv2.weight.shape) == (64, 32, 3, 3)
                and tuple(sn.conv2_s.weight.shape) == (64, 32, 3, 3)):
            self.s2fx = dict(f2=(pack_s2f_x6(sn.conv2.weight.permute(2, 3, 0, 1)).to(dev), sn.conv2.bias.detach().float().contiguous().to(dev)),
                             f2s=(pack_s2f_x6(sn.conv2_s.weight.permute(2, 3, 0, 1)).to(dev), sn.conv2_s.bias.detach().float().contiguous().to(dev)))
        # likewise `conv5(x4) + skipConv3(x2)` (models.py:294,298) and `conv3^T(g3) + skipConv3^T(g5)`: one Winograd launch each over
        # the concatenated input channels, read from two tensors (cp.conv_fwd_plan_2src)
        self.fuse_skip3 = False
        if FUSE_SKIP2 & 8 and fuse_skip2 is not False and (USE_GATE_MASKS or storage == 'f16') and cp.WINOGRAD and self.fuse_skip2:
            c5 = cp.conv_fwd_plan_2src(sn.conv5.weight, sn.skipConv3.weight, sn.conv5.bias.detach() + sn.skipConv3.bias.detach(), dev, 'conv5+skipConv3')
            c3 = cp.conv_dgrad_plan_2src(sn.conv3.weight, sn.skipConv3.weight, dev, 'conv3_dgrad+skipConv3_dgrad')
            if c5 is not None and c3 is not None:
                f['conv5x'], d['conv3x'] = c5, c3
                self.fuse_skip3 = True
        # use_rough: `relu(conv1_s(cat[s, xw * s]))` and `relu(conv1(xw) + res1_s)` (models.py:284-285,295) as ONE launch that reads xw and
        # s once, forms xw * s in registers and keeps res1_s there for conv1's epilogue (csrc/conv1pair.hip); the warp kernel then
        # no longer writes the 8-channel concatenation.  Frozen weights only (the training step refreshes the separate plans).
        self.pair1 = None
        if FUSE_SKIP2 & 16 and fuse_skip2 is not False and self.rough and (USE_GATE_MASKS or storage == 'f16') \
                and tuple(sn.conv1.weight.shape) == (32, 3, 3, 3) and tuple(sn.conv1_s.weight.shape) == (32, 6, 3, 3):
            w1, ws = sn.conv1.weight.detach().float(), sn.conv1_s.weight.detach().float()
            wp = torch.zeros(3, 32, 9, 4, device=dev)
            for gi, wsrc in enumerate((w1, ws[:, 0:3], ws[:, 3:6])):
                wp[gi, :, :, :3] = wsrc.permute(0, 2, 3, 1).reshape(32, 9, 3)
            self.pair1 = (wp.contiguous(), sn.conv1.bias.detach().float().contiguous().to(dev),
                          sn.conv1_s.bias.detach().float().contiguous().to(dev))
        # fp16 storage: the ADJOINT of that pair as one launch too (csrc/conv1pair.hip: conv1^T(g_x1) + s * conv1_s^T(g_s1)[rough] on the
        # fp16 matrix instruction: 268 MB instead of the two thin-output launches' 402 MB per step at batch 64); the raw fp32 weights
        self.pair1_bwd = None
        if FUSE_C1BWD and self.pair1 is not None and storage == 'f16':
            self.pair1_bwd = pack_pair1_bwd(sn.conv1.weight, sn.conv1_s.weight).to(dev)
        f['transConv2'] = cp.deconv_fwd_plan(sn.transConv2.weight, sn.transConv2.bias, 2, 0, dev, 'transConv2')
        d['transConv2'] = cp.deconv_dgrad_plan(sn.transConv2.weight, 2, 0, dev, 'transConv2_dgrad')
        sk = sn.skipConv1
        f['skip1a'] = cp.conv_fwd_plan(sk[0].weight, sk[0].bias, 1, 0, dev, 'skipConv1.0')
        f['skip1b'] = cp.conv_fwd_plan(sk[2].weight, sk[2].bias, 1, 1, dev, 'skipConv1.2')
        f['skip1c'] = cp.conv_fwd_plan(sk[4].weight, sk[4].bias, 1, 1, dev, 'skipConv1.4')
        self.f, self.d = f, d
        B, H, W = batch, self.Hc, self.Wc
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4

        def z(*shape):
            return torch.zeros(*shape, device=dev)

        def zh(*shape):
            return torch.zeros(*shape, device=dev, dtype=hd)

        a = _Activations(self)
        a['xw'], a['cat8'] = z(B, H, W, 4), z(B, H, W, 8)
        a['S1'], a['S2'], a['S3'], a['S4'] = zh(B, H2, W2, 32), zh(B, H4, W4, 64), zh(B, H4, W4, 128), zh(B, H4, W4, 256)
        a['X1'], a['R2'], a['X2'], a['R3'] = zh(B, H2, W2, 32), zh(B, H2, W2, 64), zh(B, H4, W4, 64), zh(B, H4, W4, 128)
        a['X3'], a['X4'], a['X5'] = zh(B, H4, W4, 128), zh(B, H4, W4, 256), zh(B, H4, W4, 128)
        a['X6'], a['X7'] = zh(B, H2, W2, 64), zh(B, H, W, 32)
        a['Y'], a['Ypre'], a['R1'] = z(B, H, W, 4), z(B, H, W, 4), z(B, H, W, 4)
        self.a = a
        # ReLU gates of the activations, one byte per 4 channels (include/spaa_hip.h: mask_out / gate_bits): written by
        # the forward launches' epilogues, read by the input-gradient launches instead of the fp32 activations (for X7
        # alone that is 34 MB instead of 537 MB per backward pass at batch 64)
        self.m = {k: torch.zeros(*a[k].shape[:3], a[k].shape[3] // 4, dtype=torch.uint8, device=dev)
                  for k in ('S1', 'S2', 'S3', 'S4', 'X1', 'X2', 'X3', 'X4', 'X5', 'X6', 'X7')}
        g = {}
        g['P7'], g['P6'], g['P5'], g['P4'] = zh(B, H, W, 32), zh(B, H2, W2, 64), zh(B, H4, W4, 128), zh(B, H4, W4, 256)
        g['S4'], g['P3'], g['t2'], g['P2'] = zh(B, H4, W4, 256), zh(B, H4, W4, 128), zh(B, H4, W4, 64), zh(B, H4, W4, 64)
        g['t1'], g['P1'] = zh(B, H2, W2, 32), zh(B, H2, W2, 32)
        g['S3'], g['S2'], g['S1'] = zh(B, H4, W4, 128), zh(B, H4, W4, 64), zh(B, H2, W2, 32)
        g['xw'], g['xs'], g['x'] = z(B, H, W, 4), z(B, H, W, 4), z(B, self.Hp, self.Wp, 4)
        self.g = g
        self.scene = None
        self._x = None
        self._clamp = 1
        # tail / head fusion (csrc/shading_tail.hip): X7 and its gradient never reach HBM.  Needs the byte gate masks; the
        # training step (weight gradients read X7 and P7) switches it off
        self.fuse_tail = FUSE_TAIL and (USE_GATE_MASKS or storage == 'f16') and H % 2 == 0 and W % 2 == 0
        self._x7_version = -1    # `version` for which a['X7'] holds the activation
        # the output's clamp gate (0 < pre <= 1 per channel) as one byte per pixel: written by the fused tail, read by the fused select head;
        # the pre-clamp tensor a['Ypre'] is then produced on demand only (whoever asks gets it from the same kernel: _materialize_ypre)
        self.gate_y = torch.zeros(B, H, W, dtype=torch.uint8, device=dev) if (self.fuse_tail and GATE_BYTE_Y) else None
        self._ypre_version = -1  # `version` for which a['Ypre'] holds the pre-clamp output
        wt, w6 = sn.transConv2.weight.detach().float().cpu(), sn.conv6.weight.detach().float().cpu()
        assert wt.shape == (64, 32, 2, 2) and w6.shape == (3, 32, 3, 3)
        self.tail = dict(
            w2s=cp.split_planes(wt.permute(2, 3, 1, 0).reshape(128, 64)).to(dev),      # [3][32 (2 py + px) + c][k]
            w2ts=cp.split_planes(wt.permute(0, 2, 3, 1).reshape(64, 128)).to(dev),     # [3][n][32 (2 py + px) + c]
            # fp16 storage: the same two matrices rounded to fp16 (one MFMA per product, as the `w_half` of every other layer of the mode)
            w2h=wt.permute(2, 3, 1, 0).reshape(128, 64).half().contiguous().to(dev) if storage == 'f16' else None,
            w2th=wt.permute(0, 2, 3, 1).reshape(64, 128).half().contiguous().to(dev) if storage == 'f16' else None,
            w6=w6.permute(0, 2, 3, 1).reshape(3, 9, 32).contiguous().to(dev),          # [o][3 ky + kx][c]
            w6h=w6.permute(0, 2, 3, 1).reshape(3, 9, 32).half().contiguous().to(dev) if storage == 'f16' else None,   # (fp16 storage: rounded like every weight of the mode)
            w6t=w6.flip(2, 3).permute(2, 3, 0, 1).reshape(27, 32).contiguous().to(dev),  # [3 t + o][c], taps mirrored
            b2=sn.transConv2.bias.detach().float().contiguous().to(dev), b6=sn.conv6.bias.detach().float().contiguous().to(dev))
        self.owner = None    # weakref to the AttackState this engine is leased to (PCNet.engine)
        self.version = 0     # bumped whenever the activation workspaces are overwritten (set_scene / forward)

    # ------------------------------------------------------------------------------------------------------
    def set_scene(self, scene4):
        """scene4: [B,Hc,Wc,4] camera-captured scene(s); precomputes the loop-invariant skipConv1(s) (models.py:291)."""
        assert scene4.shape == (self.B, self.Hc, self.Wc, 4)
        self.version += 1
        self.scene = scene4
        t0, t1 = torch.zeros_like(scene4), torch.zeros_like(scene4)
        self.f['skip1a'].run(scene4, t0, act=_lib.ACT_RELU)
        self.f['skip1b'].run(t0, t1, act=_lib.ACT_RELU)
        self.f['skip1c'].run(t1, self.a['R1'], act=_lib.ACT_RELU)
        if not self.rough:
            self._surface_branch(scene4)

    def _surface_branch(self, inp):
        a, f, R = self.a, self.f, _lib.ACT_RELU
        m = self.m if (USE_GATE_MASKS or self.storage == 'f16') else {k: None for k in self.m}
        if inp is not None:   # (None: S1 already written by the fused conv1 pair)
            f['conv1_s'].run(inp, a['S1'], act=R, mask_out=m['S1'])
        if self.fs2 is not None and 'f2s' in self.fs2:
            w, bb = self.fs2['f2s']
            _lib.call('spaa_s2f_h16', _lib.hptr(a['S1']), 32, 32, _lib.hptr(w), _lib.ptr(bb), None, None, 1, _lib.hptr(a['S2']), _lib.ptr(m['S2']), 64,
                      self.B, self.Hc // 2, self.Wc // 2)
        elif self.s2fx is not None:
            w, bb = self.s2fx['f2s']
            _lib.call('spaa_s2f_x6', _lib.ptr(a['S1']), 32, 32, C_ptr(w), _lib.ptr(bb), None, None, 1, _lib.ptr(a['S2']),
                      _lib.ptr(m['S2']) if m['S2'] is not None else None, 64, self.B, self.Hc // 2, self.Wc // 2)
        else:
            f['conv2_s'].run(a['S1'], a['S2'], act=R, mask_out=m['S2'])
        f['conv3_s'].run(a['S2'], a['S3'], act=R, mask_out=m['S3'])
        f['conv4_s'].run(a['S3'], a['S4'], act=R, mask_out=m['S4'])

    def warp(self, x4, clamp01=True):
        a = self.a
        _lib.check_dev(x4)
        assert x4.shape == (self.B, self.Hp, self.Wp, 4)
        want_cat8 = self.scene is not None and self.pair1 is None
        if TAP_TABLE_FWD and not want_cat8:
            _lib.call('spaa_warp_fwd_taps', _lib.ptr(x4), C_ptr(self.tap_src), _lib.ptr(self.tap_wm), _lib.ptr(a['xw']), self.B, self.Hp,
                      self.Wp, self.Hc, self.Wc, int(clamp01))
        else:
            _lib.call('spaa_warp_fwd', _lib.ptr(x4), _lib.ptr(self.grid), _lib.ptr(self.mask), _lib.ptr(self.scene),
                      _lib.ptr(a['xw']), _lib.ptr(a['cat8']) if want_cat8 else None, self.B, self.Hp,
                      self.Wp, self.Hc, self.Wc, int(clamp01))
        self._x, self._clamp = x4, int(clamp01)
        return a['xw']

    def forward(self, x4, clamp01=True):
        """PCNet.forward on NHWC4 input [B,Hp,Wp,4]; returns cam_infer [B,Hc,Wc,4] (a workspace view)."""
        if self.scene is None:
            raise RuntimeError('call set_scene() first')
        a, f = self.a, self.f
        R, N = _lib.ACT_RELU, _lib.ACT_NONE
        self.version += 1
        self.warp(x4, clamp01)
        m = self.m if (USE_GATE_MASKS or self.storage == 'f16') else {k: None for k in self.m}
        if self.pair1 is not None:
            wp, b1, bs = self.pair1
            _lib.call('spaa_conv1_pair_fwd', _lib.ptr(a['xw']), _lib.ptr(self.scene), _lib.ptr(wp), _lib.ptr(b1), _lib.ptr(bs),
                      _lib.ptr(a['S1']), _lib.ptr(a['X1']), C_ptr(m['S1']), C_ptr(m['X1']), self.B, self.Hc, self.Wc,
                      int(self.storage == 'f16'))
            self._surface_branch(None)
        else:
            if self.rough:
                self._surface_branch(a['cat8'])
            f['conv1'].run(a['xw'], a['X1'], add=a['S1'], act=R, mask_out=m['X1'])
        if not (self.fuse_skip2 and 'transConv1x' in f):
            f['skipConv2'].run(a['X1'], a['R2'], act=N)
        if self.fs2 is not None and 'f2' in self.fs2:
            w, bb = self.fs2['f2']
            _lib.call('spaa_s2f_h16', _lib.hptr(a['X1']), 32, 32, _lib.hptr(w), _lib.ptr(bb), _lib.hptr(a['S2']), None, 1, _lib.hptr(a['X2']),
                      _lib.ptr(m['X2']), 64, self.B, self.Hc // 2, self.Wc // 2)
        elif self.s2fx is not None:
            w, bb = self.s2fx['f2']
            _lib.call('spaa_s2f_x6', _lib.ptr(a['X1']), 32, 32, C_ptr(w), _lib.ptr(bb), _lib.ptr(a['S2']), None, 1, _lib.ptr(a['X2']),
                      _lib.ptr(m['X2']) if m['X2'] is not None else None, 64, self.B, self.Hc // 2, self.Wc // 2)
        else:
            f['conv2'].run(a['X1'], a['X2'], add=a['S2'], act=R, mask_out=m['X2'])
        if not self.fuse_skip3:
            f['skipConv3'].run(a['X2'], a['R3'], act=N)
        f['conv3'].run(a['X2'], a['X3'], add=a['S3'], act=R, mask_out=m['X3'])
        f['conv4'].run(a['X3'], a['X4'], add=a['S4'], act=R, mask_out=m['X4'])
        if self.fuse_skip3:
            f['conv5x'].run(a['X4'], a['X5'], inp2=a['X2'], act=R, mask_out=m['X5'])
        else:
            f['conv5'].run(a['X4'], a['X5'], add=a['R3'], act=R, mask_out=m['X5'])
        if self.fs2 is not None:
            w1, w2, bsum = self.fs2['tc']
            _lib.call('spaa_fs2_h16', _lib.hptr(a['X5']), 128, 128, _lib.hptr(w1), _lib.hptr(a['X1']), 32, 32, _lib.hptr(w2), _lib.ptr(bsum), None,
                      None, 1, _lib.hptr(a['X6']), _lib.ptr(m['X6']), 64, self.B, self.Hc // 4, self.Wc // 4)
        elif self.fuse_skip2 and 'transConv1x' in f:
            f['transConv1x'].run(a['X5'], a['X6'], inp2=a['X1'], act=R, mask_out=m['X6'])
        else:
            f['transConv1'].run(a['X5'], a['X6'], add=a['R2'], act=R, mask_out=m['X6'])
        if self.fuse_tail:
            t = self.tail
            f16 = self.storage == 'f16'   # (X6 and both layers' weights fp16, fp32 accumulation; X7 in LDS as the fp16 a separate launch would store)
            self._tail(dict.__getitem__(a, 'Y'), None if self.gate_y is not None else dict.__getitem__(a, 'Ypre'), m['X7'], self.gate_y)
            if self.gate_y is None:
                self._ypre_version = self.version
            return a['Y']
        x7 = dict.__getitem__(a, 'X7')
        f['transConv2'].run(a['X6'], x7, act=R, mask_out=m['X7'])
        f['conv6'].run(x7, a['Y'], add=a['R1'], act=_lib.ACT_RELU_CLAMP1, aux_out=dict.__getitem__(a, 'Ypre'))
        self._x7_version = self._ypre_version = self.version
        return a['Y']

    def _tail(self, y, ypre, mask7, gate_y):
        """The fused tail launch (csrc/shading_tail.hip) from a['X6'] into the given outputs; `ypre` or `gate_y` may be None."""
        a, t = self.a, self.tail
        f16 = self.storage == 'f16'   # (X6 and both layers' weights fp16, fp32 accumulation; X7 in LDS as the fp16 a separate launch would store)
        args = [_lib.hptr(a['X6']) if f16 else _lib.ptr(a['X6']), _lib.hptr(t['w2h']) if f16 else _lib.ptr(t['w2s']), _lib.ptr(t['b2']),
                _lib.hptr(t['w6h']) if f16 else _lib.ptr(t['w6']), _lib.ptr(t['b6']), _lib.ptr(a['R1']), _lib.ptr(y),
                _lib.ptr(ypre) if ypre is not None else None, _lib.ptr(mask7)]
        if gate_y is not None:
            _lib.call('spaa_shading_tail_fwd_f16_g' if f16 else 'spaa_shading_tail_fwd_g', *args, _lib.ptr(gate_y), self.B, self.Hc // 2, self.Wc // 2)
        else:
            _lib.call('spaa_shading_tail_fwd_f16' if f16 else 'spaa_shading_tail_fwd', *args, self.B, self.Hc // 2, self.Wc // 2)

    def _materialize_ypre(self):
        """With the gate byte the pre-clamp output never reaches HBM in the loop; whoever asks for a['Ypre'] (parity tests, the unfused
        select path, the autograd route) gets it from the SAME fused kernel run once more on the same X6 (deterministic: bitwise the values
        the gate byte was formed from), into scratch outputs."""
        if self.gate_y is None or self._ypre_version == self.version or self.scene is None:
            return
        if getattr(self, '_ypre_scratch', None) is None:
            self._ypre_scratch = (torch.zeros_like(dict.__getitem__(self.a, 'Y')), torch.zeros_like(self.m['X7']), torch.zeros_like(self.gate_y))
        ys, ms, gs = self._ypre_scratch
        self._tail(ys, dict.__getitem__(self.a, 'Ypre'), ms, gs)
        self._ypre_version = self.version

    def can_select(self):
        """True when `backward(None, select=...)` is served: the fused head kernel takes the per-sample choice between the two
        cotangents and the clamp gate itself (spaa_shading_head_bwd_select), no spaa_select_grad launch."""
        return bool(self.fuse_tail and (USE_GATE_MASKS or self.storage == 'f16') and FUSE_SELECT)

    def sumsq_tiles(self):
        """Partial sums per sample that `backward(..., sumsq=...)` writes (the tiled adjoint's 16 x 16 projector tiles), or 0 when the
        fused form is not served (no tiled structure: the caller launches spaa_grad_sumsq itself)."""
        ok = self.tiled is not None and FUSE_SUMSQ and (USE_GATE_MASKS or self.storage == 'f16')   # (SPAA_GATE_MASKS=0: the A/B backward has no such argument)
        return ((self.Wp + 15) // 16) * ((self.Hp + 15) // 16) if ok else 0

    def backward(self, gP, select=None, input_grad=True, sumsq=None, clamp_bits=None):
        """gP: gradient w.r.t. conv6's pre-activation (already gated by 0 < Ypre <= 1), [B,Hc,Wc,4]; or None with
        `select` = (g_adv, g_col, state): the two candidate cotangents at the network output [B,Hc,Wc,4] and the loop's state
        int32 [B,4] (projector_based_attack.py:302-315), see `can_select`.
        Returns the gradient w.r.t. the projector image x4 [B,Hp,Wp,4] (workspace); `input_grad=False` (the training step: the
        projector image is data, only the parameters' gradients are wanted) stops at the warped image and returns g['xw']."""
        if not USE_GATE_MASKS and self.storage == 'f32':
            return self._backward_float_gates(gP)
        g, d, m = self.g, self.d, self.m
        if self.fuse_tail:
            t = self.tail
            p6 = _lib.hptr(g['P6']) if self.storage == 'f16' else _lib.ptr(g['P6'])
            w2t = _lib.hptr(t['w2th']) if self.storage == 'f16' else _lib.ptr(t['w2ts'])
            if select is not None:
                assert gP is None and self.can_select()
                ga, gc, state = select
                _lib.check_dev(ga, gc)
                assert ga.shape == gc.shape == (self.B, self.Hc, self.Wc, 4)
                assert state.shape == (self.B, 4) and state.dtype == torch.int32 and state.is_contiguous() and state.device == ga.device
                if self.gate_y is not None:   # (the clamp gate as the tail's byte per pixel)
                    _lib.call('spaa_shading_head_bwd_select_g' if self.storage == 'f32' else 'spaa_shading_head_bwd_select_f16_g', _lib.ptr(ga),
                              _lib.ptr(gc), _lib.ptr(state), _lib.ptr(self.gate_y), _lib.ptr(t['w6t']), w2t,
                              _lib.ptr(m['X7']), _lib.ptr(m['X6']), p6, self.B, self.Hc // 2, self.Wc // 2)
                else:
                    _lib.call('spaa_shading_head_bwd_select' if self.storage == 'f32' else 'spaa_shading_head_bwd_select_f16', _lib.ptr(ga),
                              _lib.ptr(gc), _lib.ptr(state), _lib.ptr(self.a['Ypre']), _lib.ptr(t['w6t']), w2t,
                              _lib.ptr(m['X7']), _lib.ptr(m['X6']), p6, self.B, self.Hc // 2, self.Wc // 2)
            else:
                _lib.check_dev(gP)
                assert gP.shape == (self.B, self.Hc, self.Wc, 4) and gP.dtype == torch.float32
                _lib.call('spaa_shading_head_bwd' if self.storage == 'f32' else 'spaa_shading_head_bwd_f16', _lib.ptr(gP), _lib.ptr(t['w6t']),
                          w2t, _lib.ptr(m['X7']), _lib.ptr(m['X6']), p6, self.B, self.Hc // 2, self.Wc // 2)
        else:
            d['conv6'].run(gP, g['P7'], gate_bits=m['X7'])
            d['transConv2'].run(g['P7'], g['P6'], gate_bits=m['X6'])
        if self.fs2 is not None and 'tcd' in self.fs2:
            _lib.call('spaa_s2f_h16', _lib.hptr(g['P6']), 64, 64, _lib.hptr(self.fs2['tcd'][0]), None, None, _lib.ptr(m['X5']), 0, _lib.hptr(g['P5']), None,
                      128, self.B, self.Hc // 2, self.Wc // 2)
        else:
            d['transConv1'].run(g['P6'], g['P5'], gate_bits=m['X5'])
        if self.rough:
            d['conv5'].run(g['P5'], g['P4'], gate_bits=m['X4'], aux_out=g['S4'], gate2_bits=m['S4'])
        else:
            d['conv5'].run(g['P5'], g['P4'], gate_bits=m['X4'])
        d['conv4'].run(g['P4'], g['P3'], gate_bits=m['X3'])
        if self.fuse_skip3:
            d['conv3x'].run(g['P3'], g['P2'], inp2=g['P5'], gate_bits=m['X2'])
        else:
            d['skipConv3'].run(g['P5'], g['t2'])
            d['conv3'].run(g['P3'], g['P2'], add=g['t2'], gate_bits=m['X2'])
        if self.fs2 is not None:
            w1, w2 = self.fs2['c2']
            _lib.call('spaa_fs2_h16', _lib.hptr(g['P2']), 64, 64, _lib.hptr(w1), _lib.hptr(g['P6']), 64, 64, _lib.hptr(w2), None, None,
                      _lib.ptr(m['X1']), 0, _lib.hptr(g['P1']), None, 32, self.B, self.Hc // 4, self.Wc // 4)
        elif self.fuse_skip2 and 'conv2x' in d:
            d['conv2x'].run(g['P2'], g['P1'], inp2=g['P6'], gate_bits=m['X1'])
        else:
            d['skipConv2'].run(g['P6'], g['t1'])
            d['conv2'].run(g['P2'], g['P1'], add=g['t1'], gate_bits=m['X1'])
        if not self.rough:   # the surface branch is a constant: the gradient reaches the warped image through conv1 alone
            d['conv1'].run(g['P1'], g['xw'])
            return self.warp_backward(g['xw'], sumsq, clamp_bits) if input_grad else g['xw']
        # surface branch (depends on x through the rough input x*s)
        d['conv4_s'].run(g['S4'], g['S3'], add=g['P3'], gate_bits=m['S3'])
        d['conv3_s'].run(g['S3'], g['S2'], add=g['P2'], gate_bits=m['S2'])
        if self.fs2 is not None:
            _lib.call('spaa_fs2_h16', _lib.hptr(g['S2']), 64, 64, _lib.hptr(self.fs2['c2s'][0]), None, 0, 0, None, None, _lib.hptr(g['P1']),
                      _lib.ptr(m['S1']), 0, _lib.hptr(g['S1']), None, 32, self.B, self.Hc // 4, self.Wc // 4)
        else:
            d['conv2_s'].run(g['S2'], g['S1'], add=g['P1'], gate_bits=m['S1'])
        # the two 3-channel gradients meet at the warped image: d/d(x_w) = g_direct + g_rough * s (models.py:342); the
        # product and the sum are epilogues of the two thin convolutions instead of extra reads in the gather
        if self.pair1_bwd is not None:
            _lib.call('spaa_conv1_pair_bwd_f16', _lib.hptr(g['P1']), _lib.hptr(g['S1']), _lib.ptr(self.scene), _lib.hptr(self.pair1_bwd),
                      _lib.ptr(g['xw']), self.B, self.Hc, self.Wc)
        else:
            d['conv1_s'].run(g['S1'], g['xs'], gate=self.scene, gate_mode=_lib.GATE_MUL)
            d['conv1'].run(g['P1'], g['xw'], add=g['xs'])
        return self.warp_backward(g['xw'], sumsq, clamp_bits) if input_grad else g['xw']

    def _backward_float_gates(self, gP):
        """The same backward pass reading the fp32 activations as gates (SPAA_GATE_MASKS=0: A/B measurements)."""
        a, g, d = self.a, self.g, self.d
        if not self.rough:
            raise NotImplementedError('SPAA_GATE_MASKS=0 (A/B measurements) covers the use_rough=True network only')
        d['conv6'].run(gP, g['P7'], gate=a['X7'])
        d['transConv2'].run(g['P7'], g['P6'], gate=a['X6'])
        d['transConv1'].run(g['P6'], g['P5'], gate=a['X5'])
        d['conv5'].run(g['P5'], g['P4'], gate=a['X4'], aux_out=g['S4'], gate2=a['S4'])
        d['conv4'].run(g['P4'], g['P3'], gate=a['X3'])
        d['skipConv3'].run(g['P5'], g['t2'])
        d['conv3'].run(g['P3'], g['P2'], add=g['t2'], gate=a['X2'])
        d['skipConv2'].run(g['P6'], g['t1'])
        d['conv2'].run(g['P2'], g['P1'], add=g['t1'], gate=a['X1'])
        d['conv4_s'].run(g['S4'], g['S3'], add=g['P3'], gate=a['S3'])
        d['conv3_s'].run(g['S3'], g['S2'], add=g['P2'], gate=a['S2'])
        d['conv2_s'].run(g['S2'], g['S1'], add=g['P1'], gate=a['S1'])
        d['conv1_s'].run(g['S1'], g['xs'], gate=self.scene, gate_mode=_lib.GATE_MUL)
        d['conv1'].run(g['P1'], g['xw'], add=g['xs'])
        return self.warp_backward(g['xw'])

    def refresh_masks(self):
        """Recompute the gate masks from the activation buffers (after a test has overwritten the activations)."""
        for k, mk in self.m.items():
            mk.copy_(_lib.pack_gate_mask(self.a[k].float()))
        if self.gate_y is not None:   # (... and the output's clamp gate from the pre-clamp tensor a test has materialised / overwritten)
            yp = self.a['Ypre'][..., :3]
            ok = (yp > 0) & (yp <= 1)
            self.gate_y.copy_((ok[..., 0].to(torch.uint8) | (ok[..., 1].to(torch.uint8) << 1) | (ok[..., 2].to(torch.uint8) << 2)))

    def _materialize_x7(self):
        """With the fused tail X7 lives in LDS only; whoever asks for a['X7'] (parity tests, tools) gets it recomputed from X6
        by the stand-alone transConv2 launch."""
        if getattr(self, 'fuse_tail', False) and self._x7_version != self.version and self.scene is not None:
            self.f['transConv2'].run(dict.__getitem__(self.a, 'X6'), dict.__getitem__(self.a, 'X7'), act=_lib.ACT_RELU)
            self._x7_version = self.version

    def warp_backward(self, g_xw, sumsq=None, clamp_bits=None):
        """Adjoint of the masked grid_sample (models.py:184,340): deterministic gather over the transposed tap lists; the
        mask is folded into the tap weights.  `sumsq` = (partial [B, sumsq_tiles()], gray, prjl2_scale, state): spaa_grad_sumsq folded
        into the tiled kernel's epilogue (only with sumsq_tiles() > 0).  `clamp_bits` [B, Hp * Wp] uint8 (with `sumsq` only): the clamp
        gate's comparisons for the x of the last forward pass, as spaa_step_and_track_n wrote them (the caller answers for that)."""
        g = self.g
        if sumsq is not None:
            assert self.sumsq_tiles() > 0
            part, gray, scale, state = sumsq
            assert part.shape == (self.B, self.sumsq_tiles()) and part.dtype == torch.float32 and part.is_contiguous()
            if clamp_bits is not None:
                assert clamp_bits.shape == (self.B, self.Hp * self.Wp) and clamp_bits.dtype == torch.uint8 and clamp_bits.is_contiguous()
            lidx, w_e, tbox, cap = self.tiled
            _lib.call('spaa_warp_bwd_tiled_sumsq', _lib.ptr(g_xw), _lib.ptr(self._x), C_ptr(self.tap_off), C_ptr(lidx), _lib.ptr(w_e),
                      C_ptr(tbox), cap, _lib.ptr(g['x']), self.B, self.Hp, self.Wp, self.Hc, self.Wc, self._clamp, float(gray), float(scale),
                      _lib.ptr(state), _lib.ptr(part), _lib.ptr(clamp_bits) if clamp_bits is not None else None)
            return g['x']
        if self.tiled is not None:
            lidx, w_e, tbox, cap = self.tiled
            _lib.call('spaa_warp_bwd_tiled', _lib.ptr(g_xw), _lib.ptr(self._x), C_ptr(self.tap_off), C_ptr(lidx), _lib.ptr(w_e),
                      C_ptr(tbox), cap, _lib.ptr(g['x']), self.B, self.Hp, self.Wp, self.Hc, self.Wc, self._clamp)
            return g['x']
        _lib.call('spaa_warp_bwd_gather', _lib.ptr(g_xw), None, _lib.ptr(self._x), None, None, C_ptr(self.tap_off),
                  C_ptr(self.tap_order), _lib.ptr(self.tap_wm), _lib.ptr(g['x']), self.B, self.Hp, self.Wp, self.Hc,
                  self.Wc, self._clamp)
        return g['x']

    def flops_fwd(self):
        B, H, W = self.B, self.Hc, self.Wc
        sizes = {'conv1_s': (H // 2, W // 2), 'conv2_s': (H // 4, W // 4), 'conv3_s': (H // 4, W // 4),
                 'conv4_s': (H // 4, W // 4), 'conv1': (H // 2, W // 2), 'skipConv2': (H // 2, W // 2),
                 'conv2': (H // 4, W // 4), 'skipConv3': (H // 4, W // 4), 'conv3': (H // 4, W // 4),
                 'conv4': (H // 4, W // 4), 'conv5': (H // 4, W // 4), 'transConv1': (H // 2, W // 2),
                 'transConv2': (H, W), 'conv6': (H, W)}
        return sum(self.f[k].flops(B, *v) for k, v in sizes.items())


def _pcnet_forward_impl(pcnet, x, s):
    """spaa::pcnet_forward.  Returns (y NCHW, saved) — `saved` is what the input-gradient pass needs."""
    b = x.shape[0]
    with _lib.on_device(x.device):
        eng = pcnet.engine(b, x.shape[-2:])
        s4 = to_nhwc4(s.expand(b, -1, -1, -1) if s.shape[0] != b else s)
        eng.set_scene(s4)
        x4 = to_nhwc4(x)
        y4 = eng.forward(x4, clamp01=False)
        saved = dict(eng=eng, version=eng.version, x4=x4, s4=s4, pcnet=weakref.ref(pcnet), key=(b, tuple(x.shape[-2:])))
        pcnet._last_saved = saved
        return to_nchw(y4), saved


def _pcnet_backward_impl(saved, gy):
    eng = saved['eng']
    with _lib.on_device(gy.device):
        if eng.version != saved['version']:
            # the engine's workspaces were overwritten by a later forward (y1 = pcnet(x1, s); y2 = pcnet(x2, s);
            # (l1 + l2).backward()): recompute this call's activations from its own saved inputs -- on a free engine if this
            # one has since been leased to an AttackState (whose scene and activations must not be touched)
            if eng.owner is not None and eng.owner() is not None and saved['pcnet']() is not None:
                eng = saved['eng'] = saved['pcnet']().engine(*saved['key'])
            eng.set_scene(saved['s4'])
            eng.forward(saved['x4'], clamp01=False)
            saved['version'] = eng.version
        g4 = to_nhwc4(gy)
        gP = torch.zeros_like(g4)
        state = torch.ones(eng.B, 4, dtype=torch.int32, device=g4.device)  # best_adv=1 -> take the 2nd argument
        _lib.call('spaa_select_grad', _lib.ptr(g4), _lib.ptr(g4), _lib.ptr(state), _lib.ptr(eng.a['Ypre']),
                  _lib.ptr(gP), eng.B, eng.Hc * eng.Wc)
        return to_nchw(eng.backward(gP))


def _warp_forward_impl(wn, x):
    """spaa::warp: WarpingNet.forward (models.py:163-185)."""
    b = x.shape[0]
    with _lib.on_device(x.device):
        x4 = to_nhwc4(x)
        if wn.fine_grid is None:
            grid = wn.build_fine_grid(x.shape[-2:])
        else:
            grid = torch.zeros(*wn.out_size, 4, device=x.device)
            grid[..., :2] = wn.fine_grid[0]
        hc, wc = wn.out_size
        xw = torch.zeros(b, hc, wc, 4, device=x.device)
        _lib.call('spaa_warp_fwd', _lib.ptr(x4), _lib.ptr(grid), None, None, _lib.ptr(xw), None, b, x.shape[-2],
                  x.shape[-1], hc, wc, 0)
        saved = dict(x4=x4, grid=grid, cam=(hc, wc))
        wn._last_saved = saved
        return to_nchw(xw), saved


def _warp_backward_impl(saved, gy):
    """Deterministic gather over transposed tap lists, as in the attack loop (no float atomics)."""
    with _lib.on_device(gy.device):
        g4 = to_nhwc4(gy)
        b, hp, wp, _ = saved['x4'].shape
        hc, wc = saved['cam']
        off, order, wgt = transposed_taps(saved['grid'], (hp, wp), (hc, wc))
        gx = torch.zeros_like(saved['x4'])
        _lib.call('spaa_warp_bwd_gather', _lib.ptr(g4), None, _lib.ptr(saved['x4']), None, None, C_ptr(off), C_ptr(order),
                  _lib.ptr(wgt), _lib.ptr(gx), b, hp, wp, hc, wc, 0)
        return to_nchw(gx)


# ----------------------------------------------------------------------------------------------------------------
# CompenNet / CompenNet++ (models.py:11-94, :188-212): forward only — used once, after the PerC-AL loop, to turn the
# adversarial camera image into a projector image (projector_based_attack.py:357).
class CompenNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.name = 'CompenNet'
        self.conv1, self.conv2, self.conv3 = _conv(3, 32, 3), _conv(32, 64, 3), _conv(64, 128, 3)
        self.conv4, self.conv5 = _conv(128, 256, 3), _conv(256, 128, 3)
        self.conv1_s, self.conv2_s, self.conv3_s = _conv(3, 32, 3), _conv(32, 64, 3), _conv(64, 128, 3)
        self.conv4_s = _conv(128, 256, 3)
        self.transConv1, self.transConv2 = _deconv(128, 64, 2), _deconv(64, 32, 2)
        self.conv6 = _conv(32, 3, 3)
        self.skipConv1 = nn.Sequential(_conv(3, 3, 3), _Identity(), _conv(3, 3, 3), _Identity(), _conv(3, 3, 3),
                                       _Identity())
        self.skipConv2 = _conv(32, 64, 1)
        self.skipConv3 = _conv(64, 128, 1)
        for n in ('res1_s', 'res2_s', 'res3_s', 'res4_s'):
            self.register_buffer(n, None)
        self._plans = None

    def plans(self):
        dev = self.conv1.weight.device
        if self._plans is None or self._plans[0] != dev:
            f = {}
            for nm, st, pad in (('conv1', 2, 1), ('conv2', 2, 1), ('conv3', 1, 1), ('conv4', 1, 1), ('conv5', 1, 1),
                                ('conv1_s', 2, 1), ('conv2_s', 2, 1), ('conv3_s', 1, 1), ('conv4_s', 1, 1),
                                ('conv6', 1, 1), ('skipConv2', 1, 0), ('skipConv3', 1, 0)):
                m = getattr(self, nm)
                f[nm] = cp.conv_fwd_plan(m.weight, m.bias, st, pad, dev, 'compen.' + nm)
            for i, nm in ((0, 'skip1a'), (2, 'skip1b'), (4, 'skip1c')):
                f[nm] = cp.conv_fwd_plan(self.skipConv1[i].weight, self.skipConv1[i].bias, 1, 1, dev, 'compen.' + nm)
            f['transConv1'] = cp.deconv_fwd_plan(self.transConv1.weight, self.transConv1.bias, 2, 0, dev, 'compen.tc1')
            f['transConv2'] = cp.deconv_fwd_plan(self.transConv2.weight, self.transConv2.bias, 2, 0, dev, 'compen.tc2')
            self._plans = (dev, f)
        return self._plans[1]

    def forward_nhwc4(self, x4, s4):
        """models.py:74-94 on NHWC4 tensors [B,H,W,4] (H, W divisible by 4)."""
        f = self.plans()
        b, h, w, _ = x4.shape
        dev = x4.device
        R, N = _lib.ACT_RELU, _lib.ACT_NONE

        def z(*shape):
            return torch.zeros(*shape, device=dev)

        s1, s2 = z(b, h // 2, w // 2, 32), z(b, h // 4, w // 4, 64)
        s3, s4_ = z(b, h // 4, w // 4, 128), z(b, h // 4, w // 4, 256)
        f['conv1_s'].run(s4, s1, act=R)
        f['conv2_s'].run(s1, s2, act=R)
        f['conv3_s'].run(s2, s3, act=R)
        f['conv4_s'].run(s3, s4_, act=R)
        t0, t1, r1 = z(b, h, w, 4), z(b, h, w, 4), z(b, h, w, 4)
        f['skip1a'].run(x4, t0, act=R)
        f['skip1b'].run(t0, t1, act=R)
        f['skip1c'].run(t1, r1, act=R)
        x1, r2, x2, r3 = z(b, h // 2, w // 2, 32), z(b, h // 2, w // 2, 64), z(b, h // 4, w // 4, 64), z(b, h // 4, w // 4, 128)
        f['conv1'].run(x4, x1, add=s1, act=R)
        f['skipConv2'].run(x1, r2, act=N)
        f['conv2'].run(x1, x2, add=s2, act=R)
        f['skipConv3'].run(x2, r3, act=N)
        x3, x4_, x5 = z(b, h // 4, w // 4, 128), z(b, h // 4, w // 4, 256), z(b, h // 4, w // 4, 128)
        f['conv3'].run(x2, x3, add=s3, act=R)
        f['conv4'].run(x3, x4_, add=s4_, act=R)
        f['conv5'].run(x4_, x5, add=r3, act=R)
        x6, x7, y = z(b, h // 2, w // 2, 64), z(b, h, w, 32), z(b, h, w, 4)
        f['transConv1'].run(x5, x6, add=r2, act=R)
        f['transConv2'].run(x6, x7, act=R)
        f['conv6'].run(x7, y, add=r1, act=_lib.ACT_RELU_CLAMP1)
        return y

    def forward(self, x, s):
        b = x.shape[0]
        return to_nchw(self.forward_nhwc4(to_nhwc4(x), to_nhwc4(s.expand(b, -1, -1, -1) if s.shape[0] != b else s)))


class CompenNetPlusplus(nn.Module):
    """models.py:188-212: warp both the image and the surface with WarpingNet, then CompenNet."""

    def __init__(self, warping_net=None, compen_net=None):
        super().__init__()
        self.name = 'CompenNet++'

        def unwrap(m):
            return copy.deepcopy(m.module if hasattr(m, 'module') else m)

        self.warping_net = unwrap(warping_net) if warping_net is not None else WarpingNet()
        self.compen_net = unwrap(compen_net) if compen_net is not None else CompenNet()
        self._grid = None

    def load_state_dict(self, state_dict, strict=True):
        self._grid = None
        self.compen_net._plans = None
        return super().load_state_dict(_strip(state_dict), strict)

    def forward(self, x, s):
        wn = self.warping_net
        b = x.shape[0]
        x4 = to_nhwc4(x)
        s4 = to_nhwc4(s.expand(b, -1, -1, -1) if s.shape[0] != b else s)
        if self._grid is None or self._grid[0] != tuple(x.shape[-2:]):
            self._grid = (tuple(x.shape[-2:]), wn.build_fine_grid(x.shape[-2:]))
        grid = self._grid[1]
        ho, wo = wn.out_size
        hi, wi = x.shape[-2:]
        xw = torch.zeros(b, ho, wo, 4, device=x4.device)
        sw = torch.zeros(b, ho, wo, 4, device=x4.device)
        for src, dst in ((x4, xw), (s4, sw)):
            _lib.call('spaa_warp_fwd', _lib.ptr(src), _lib.ptr(grid), None, None, _lib.ptr(dst), None, b, hi, wi, ho, wo, 0)
        return to_nchw(self.compen_net.forward_nhwc4(xw, sw))
