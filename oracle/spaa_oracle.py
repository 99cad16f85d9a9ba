"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (`spaa_amd/`).

CPU (PyTorch fp32, autograd) restatement of the reference's SPAA hot path, function by function,
used as the parity checker by `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg.  Each function cites the reference file:line (relative to /root/reference/src/python) it follows.

Pinning: the PCNet / colour / spaa() / PerC-AL restatements here are checked against the real
reference (imported with `oracle/ref_shims.py`, this container only) by `tests/golden/make_golden.py`
and against the golden fixtures it commits (`tests/golden/*.npz`) by `tests/test_oracle_golden.py`.
The classifier *network bodies* (torchvision resnet18/vgg16/inception_v3, third-party, absent from
/root/reference and from this image; pinned version torchvision==0.15.1, requirements.txt:2) are
restated from their published architecture: parity for them is UNPINNED (no reference fixture exists);
the wrapper contract and preprocessing (classifier.py:55-72, img_proc.py:110-132) are pinned.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# img_proc.py:110-132
# --------------------------------------------------------------------------------------
def expand_4d(x):
    for _ in range(4 - x.ndim):
        x = x[None]
    return x


def center_crop(x, size):
    h, w = x.shape[-2:]
    th, tw = size
    i = int(round((h - th) / 2.))
    j = int(round((w - tw) / 2.))
    return x[..., i:i + th, j:j + tw]


def resize_area(x, size):
    return F.interpolate(x, size, mode='area')


# --------------------------------------------------------------------------------------
# pytorch_tps.py:29-106
# --------------------------------------------------------------------------------------
def tps_grid(theta, ctrl, size):
    n, _, h, w = size
    grid = theta.new_empty(n, h, w, 3)
    grid[..., 0] = 1.
    grid[..., 1] = torch.linspace(0, 1, w)
    grid[..., 2] = torch.linspace(0, 1, h).unsqueeze(-1)
    ctrl_b = ctrl.expand(n, *ctrl.size()) if ctrl.dim() == 2 else ctrl
    t = ctrl_b.shape[1]
    diff = grid[..., 1:].unsqueeze(-2) - ctrl_b.unsqueeze(1).unsqueeze(1)
    d = torch.sqrt((diff ** 2).sum(-1))
    u = (d ** 2) * torch.log(d + 1e-6)
    w_, a = theta[:, :-3, :], theta[:, -3:, :]
    if t + 2 == theta.shape[1]:  # reduced form (:67-69)
        w_ = torch.cat((-w_.sum(dim=1, keepdim=True), w_), dim=1)
    b = torch.bmm(u.view(n, -1, t), w_).view(n, h, w, 2)
    z = torch.bmm(grid.view(n, -1, 3), a).view(n, h, w, 2) + b
    return (grid[..., 1:] + z) * 2 - 1


# --------------------------------------------------------------------------------------
# models.py:98-185 (WarpingNet), :214-303 (ShadingNetSPAA), :305-346 (PCNet)
# --------------------------------------------------------------------------------------
def _sd(sd, prefix):
    """Accept an optional 'module.' prefix (nested DataParallel checkpoints)."""
    out = {}
    for k, v in sd.items():
        while k.startswith('module.'):
            k = k[len('module.'):]
        out[k] = v
    return {k[len(prefix):]: v for k, v in out.items() if k.startswith(prefix)} if prefix else out


def warping_fine_grid(sd, x_shape, out_size):
    """models.py:163-178 with batch 1 (the reference repeats it B times; it is independent of x)."""
    w = _sd(sd, 'warping_net.')
    _, c, hi, wi = x_shape
    coarse_affine = F.affine_grid(w['affine_mat'], torch.Size([1, c, hi, wi]), align_corners=True).permute(0, 3, 1, 2)
    coarse_tps = tps_grid(w['theta'], w['ctrl_pts'], (1, c) + tuple(out_size))
    g = F.grid_sample(coarse_affine, coarse_tps, align_corners=True)
    if 'grid_refine_net.0.weight' in w:
        r = F.relu(F.conv2d(g, w['grid_refine_net.0.weight'], w['grid_refine_net.0.bias'], 2, 1))
        r = F.relu(F.conv2d(r, w['grid_refine_net.2.weight'], w['grid_refine_net.2.bias'], 2, 1))
        r = F.relu(F.conv_transpose2d(r, w['grid_refine_net.4.weight'], w['grid_refine_net.4.bias'], 2, 0))
        r = F.leaky_relu(F.conv_transpose2d(r, w['grid_refine_net.6.weight'], w['grid_refine_net.6.bias'], 2, 0), 0.1)
        g = r + g
    return torch.clamp(g, min=-1, max=1).permute(0, 2, 3, 1)


def warp(sd, x, out_size, per_batch_grid=False):
    """WarpingNet.forward (models.py:163-185)."""
    if per_batch_grid:  # literal reference composition (grid rebuilt for the whole batch)
        fine = warping_fine_grid_batched(sd, x.shape, out_size)
    else:
        fine = warping_fine_grid(sd, x.shape, out_size).expand(x.shape[0], -1, -1, -1)
    return F.grid_sample(x, fine, align_corners=True)


def warping_fine_grid_batched(sd, x_shape, out_size):
    """Literal models.py:168-176: tps grid repeated B times, refine net run on the repeated batch."""
    w = _sd(sd, 'warping_net.')
    b, c, hi, wi = x_shape
    coarse_affine = F.affine_grid(w['affine_mat'], torch.Size([1, c, hi, wi]), align_corners=True).permute(0, 3, 1, 2)
    coarse_tps = tps_grid(w['theta'], w['ctrl_pts'], (1, c) + tuple(out_size))
    g = F.grid_sample(coarse_affine, coarse_tps, align_corners=True).repeat(b, 1, 1, 1)
    r = F.relu(F.conv2d(g, w['grid_refine_net.0.weight'], w['grid_refine_net.0.bias'], 2, 1))
    r = F.relu(F.conv2d(r, w['grid_refine_net.2.weight'], w['grid_refine_net.2.bias'], 2, 1))
    r = F.relu(F.conv_transpose2d(r, w['grid_refine_net.4.weight'], w['grid_refine_net.4.bias'], 2, 0))
    r = F.leaky_relu(F.conv_transpose2d(r, w['grid_refine_net.6.weight'], w['grid_refine_net.6.bias'], 2, 0), 0.1)
    return torch.clamp(r + g, min=-1, max=1).permute(0, 2, 3, 1)


def shading_net(sd, x, s_list, return_all=False):
    """ShadingNetSPAA.forward (models.py:280-303); `s_list` = argv = (s, x*s)."""
    p = _sd(sd, 'shading_net.')

    def conv(name, t, stride=1, pad=1):
        return F.conv2d(t, p[name + '.weight'], p[name + '.bias'], stride, pad)

    s = torch.cat(s_list, 1)
    res1_s = F.relu(conv('conv1_s', s, 2))
    res2_s = F.relu(conv('conv2_s', res1_s, 2))
    res3_s = F.relu(conv('conv3_s', res2_s))
    res4_s = F.relu(conv('conv4_s', res3_s))
    r = F.relu(conv('skipConv1.0', s_list[0], 1, 0))  # applied to s, not x (Q8)
    r = F.relu(conv('skipConv1.2', r))
    res1 = F.relu(conv('skipConv1.4', r))
    x1 = F.relu(conv('conv1', x, 2) + res1_s)
    res2 = conv('skipConv2', x1, 1, 0)
    x2 = F.relu(conv('conv2', x1, 2) + res2_s)
    res3 = conv('skipConv3', x2)
    x3 = F.relu(conv('conv3', x2) + res3_s)
    x4 = F.relu(conv('conv4', x3) + res4_s)
    x5 = F.relu(conv('conv5', x4) + res3)
    x6 = F.relu(F.conv_transpose2d(x5, p['transConv1.weight'], p['transConv1.bias'], 2, 1, 1) + res2)
    x7 = F.relu(F.conv_transpose2d(x6, p['transConv2.weight'], p['transConv2.bias'], 2, 0))
    ypre = F.relu(conv('conv6', x7) + res1)
    y = torch.clamp(ypre, max=1)
    if return_all:
        return y, dict(res1_s=res1_s, res2_s=res2_s, res3_s=res3_s, res4_s=res4_s, res1=res1, x1=x1, res2=res2,
                       x2=x2, res3=res3, x3=x3, x4=x4, x5=x5, x6=x6, x7=x7, ypre=ypre)
    return y


def compennet_forward(sd, x, s, prefix='compen_net.'):
    """CompenNet.forward (models.py:74-94)."""
    p = _sd(sd, prefix)

    def conv(name, t, stride=1, pad=1):
        return F.conv2d(t, p[name + '.weight'], p[name + '.bias'], stride, pad)

    res1_s = F.relu(conv('conv1_s', s, 2))
    res2_s = F.relu(conv('conv2_s', res1_s, 2))
    res3_s = F.relu(conv('conv3_s', res2_s))
    res4_s = F.relu(conv('conv4_s', res3_s))
    r = F.relu(conv('skipConv1.0', x))
    r = F.relu(conv('skipConv1.2', r))
    res1 = F.relu(conv('skipConv1.4', r))
    x = F.relu(conv('conv1', x, 2) + res1_s)
    res2 = conv('skipConv2', x, 1, 0)
    x = F.relu(conv('conv2', x, 2) + res2_s)
    res3 = conv('skipConv3', x, 1, 0)
    x = F.relu(conv('conv3', x) + res3_s)
    x = F.relu(conv('conv4', x) + res4_s)
    x = F.relu(conv('conv5', x) + res3)
    x = F.relu(F.conv_transpose2d(x, p['transConv1.weight'], p['transConv1.bias'], 2, 0) + res2)
    x = F.relu(F.conv_transpose2d(x, p['transConv2.weight'], p['transConv2.bias'], 2, 0))
    return torch.clamp(F.relu(conv('conv6', x) + res1), max=1)


def compennet_pp_forward(sd, x, s, out_size):
    """CompenNetPlusplus.forward (models.py:204-212): warp x and s, then CompenNet."""
    return compennet_forward(sd, warp(sd, x, out_size), warp(sd, s, out_size))


def pcnet_forward(sd, x, s, per_batch_grid=False, use_rough=True):
    """PCNet.forward (models.py:335-346), use_mask=True; `use_rough=False`: the surface branch sees s alone (:344-345)."""
    full = _sd(sd, '')
    mask = full['mask']
    out_size = mask.shape[-2:]
    xw = warp(sd, x, out_size, per_batch_grid) * mask
    return shading_net(sd, xw, (s, xw * s) if use_rough else (s,))


# --------------------------------------------------------------------------------------
# perc_al/differential_color_functions.py:12-180
# --------------------------------------------------------------------------------------
def rgb2xyz(rgb):
    mt = torch.tensor([[0.4124, 0.3576, 0.1805],
                       [0.2126, 0.7152, 0.0722],
                       [0.0193, 0.1192, 0.9504]])
    mask1 = (rgb > 0.0405).float()
    mask1_no = 1 - mask1
    t = mask1 * (((rgb + 0.055) / 1.055) ** 2.4)
    t = t + mask1_no * (rgb / 12.92)
    t = 100 * t
    return torch.matmul(mt, t.permute(1, 0, 2, 3).contiguous().view(3, -1)).view(
        3, rgb.size(0), rgb.size(2), rgb.size(3)).permute(1, 0, 2, 3)


def xyz_lab(v):
    m0 = (v == 0).float()
    m0_no = 1 - m0
    v = v + 0.0001 * m0
    m1 = (v > 0.008856).float()
    m1_no = 1 - m1
    res = m1 * v ** (1 / 3)
    res = res + m1_no * ((7.787 * v) + (16 / 116))
    return res * m0_no


def rgb2lab_diff(rgb, device=None):
    res = torch.zeros_like(rgb)
    xyz = rgb2xyz(rgb)
    xn, yn, zn = 95.0489, 100, 108.8840
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    res[:, 0] = 116 * xyz_lab(y / yn) - 16
    res[:, 1] = 500 * (xyz_lab(x / xn) - xyz_lab(y / yn))
    res[:, 2] = 200 * (xyz_lab(y / yn) - xyz_lab(z / zn))
    return res


def _deg(n):
    return n * (180. / np.pi)


def _rad(n):
    return n * (np.pi / 180.)


def _hpf(x, y):
    m = ((x == 0) * (y == 0)).float()
    m_no = 1 - m
    t = _deg(torch.atan2(x * m_no, y * m_no))
    return t * (t >= 0).float() + (360 + t) * (t < 0).float()


def _dhpf(c1, c2, h1p, h2p):
    m = ((c1 * c2) == 0).float()
    m_no = 1 - m
    d = h2p - h1p
    return d * m_no * (torch.abs(d) <= 180).float() + (d - 360) * (d > 180).float() * m_no + \
        (d + 360) * (d < -180).float() * m_no


def _ahpf(c1, c2, h1p, h2p):
    m1 = ((c1 * c2) == 0).float()
    m1_no = 1 - m1
    m2 = (torch.abs(h2p - h1p) <= 180).float()
    m2_no = 1 - m2
    m3 = (torch.abs(h2p + h1p) < 360).float()
    m3_no = 1 - m3
    r1 = (h1p + h2p) * m1_no * m2
    r2 = (h1p + h2p + 360.) * m1_no * m2_no * m3
    r3 = (h1p + h2p - 360.) * m1_no * m2_no * m3_no
    return ((r1 + r2 + r3) + (r1 + r2 + r3) * m1) * 0.5


def ciede2000_diff(lab1, lab2, device=None):
    L1, A1, B1 = lab1[:, 0], lab1[:, 1], lab1[:, 2]
    L2, A2, B2 = lab2[:, 0], lab2[:, 1], lab2[:, 2]
    m01 = ((A1 == 0) * (B1 == 0)).float()
    m02 = ((A2 == 0) * (B2 == 0)).float()
    B1 = B1 + 0.0001 * m01
    B2 = B2 + 0.0001 * m02
    C1 = torch.sqrt(A1 ** 2. + B1 ** 2.)
    C2 = torch.sqrt(A2 ** 2. + B2 ** 2.)
    aC = (C1 + C2) / 2.
    G = 0.5 * (1. - torch.sqrt(aC ** 7. / (aC ** 7. + 25 ** 7.)))
    a1P = (1. + G) * A1
    a2P = (1. + G) * A2
    c1P = torch.sqrt(a1P ** 2. + B1 ** 2.)
    c2P = torch.sqrt(a2P ** 2. + B2 ** 2.)
    h1P = _hpf(B1, a1P) * (1 - m01)
    h2P = _hpf(B2, a2P) * (1 - m02)
    dLP = L2 - L1
    dCP = c2P - c1P
    dhP = _dhpf(C1, C2, h1P, h2P)
    dHP = 2. * torch.sqrt(c1P * c2P) * torch.sin(_rad(dhP) / 2.)
    m_no = 1 - torch.max(m01, m02)
    dHP = dHP * m_no
    aL = (L1 + L2) / 2.
    aCP = (c1P + c2P) / 2.
    aHP = _ahpf(C1, C2, h1P, h2P)
    T = 1. - 0.17 * torch.cos(_rad(aHP - 39)) + 0.24 * torch.cos(_rad(2. * aHP)) + \
        0.32 * torch.cos(_rad(3. * aHP + 6.)) - 0.2 * torch.cos(_rad(4. * aHP - 63.))  # Q1: 39
    dRO = 30. * torch.exp(-1. * (((aHP - 275.) / 25.) ** 2.))
    rC = torch.sqrt(aCP ** 7. / (aCP ** 7. + 25. ** 7.))
    sL = 1. + (0.015 * (aL - 50.) ** 2.) / torch.sqrt(20. + (aL - 50.) ** 2.)
    sC = 1. + 0.045 * aCP
    sH = 1. + 0.015 * aCP * T
    rT = -2. * rC * torch.sin(_rad(2. * dRO))
    rs = (dLP / sL) ** 2. + ((dCP / sC) ** 2.) * m_no + ((dHP / sH) ** 2.) * m_no + rT * (dCP / sC) * (dHP / sH) * m_no
    m0 = (rs <= 0).float()
    rs = rs + 0.0001 * m0
    return torch.sqrt(rs) * (1 - m0)


# --------------------------------------------------------------------------------------
# classifier.py:12-75 — wrapper contract; network bodies restated from torchvision's architecture
# --------------------------------------------------------------------------------------
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def classifier_preprocess(im, crop_sz, input_sz):
    """classifier.py:59: normalize(resize(center_crop(expand_4d(im), crop_sz), input_sz))."""
    x = resize_area(center_crop(expand_4d(im), crop_sz), input_sz)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return (x - mean) / std


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        False, 0.0, eps)


def resnet18_forward(sd, x, return_all=False):
    """torchvision.models.resnet18 (eval mode), v0.15.1 architecture.  `return_all`: also every post-ReLU activation
    and the max-pool arg-max (flat input index), for the gate comparisons of tests/test_gpu_parity.py."""
    acts = {}
    x = F.relu(_bn(sd, 'bn1', F.conv2d(x, sd['conv1.weight'], None, 2, 3)))
    acts['c1'] = x
    x, acts['mp_idx'] = F.max_pool2d(x, 3, 2, 1, return_indices=True)
    for li in range(1, 5):
        for bi in range(2):
            p = f'layer{li}.{bi}'
            stride = 2 if (li > 1 and bi == 0) else 1
            idt = x
            o = F.relu(_bn(sd, p + '.bn1', F.conv2d(x, sd[p + '.conv1.weight'], None, stride, 1)))
            acts[p + '.o1'] = o
            o = _bn(sd, p + '.bn2', F.conv2d(o, sd[p + '.conv2.weight'], None, 1, 1))
            if p + '.downsample.0.weight' in sd:
                idt = _bn(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride, 0))
            x = F.relu(o + idt)
            acts[p + '.out'] = x
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    logits = F.linear(x, sd['fc.weight'], sd['fc.bias'])
    return (logits, acts) if return_all else logits


def vgg16_forward(sd, x, return_all=False):
    """torchvision.models.vgg16 (eval), v0.15.1 architecture (configuration 'D', no batch norm).  `return_all`: also the
    post-ReLU activations ('conv<i>', 'fc1', 'fc2') and max-pool arg-maxes ('pool<i>', flat input index) in layer order."""
    idx, acts, nc, npool = 0, {}, 0, 0
    for v in [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']:
        if v == 'M':
            x, acts[f'pool{npool}'] = F.max_pool2d(x, 2, 2, return_indices=True)
            npool += 1
            idx += 1
        else:
            x = F.relu(F.conv2d(x, sd[f'features.{idx}.weight'], sd[f'features.{idx}.bias'], 1, 1))
            acts[f'conv{nc}'] = x
            nc += 1
            idx += 2
    x = F.adaptive_avg_pool2d(x, (7, 7)).flatten(1)
    x = F.relu(F.linear(x, sd['classifier.0.weight'], sd['classifier.0.bias']))
    acts['fc1'] = x
    x = F.relu(F.linear(x, sd['classifier.3.weight'], sd['classifier.3.bias']))
    acts['fc2'] = x
    logits = F.linear(x, sd['classifier.6.weight'], sd['classifier.6.bias'])
    return (logits, acts) if return_all else logits


def inception_v3_forward(sd, x):
    """torchvision.models.inception_v3(transform_input=True) in eval mode (aux head unused), v0.15.1 architecture."""
    def bc(name, t, stride=1, padding=0):
        t = F.conv2d(t, sd[name + '.conv.weight'], None, stride, padding)
        t = F.batch_norm(t, sd[name + '.bn.running_mean'], sd[name + '.bn.running_var'], sd[name + '.bn.weight'],
                         sd[name + '.bn.bias'], False, 0.0, 0.001)
        return F.relu(t)

    x = torch.cat((x[:, 0:1] * (0.229 / 0.5) + (0.485 - 0.5) / 0.5, x[:, 1:2] * (0.224 / 0.5) + (0.456 - 0.5) / 0.5,
                   x[:, 2:3] * (0.225 / 0.5) + (0.406 - 0.5) / 0.5), 1)
    x = bc('Conv2d_1a_3x3', x, 2)
    x = bc('Conv2d_2a_3x3', x)
    x = bc('Conv2d_2b_3x3', x, 1, 1)
    x = F.max_pool2d(x, 3, 2)
    x = bc('Conv2d_3b_1x1', x)
    x = bc('Conv2d_4a_3x3', x)
    x = F.max_pool2d(x, 3, 2)

    def inc_a(n, t):
        b1 = bc(n + '.branch1x1', t)
        b5 = bc(n + '.branch5x5_2', bc(n + '.branch5x5_1', t), 1, 2)
        b3 = bc(n + '.branch3x3dbl_3', bc(n + '.branch3x3dbl_2', bc(n + '.branch3x3dbl_1', t), 1, 1), 1, 1)
        bp = bc(n + '.branch_pool', F.avg_pool2d(t, 3, 1, 1))
        return torch.cat([b1, b5, b3, bp], 1)

    def inc_b(n, t):
        b3 = bc(n + '.branch3x3', t, 2)
        bd = bc(n + '.branch3x3dbl_3', bc(n + '.branch3x3dbl_2', bc(n + '.branch3x3dbl_1', t), 1, 1), 2)
        return torch.cat([b3, bd, F.max_pool2d(t, 3, 2)], 1)

    def inc_c(n, t):
        b1 = bc(n + '.branch1x1', t)
        b7 = bc(n + '.branch7x7_3', bc(n + '.branch7x7_2', bc(n + '.branch7x7_1', t), 1, (0, 3)), 1, (3, 0))
        bd = bc(n + '.branch7x7dbl_1', t)
        bd = bc(n + '.branch7x7dbl_2', bd, 1, (3, 0))
        bd = bc(n + '.branch7x7dbl_3', bd, 1, (0, 3))
        bd = bc(n + '.branch7x7dbl_4', bd, 1, (3, 0))
        bd = bc(n + '.branch7x7dbl_5', bd, 1, (0, 3))
        bp = bc(n + '.branch_pool', F.avg_pool2d(t, 3, 1, 1))
        return torch.cat([b1, b7, bd, bp], 1)

    def inc_d(n, t):
        b3 = bc(n + '.branch3x3_2', bc(n + '.branch3x3_1', t), 2)
        b7 = bc(n + '.branch7x7x3_1', t)
        b7 = bc(n + '.branch7x7x3_2', b7, 1, (0, 3))
        b7 = bc(n + '.branch7x7x3_3', b7, 1, (3, 0))
        b7 = bc(n + '.branch7x7x3_4', b7, 2)
        return torch.cat([b3, b7, F.max_pool2d(t, 3, 2)], 1)

    def inc_e(n, t):
        b1 = bc(n + '.branch1x1', t)
        b3 = bc(n + '.branch3x3_1', t)
        b3 = torch.cat([bc(n + '.branch3x3_2a', b3, 1, (0, 1)), bc(n + '.branch3x3_2b', b3, 1, (1, 0))], 1)
        bd = bc(n + '.branch3x3dbl_2', bc(n + '.branch3x3dbl_1', t), 1, 1)
        bd = torch.cat([bc(n + '.branch3x3dbl_3a', bd, 1, (0, 1)), bc(n + '.branch3x3dbl_3b', bd, 1, (1, 0))], 1)
        bp = bc(n + '.branch_pool', F.avg_pool2d(t, 3, 1, 1))
        return torch.cat([b1, b3, bd, bp], 1)

    for n in ('Mixed_5b', 'Mixed_5c', 'Mixed_5d'):
        x = inc_a(n, x)
    x = inc_b('Mixed_6a', x)
    for n in ('Mixed_6b', 'Mixed_6c', 'Mixed_6d', 'Mixed_6e'):
        x = inc_c(n, x)
    x = inc_d('Mixed_7a', x)
    x = inc_e('Mixed_7b', x)
    x = inc_e('Mixed_7c', x)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, sd['fc.weight'], sd['fc.bias'])


class OracleClassifier:
    """Duck-typed stand-in for classifier.Classifier: same call contract (classifier.py:55-75)."""
    INPUT_SZ = {'resnet18': (224, 224), 'vgg16': (224, 224), 'inception_v3': (299, 299)}

    def __init__(self, name, state_dict, sort_results=True, input_sz=None):
        self.name = name
        self.sd = state_dict
        self.input_sz = tuple(input_sz) if input_sz is not None else self.INPUT_SZ[name]
        self.sort_results = sort_results
        self.body = {'resnet18': resnet18_forward, 'vgg16': vgg16_forward,
                     'inception_v3': globals().get('inception_v3_forward')}[name]

    def __call__(self, im, crop_sz=(240, 240)):
        if im.dtype == torch.uint8:
            im = im.type(torch.float32) / 255
        raw_score = self.body(self.sd, classifier_preprocess(im, crop_sz, self.input_sz))
        p = F.softmax(raw_score, dim=1).detach().cpu()
        if self.sort_results:
            p_sorted, idx = p.sort(descending=True)
        else:
            p_sorted, idx = p, torch.arange(p.shape[1]).repeat(p.shape[0], 1)
        return raw_score, p_sorted.numpy(), idx.numpy()


# --------------------------------------------------------------------------------------
# projector_based_attack.py:212-339 — spaa(), op-for-op (two backward passes, per-iteration grid rebuild)
# --------------------------------------------------------------------------------------
def spaa(pcnet_sd, classifier, target_idx, targeted, cam_scene, d_thr, stealth_loss, setup_info, iters=50,
         adv_lr=2, col_lr=1, p_thresh=0.9, trace=None, per_batch_grid=False):
    """`cam_scene`: [3,H,W]/[1,3,H,W] (reference: one scene x B targets) or [B,3,H,W] (one scene per sample)."""
    num_target = len(target_idx)
    cp_sz = setup_info['classifier_crop_sz']
    cam_scene = expand_4d(cam_scene)
    cam_scene_batch = cam_scene.expand(num_target, -1, -1, -1) if cam_scene.shape[0] == 1 else cam_scene
    assert cam_scene_batch.shape[0] == num_target
    im_gray = setup_info['prj_brightness'] * torch.ones(num_target, 3, *setup_info['prj_im_sz'])
    prj_adv = im_gray.clone()
    prj_adv.requires_grad = True
    adv_w = 1
    prjl2_w = 0.1 if 'prjl2' in stealth_loss else 0
    caml2_w = 1 if 'caml2' in stealth_loss else 0
    camdE_w = 1 if 'camdE' in stealth_loss else 0
    tgt = np.asarray(target_idx)
    prj_adv_best = prj_adv.clone()
    cam_infer_best = cam_scene_batch.clone()
    col_loss_best = 1e6 * torch.ones(num_target)
    for i in range(iters):
        cam_infer = pcnet_forward(pcnet_sd, torch.clamp(expand_4d(prj_adv), 0, 1), cam_scene_batch, per_batch_grid)
        raw_score, p, idx = classifier(cam_infer, cp_sz)
        sel = raw_score[torch.arange(num_target), torch.as_tensor(tgt)]
        adv_loss = adv_w * (-sel).mean() if targeted else adv_w * sel.mean()
        prjl2 = torch.norm(im_gray - prj_adv, dim=1).mean(1).mean(1)
        col_loss_batch = prjl2_w * prjl2
        caml2 = torch.norm(cam_scene_batch - cam_infer, dim=1).mean(1).mean(1)
        col_loss_batch = col_loss_batch + caml2_w * caml2
        camdE = ciede2000_diff(rgb2lab_diff(cam_infer), rgb2lab_diff(cam_scene_batch)).mean(1).mean(1)
        col_loss_batch = col_loss_batch + camdE_w * camdE
        col_loss = col_loss_batch.mean()
        mask_high_conf = p[:, 0] > p_thresh
        mask_high_pert = (caml2 * 255 > d_thr).detach().cpu().numpy()
        if targeted:
            mask_succ_adv = idx[:, 0] == tgt
            mask_best_adv = mask_succ_adv & mask_high_conf & mask_high_pert
        else:
            mask_succ_adv = idx[:, 0] != tgt
            mask_best_adv = mask_succ_adv & mask_high_pert
        adv_loss.backward(retain_graph=True)
        adv_grad = prj_adv.grad.clone()
        prj_adv.grad.zero_()
        prj_adv.data[~mask_best_adv] -= adv_lr * (adv_grad.permute(1, 2, 3, 0) / torch.norm(
            adv_grad.view(adv_grad.shape[0], -1), dim=1)).permute(3, 0, 1, 2)[~mask_best_adv]
        col_loss.backward()
        col_grad = prj_adv.grad.clone()
        prj_adv.grad.zero_()
        prj_adv.data[mask_best_adv] -= col_lr * (col_grad.permute(1, 2, 3, 0) / torch.norm(
            col_grad.view(col_grad.shape[0], -1), dim=1)).permute(3, 0, 1, 2)[mask_best_adv]
        col_loss_best_before = col_loss_best.clone().numpy()
        mask_best_color = (col_loss_batch < col_loss_best).detach().cpu().numpy()
        mask_best = mask_best_color * mask_best_adv
        col_loss_best[mask_best] = col_loss_batch.data[mask_best].clone()
        prj_adv_best[mask_succ_adv] = prj_adv[mask_succ_adv].clone()  # post-step image (Q4)
        cam_infer_best[mask_succ_adv] = cam_infer[mask_succ_adv].clone()
        prj_adv_best[mask_best] = prj_adv[mask_best].clone()
        cam_infer_best[mask_best] = cam_infer[mask_best].clone()
        if trace is not None:
            trace.append(dict(succ=mask_succ_adv.copy(), best_adv=mask_best_adv.copy(), best=np.asarray(mask_best).copy(),
                              top1=idx[:, 0].copy(), p1=p[:, 0].copy(), caml2=caml2.detach().numpy().copy(),
                              camdE=camdE.detach().numpy().copy(), prjl2=prjl2.detach().numpy().copy(),
                              col_loss=col_loss_batch.detach().numpy().copy(), adv_loss=float(adv_loss.detach()),
                              prj_adv=prj_adv.detach().clone().numpy(), col_loss_best_before=col_loss_best_before,
                              target_logit=sel.detach().numpy().copy(), cam_infer=cam_infer.detach().numpy().copy()))
    prj_adv_best = torch.clamp(prj_adv_best, 0, 1)
    return cam_infer_best.detach(), prj_adv_best.detach()


# --------------------------------------------------------------------------------------
# perc_al/__init__.py:133-256 — PerC_AL.adversary_projector
# --------------------------------------------------------------------------------------
def quantization(x):
    return torch.round(x * 255) / 255


def perc_al_adversary_projector(classifier, inputs, labels, d_thr, targeted=True, cp_sz=(240, 240), max_iterations=50,
                                alpha_l_init=1., alpha_c_init=0.5, confidence=0, p_thresh=0.9, stop_after=None,
                                trace=None):
    if inputs.min() < 0 or inputs.max() > 1:
        raise ValueError('Input values should be in the [0, 1] range.')
    alpha_l_min = alpha_l_init / 100
    alpha_c_min = alpha_c_init / 10
    multiplier = -1 if targeted else 1
    x_best = inputs.clone()
    inputs_lab = rgb2lab_diff(inputs)
    bsz = inputs.shape[0]
    delta = torch.zeros_like(inputs, requires_grad=True)
    mask_best_adv = torch.zeros(bsz, dtype=torch.bool)
    bound_best = torch.ones(bsz) * 100000
    if (not targeted) and confidence != 0:
        infhot = torch.zeros(labels.size(0), 1000).scatter_(1, labels.unsqueeze(1), float('inf'))
    if targeted and confidence != 0:
        print('Only support setting confidence in untargeted case!')
        return None
    for i in range(max_iterations):
        if stop_after is not None and i >= stop_after:
            break
        raw_score, p, idx = classifier(inputs + delta, cp_sz)
        alpha_c = alpha_c_min + 0.5 * (alpha_c_init - alpha_c_min) * (1 + math.cos(i / max_iterations * math.pi))
        alpha_l = alpha_l_min + 0.5 * (alpha_l_init - alpha_l_min) * (1 + math.cos(i / max_iterations * math.pi))
        loss = multiplier * F.cross_entropy(raw_score, labels, reduction='sum')
        loss.backward()
        grad_a = delta.grad.clone()
        delta.grad.zero_()
        delta.data[~mask_best_adv] = delta.data[~mask_best_adv] + alpha_l * (grad_a.permute(1, 2, 3, 0) / torch.norm(
            grad_a.reshape(bsz, -1), dim=1)).permute(3, 0, 1, 2)[~mask_best_adv]
        d_map = ciede2000_diff(inputs_lab, rgb2lab_diff(inputs + delta)).unsqueeze(1)
        color_dis = torch.norm(d_map.view(bsz, -1), dim=1)
        color_dis.sum().backward()
        grad_color = delta.grad.clone()
        delta.grad.zero_()
        delta.data[mask_best_adv] = delta.data[mask_best_adv] - alpha_c * (grad_color.permute(1, 2, 3, 0) / torch.norm(
            grad_color.reshape(bsz, -1), dim=1)).permute(3, 0, 1, 2)[mask_best_adv]
        delta.data = (inputs + delta.data).clamp(0, 1) - inputs
        x_round = quantization(inputs + delta.data)
        caml2 = torch.norm(delta.detach(), dim=1).mean(1).mean(1)
        mask_high_pert = (caml2 * 255 > d_thr).detach()
        raw_score, p, idx = classifier(x_round, cp_sz)
        mask_high_conf = torch.tensor(p[:, 0] > p_thresh, dtype=torch.bool)
        if (not targeted) and confidence != 0:
            real = raw_score.gather(1, labels.unsqueeze(1)).squeeze(1)
            other = (raw_score - infhot).max(1)[0]
            mask_isadv = (real - other) <= -40
            mask_best_adv = mask_isadv & mask_high_pert
        elif targeted:
            mask_isadv = torch.tensor(idx[:, 0]) == labels
            mask_best_adv = mask_isadv & mask_high_conf & mask_high_pert
        else:
            mask_isadv = torch.tensor(idx[:, 0]) != labels
            mask_best_adv = mask_isadv & mask_high_pert
        mask_best = (color_dis.data < bound_best) * mask_best_adv
        bound_best[mask_best] = color_dis.data[mask_best].clone()
        x_best[mask_isadv] = x_round[mask_isadv].clone()
        x_best[mask_best] = x_round[mask_best].clone()
        if trace is not None:
            trace.append(dict(delta=delta.detach().clone(), isadv=mask_isadv.clone(), best_adv=mask_best_adv.clone(),
                              color_dis=color_dis.detach().clone(), caml2=caml2.clone(), p1=p[:, 0].copy(),
                              top1=idx[:, 0].copy(), x_round=x_round.clone()))
    return x_best


# --------------------------------------------------------------------------------------
# utils.py:420-491 (calc_img_dists) and pytorch_ssim/__init__.py:9-58
# --------------------------------------------------------------------------------------
def ssim_window(window_size=11, sigma=1.5, channel=3):
    g = torch.Tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * sigma ** 2)) for i in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    return g.mm(g.t()).float()[None, None].expand(channel, 1, window_size, window_size).contiguous()


def ssim(x, y, window_size=11):
    """The metric as utils.py uses it: a Python float."""
    return ssim_tensor(x, y, window_size).item()


def ssim_tensor(x, y, window_size=11):
    """pytorch_ssim.ssim (:98-107) -> _ssim (:26-58): replicate padding, grouped 11x11 Gaussian, mean of the map
    (differentiable: the training loss uses pytorch_ssim.SSIM(), same arithmetic)."""
    c = x.shape[1]
    w = ssim_window(window_size, 1.5, c)
    pad = window_size // 2
    x, y = F.pad(x, (pad,) * 4, mode='replicate'), F.pad(y, (pad,) * 4, mode='replicate')
    mu1, mu2 = F.conv2d(x, w, groups=c), F.conv2d(y, w, groups=c)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(x * x, w, groups=c) - mu1_sq
    s2 = F.conv2d(y * y, w, groups=c) - mu2_sq
    s12 = F.conv2d(x * y, w, groups=c) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))).mean()


def calc_img_dists(x, y):
    """utils.py:420-491: (PSNR, RMSE, SSIM, mean L2 * 255, mean L_inf * 255, mean dE2000)."""
    x, y = expand_4d(x), expand_4d(y)
    mse = F.mse_loss(x, y)
    d = x - y
    de = ciede2000_diff(rgb2lab_diff(x), rgb2lab_diff(y)).mean().item()
    return (10 * math.log10(1 / mse), math.sqrt(mse.item() * 3), ssim(x, y), torch.norm(d, p=2, dim=1).mean().item() * 255,
            torch.norm(d, p=float('inf'), dim=1).mean().item() * 255, de)


# --------------------------------------------------------------------------------------
# train_network.py:235-363 (train_pcnet, one iteration) and :367-392 (compute_loss)
# --------------------------------------------------------------------------------------
def compute_loss(prj_infer, prj_train, loss_option):
    """train_network.py:367-392 ('huber' omitted: not used for PCNet)."""
    if loss_option == '':
        raise TypeError('Loss type not specified')
    train_loss = 0
    if 'l1' in loss_option:
        train_loss = train_loss + F.l1_loss(prj_infer, prj_train, reduction='mean')
    l2_loss = F.mse_loss(prj_infer, prj_train, reduction='mean')
    if 'l2' in loss_option:
        train_loss = train_loss + l2_loss
    if 'ssim' in loss_option:
        train_loss = train_loss + 1 * (1 - ssim_tensor(prj_infer, prj_train))
    return train_loss, l2_loss


class PCNetTrainOracle:
    """The reference's training iteration (train_network.py:247-265 optimisers / schedulers, :300-320 loop body) on a
    PCNet state_dict, with torch.autograd and torch.optim on the CPU."""

    def __init__(self, sd, cam_scene, batch_size, l2_reg=1e-4, lr_drop_ratio=0.2):
        self.buffers = {k: v.clone() for k, v in sd.items() if k in ('mask', 'warping_net.ctrl_pts')}
        self.p = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k not in self.buffers}
        aff = [self.p['warping_net.affine_mat'], self.p['warping_net.theta']]
        ref = [v for k, v in self.p.items() if 'warping_net.grid_refine_net' in k]
        shd = [v for k, v in self.p.items() if 'warping_net' not in k]
        self.opts = [torch.optim.Adam([{'params': aff}], lr=1e-2, weight_decay=0),
                     torch.optim.Adam([{'params': ref}], lr=5e-3, weight_decay=0),
                     torch.optim.Adam([{'params': shd}], lr=1e-3, weight_decay=l2_reg)]
        self.scheds = [torch.optim.lr_scheduler.MultiStepLR(o, milestones=[m], gamma=lr_drop_ratio)
                       for o, m in zip(self.opts, (100, 1200, 1800))]
        self.scene = expand_4d(cam_scene).expand(batch_size, -1, -1, -1)
        self.iters = 0

    def sd(self):
        d = dict(self.p)
        d.update(self.buffers)
        return d

    def step(self, prj, cam, loss=None):
        if loss is None:
            loss = 'l1' if self.iters <= 400 else 'l1+ssim'
        infer = pcnet_forward(self.sd(), prj, self.scene, per_batch_grid=True)
        train_loss, l2 = compute_loss(infer, cam, loss)
        for o in self.opts:
            o.zero_grad()
        train_loss.backward()
        self.grads = {k: v.grad.detach().clone() for k, v in self.p.items()}
        for o in self.opts:
            o.step()
        for s in self.scheds:
            s.step()
        self.iters += 1
        return float(train_loss.detach()), float(l2.detach())
