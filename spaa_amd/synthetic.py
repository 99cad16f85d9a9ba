"""Deterministic synthetic inputs for tests and bench.

Everything here is *data generation* on the host (numpy RNG -> torch CPU tensors):
random-init PCNet / classifier weights of the reference's architecture and
low-pass-filtered random scenes.  There is no network access, so no trained
checkpoints or datasets exist; bench.py says "synthetic" in its `data` field.

Reference shapes: PCNet state_dict of 44 parameters + 2 buffers
(/root/reference/src/python/models.py:98-140,214-265,305-327); ResNet-18 keys follow
the third-party torchvision naming the reference loads (classifier.py:25-36).
"""
import math

import numpy as np
import torch

# ImageNet-10 target ids used by the reference (data/imagenet10_clsidx_to_labels.txt)
IMAGENET10_TARGETS = [1, 7, 21, 207, 340, 745, 779, 846, 947, 950]


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _conv_w(rng, cout, cin, k, kaiming=True):
    fan_in = cin * k * k
    if kaiming:  # nn.init.kaiming_normal_ (models.py:261-265)
        w = rng.standard_normal((cout, cin, k, k)) * math.sqrt(2.0 / fan_in)
    else:
        bound = 1.0 / math.sqrt(fan_in)
        w = rng.uniform(-bound, bound, (cout, cin, k, k))
    b = rng.uniform(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in), (cout,))
    return _t(w), _t(b)


def _deconv_w(rng, cin, cout, k):
    # ConvTranspose2d keeps PyTorch's default init: weight [cin, cout, k, k], fan_in = cout*k*k
    fan_in = cout * k * k
    bound = 1.0 / math.sqrt(fan_in)
    w = rng.uniform(-bound, bound, (cin, cout, k, k))
    b = rng.uniform(-bound, bound, (cout,))
    return _t(w), _t(b)


def uniform_ctrl_pts(grid_shape=(6, 6)):
    """Control points on [0,1]^2, x fastest (pytorch_tps.py:201-217)."""
    h, w = grid_shape
    c = torch.zeros(h, w, 2)
    c[..., 0] = torch.linspace(0, 1, w)
    c[..., 1] = torch.linspace(0, 1, h).unsqueeze(-1)
    return c.view(-1, 2)


def pcnet_state_dict(seed=0, cam_sz=(256, 256), mask='ones', affine=(0.9, 0.02, 0.05, -0.03, 0.9, 0.05),
                     theta_scale=1e-3, refine_std=1e-4, scene_like=True):
    """Random-init PCNet weights: WarpingNet defaults perturbed by a small
    affine so the warp is non-trivial; ShadingNetSPAA kaiming-normal.

    `scene_like`: additionally make skipConv1 pass the scene through (identity taps + small noise) and
    shrink conv6, so that PCNet(x, s) ~ s + a projector-dependent modulation, as a trained PCNet
    behaves (un-lit scene plus projected light).  Without it the random net's output is unrelated to
    the scene and the d_thr branch of SPAA Algorithm 1 is never exercised."""
    rng = np.random.default_rng(seed)
    sd = {}
    h, w = cam_sz
    if mask == 'ones':
        m = torch.ones(1, 1, h, w)
    elif mask == 'rect':
        m = torch.zeros(1, 1, h, w)
        y0, x0 = h // 8, w // 8
        m[:, :, y0:h - y0, x0:w - x0] = 1
    else:
        m = mask.clone().float().view(1, 1, h, w)
    sd['mask'] = m
    sd['warping_net.affine_mat'] = _t(np.array(affine, dtype=np.float32).reshape(1, 2, 3))
    sd['warping_net.theta'] = _t(theta_scale * (1.0 + 0.5 * rng.standard_normal((1, 38, 2))))
    sd['warping_net.ctrl_pts'] = uniform_ctrl_pts()
    for idx, (cout, cin, k) in zip((0, 2), ((32, 2, 3), (64, 32, 3))):
        wgt = _t(rng.standard_normal((cout, cin, k, k)) * refine_std)
        fan_in = cin * k * k
        sd[f'warping_net.grid_refine_net.{idx}.weight'] = wgt
        sd[f'warping_net.grid_refine_net.{idx}.bias'] = _t(
            rng.uniform(-1, 1, (cout,)) / math.sqrt(fan_in) * 1e-2)
    for idx, (cin, cout, k) in zip((4, 6), ((64, 32, 2), (32, 2, 2))):
        wgt, b = _deconv_w(rng, cin, cout, k)
        sd[f'warping_net.grid_refine_net.{idx}.weight'] = wgt * 0.05
        sd[f'warping_net.grid_refine_net.{idx}.bias'] = b * 0.05
    sn = 'shading_net.'
    for name, (cout, cin, k) in {
        'conv1': (32, 3, 3), 'conv2': (64, 32, 3), 'conv3': (128, 64, 3), 'conv4': (256, 128, 3),
        'conv5': (128, 256, 3), 'conv1_s': (32, 6, 3), 'conv2_s': (64, 32, 3), 'conv3_s': (128, 64, 3),
        'conv4_s': (256, 128, 3), 'conv6': (3, 32, 3), 'skipConv1.0': (3, 3, 1), 'skipConv1.2': (3, 3, 3),
        'skipConv1.4': (3, 3, 3), 'skipConv2': (64, 32, 1), 'skipConv3': (128, 64, 3),
    }.items():
        wgt, b = _conv_w(rng, cout, cin, k)
        sd[sn + name + '.weight'], sd[sn + name + '.bias'] = wgt, b
    for name, (cin, cout, k) in {'transConv1': (128, 64, 3), 'transConv2': (64, 32, 2)}.items():
        wgt, b = _deconv_w(rng, cin, cout, k)
        sd[sn + name + '.weight'], sd[sn + name + '.bias'] = wgt, b
    if scene_like:
        eye = torch.eye(3)
        w0 = sd[sn + 'skipConv1.0.weight'] * 0.05
        w0[:, :, 0, 0] += eye
        sd[sn + 'skipConv1.0.weight'] = w0
        for nm in ('skipConv1.2', 'skipConv1.4'):
            wk = sd[sn + nm + '.weight'] * 0.02
            wk[:, :, 1, 1] += eye
            sd[sn + nm + '.weight'] = wk
        for nm in ('skipConv1.0', 'skipConv1.2', 'skipConv1.4'):
            sd[sn + nm + '.bias'] = sd[sn + nm + '.bias'] * 0.02
        sd[sn + 'conv6.weight'] = sd[sn + 'conv6.weight'] * 0.25
        sd[sn + 'conv6.bias'] = sd[sn + 'conv6.bias'] * 0.1 - 0.02
    return sd


def compennet_pp_state_dict(seed=5, out_size=(256, 256)):
    """Random-init CompenNet++ (warping_net.* as in PCNet, compen_net.* per models.py:11-59)."""
    rng = np.random.default_rng(seed)
    base = pcnet_state_dict(seed, cam_sz=out_size)
    sd = {k: v for k, v in base.items() if k.startswith('warping_net.')}
    cn = 'compen_net.'
    for name, (cout, cin, k) in {
        'conv1': (32, 3, 3), 'conv2': (64, 32, 3), 'conv3': (128, 64, 3), 'conv4': (256, 128, 3), 'conv5': (128, 256, 3),
        'conv1_s': (32, 3, 3), 'conv2_s': (64, 32, 3), 'conv3_s': (128, 64, 3), 'conv4_s': (256, 128, 3),
        'conv6': (3, 32, 3), 'skipConv1.0': (3, 3, 3), 'skipConv1.2': (3, 3, 3), 'skipConv1.4': (3, 3, 3),
        'skipConv2': (64, 32, 1), 'skipConv3': (128, 64, 1),
    }.items():
        wgt, b = _conv_w(rng, cout, cin, k)
        sd[cn + name + '.weight'], sd[cn + name + '.bias'] = wgt, b
    for name, (cin, cout, k) in {'transConv1': (128, 64, 2), 'transConv2': (64, 32, 2)}.items():
        wgt, b = _deconv_w(rng, cin, cout, k)
        sd[cn + name + '.weight'], sd[cn + name + '.bias'] = wgt, b
    return sd


def _bn(rng, c, sd, prefix):
    sd[prefix + '.weight'] = _t(rng.uniform(0.5, 1.5, (c,)))
    sd[prefix + '.bias'] = _t(rng.uniform(-0.2, 0.2, (c,)))
    sd[prefix + '.running_mean'] = _t(rng.uniform(-0.2, 0.2, (c,)))
    sd[prefix + '.running_var'] = _t(rng.uniform(0.5, 1.5, (c,)))


def resnet18_state_dict(seed=2, num_classes=1000, logit_gain=1.0):
    """Random-init ResNet-18 (torchvision key names), eval-mode BN with non-trivial statistics."""
    rng = np.random.default_rng(seed)
    sd = {}

    def conv(name, cout, cin, k):
        fan_out = cout * k * k  # torchvision: kaiming_normal_(mode='fan_out')
        sd[name + '.weight'] = _t(rng.standard_normal((cout, cin, k, k)) * math.sqrt(2.0 / fan_out))

    conv('conv1', 64, 3, 7)
    _bn(rng, 64, sd, 'bn1')
    cin = 64
    for li, cout in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f'layer{li}.{bi}'
            stride = 2 if (li > 1 and bi == 0) else 1
            conv(p + '.conv1', cout, cin, 3)
            _bn(rng, cout, sd, p + '.bn1')
            conv(p + '.conv2', cout, cout, 3)
            _bn(rng, cout, sd, p + '.bn2')
            if stride != 1 or cin != cout:
                conv(p + '.downsample.0', cout, cin, 1)
                _bn(rng, cout, sd, p + '.downsample.1')
            cin = cout
    bound = 1.0 / math.sqrt(512)
    sd['fc.weight'] = _t(rng.uniform(-bound, bound, (num_classes, 512)) * logit_gain)
    sd['fc.bias'] = _t(rng.uniform(-bound, bound, (num_classes,)) * logit_gain)
    return sd


def vgg16_state_dict(seed=3, num_classes=1000, logit_gain=1.0, fc_width=4096):
    """Random-init VGG-16 (torchvision key names: features.N / classifier.N).  `fc_width` < 4096 gives a small
    test-sized head (the architecture code reads the widths from the tensors)."""
    rng = np.random.default_rng(seed)
    sd = {}
    cin, idx = 3, 0
    for v in [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']:
        if v == 'M':
            idx += 1
            continue
        fan_out = v * 9  # torchvision: kaiming_normal_(mode='fan_out', nonlinearity='relu'), bias 0
        sd[f'features.{idx}.weight'] = _t(rng.standard_normal((v, cin, 3, 3)) * math.sqrt(2.0 / fan_out))
        sd[f'features.{idx}.bias'] = _t(rng.uniform(-0.05, 0.05, (v,)))
        cin = v
        idx += 2
    dims = [(fc_width, 512 * 49), (fc_width, fc_width), (num_classes, fc_width)]
    for i, (o, n) in zip((0, 3, 6), dims):
        g = logit_gain if i == 6 else 1.0
        sd[f'classifier.{i}.weight'] = _t(rng.standard_normal((o, n)) * math.sqrt(2.0 / n) * g)
        sd[f'classifier.{i}.bias'] = _t(rng.uniform(-0.05, 0.05, (o,)) * g)
    return sd


def inception_v3_state_dict(seed=4, num_classes=1000, logit_gain=1.0):
    """Random-init Inception-v3 (torchvision key names; AuxLogits omitted: unused in eval)."""
    rng = np.random.default_rng(seed)
    sd = {}

    def bc(name, cin, cout, k):
        kh, kw = k
        sd[name + '.conv.weight'] = _t(rng.standard_normal((cout, cin, kh, kw)) * math.sqrt(2.0 / (cin * kh * kw)))
        _bn(rng, cout, sd, name + '.bn')
        return cout

    c = bc('Conv2d_1a_3x3', 3, 32, (3, 3))
    c = bc('Conv2d_2a_3x3', c, 32, (3, 3))
    c = bc('Conv2d_2b_3x3', c, 64, (3, 3))
    c = bc('Conv2d_3b_1x1', c, 80, (1, 1))
    c = bc('Conv2d_4a_3x3', c, 192, (3, 3))

    def inc_a(n, cin, pf):
        bc(n + '.branch1x1', cin, 64, (1, 1))
        bc(n + '.branch5x5_1', cin, 48, (1, 1)); bc(n + '.branch5x5_2', 48, 64, (5, 5))
        bc(n + '.branch3x3dbl_1', cin, 64, (1, 1)); bc(n + '.branch3x3dbl_2', 64, 96, (3, 3))
        bc(n + '.branch3x3dbl_3', 96, 96, (3, 3))
        bc(n + '.branch_pool', cin, pf, (1, 1))
        return 64 + 64 + 96 + pf

    def inc_b(n, cin):
        bc(n + '.branch3x3', cin, 384, (3, 3))
        bc(n + '.branch3x3dbl_1', cin, 64, (1, 1)); bc(n + '.branch3x3dbl_2', 64, 96, (3, 3))
        bc(n + '.branch3x3dbl_3', 96, 96, (3, 3))
        return 384 + 96 + cin

    def inc_c(n, cin, c7):
        bc(n + '.branch1x1', cin, 192, (1, 1))
        bc(n + '.branch7x7_1', cin, c7, (1, 1)); bc(n + '.branch7x7_2', c7, c7, (1, 7)); bc(n + '.branch7x7_3', c7, 192, (7, 1))
        bc(n + '.branch7x7dbl_1', cin, c7, (1, 1)); bc(n + '.branch7x7dbl_2', c7, c7, (7, 1))
        bc(n + '.branch7x7dbl_3', c7, c7, (1, 7)); bc(n + '.branch7x7dbl_4', c7, c7, (7, 1))
        bc(n + '.branch7x7dbl_5', c7, 192, (1, 7))
        bc(n + '.branch_pool', cin, 192, (1, 1))
        return 768

    def inc_d(n, cin):
        bc(n + '.branch3x3_1', cin, 192, (1, 1)); bc(n + '.branch3x3_2', 192, 320, (3, 3))
        bc(n + '.branch7x7x3_1', cin, 192, (1, 1)); bc(n + '.branch7x7x3_2', 192, 192, (1, 7))
        bc(n + '.branch7x7x3_3', 192, 192, (7, 1)); bc(n + '.branch7x7x3_4', 192, 192, (3, 3))
        return 320 + 192 + cin

    def inc_e(n, cin):
        bc(n + '.branch1x1', cin, 320, (1, 1))
        bc(n + '.branch3x3_1', cin, 384, (1, 1)); bc(n + '.branch3x3_2a', 384, 384, (1, 3)); bc(n + '.branch3x3_2b', 384, 384, (3, 1))
        bc(n + '.branch3x3dbl_1', cin, 448, (1, 1)); bc(n + '.branch3x3dbl_2', 448, 384, (3, 3))
        bc(n + '.branch3x3dbl_3a', 384, 384, (1, 3)); bc(n + '.branch3x3dbl_3b', 384, 384, (3, 1))
        bc(n + '.branch_pool', cin, 192, (1, 1))
        return 2048

    c = inc_a('Mixed_5b', c, 32)
    c = inc_a('Mixed_5c', c, 64)
    c = inc_a('Mixed_5d', c, 64)
    c = inc_b('Mixed_6a', c)
    for n, c7 in (('Mixed_6b', 128), ('Mixed_6c', 160), ('Mixed_6d', 160), ('Mixed_6e', 192)):
        c = inc_c(n, c, c7)
    c = inc_d('Mixed_7a', c)
    c = inc_e('Mixed_7b', c)
    c = inc_e('Mixed_7c', c)
    bound = 1.0 / math.sqrt(2048)
    sd['fc.weight'] = _t(rng.uniform(-bound, bound, (num_classes, 2048)) * logit_gain)
    sd['fc.bias'] = _t(rng.uniform(-bound, bound, (num_classes,)) * logit_gain)
    return sd


def scenes(seed=1, n=1, sz=(256, 256), box=8, lo=0.05, hi=0.95):
    """`n` smooth random camera scenes in [lo,hi]: U[0,1) low-pass filtered by a box x box mean."""
    rng = np.random.default_rng(seed)
    h, w = sz
    raw = torch.from_numpy(rng.random((n, 3, h + box - 1, w + box - 1)).astype(np.float32))
    sm = torch.nn.functional.avg_pool2d(raw, box, stride=1)
    # stretch the contrast back (the box filter shrinks it), then map to [lo, hi]
    mn = sm.amin(dim=(1, 2, 3), keepdim=True)
    mx = sm.amax(dim=(1, 2, 3), keepdim=True)
    sm = (sm - mn) / (mx - mn)
    return (lo + (hi - lo) * sm).contiguous()
