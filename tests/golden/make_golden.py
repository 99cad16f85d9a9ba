"""Generates tests/golden/*.npz by running the UNMODIFIED reference (imported via oracle/ref_shims.py)
on deterministic synthetic inputs.  Runs only in the build container (needs /root/reference);
the fixtures it writes are data (inputs' seeds + expected outputs), not reference source.

    python tests/golden/make_golden.py [--only NAME ...]

While generating, every case is also run through the oracle restatement (oracle/spaa_oracle.py) and the
max |difference| is printed and stored (`oracle_maxdiff`), which is how the oracle was pinned.
"""
import argparse
import types
import io
import contextlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import spaa_oracle as so  # noqa: E402
import ref_shims  # noqa: E402
from spaa_amd import synthetic as syn  # noqa: E402

LABELS = {i: f'class{i}' for i in range(1000)}
GAIN = 20.0


def weights_checksum(sd):
    return np.array([float(sum(v.double().sum() for v in sd.values())),
                     float(sum(v.double().abs().sum() for v in sd.values()))])


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f'  wrote {name}.npz ({os.path.getsize(path) / 1e6:.2f} MB)')


def color_inputs():
    """Random pairs plus the edge cases of SURVEY §8c(1)."""
    rng = np.random.default_rng(7)
    a = rng.random((2, 3, 16, 16)).astype(np.float32)
    b = rng.random((2, 3, 16, 16)).astype(np.float32)
    a[0, :, 0, 0:4] = 0.0                      # exactly black (X=Y=Z=0 -> f(0)=0, Q3)
    b[0, :, 0, 2:6] = 0.0
    a[0, :, 1, :] = b[0, :, 1, :]              # identical pixels -> dE = 0, zero gradient
    g = np.linspace(0, 1, 16, dtype=np.float32)
    a[0, :, 2, :] = g                          # greys: a=b=0 in Lab up to rounding
    b[0, :, 2, :] = g[::-1]
    a[0, :, 3, :] = np.array([0.0404, 0.0405, 0.0406, 0.04049999, 0.0405001, 0.03, 0.05, 0.0] * 2, np.float32)
    b[0, :, 3, :] = 0.3
    a[0, 0, 4, :], a[0, 1, 4, :], a[0, 2, 4, :] = 1.0, g * 0.2, 0.0      # saturated reds: hue near 0/360 wrap
    b[0, 0, 4, :], b[0, 1, 4, :], b[0, 2, 4, :] = 1.0, 0.0, g * 0.2
    a[0, 0, 5, :], a[0, 1, 5, :], a[0, 2, 5, :] = g, 0.0, 1.0            # blue/purple: hue ~ 275 (dRO peak)
    b[0, 0, 5, :], b[0, 1, 5, :], b[0, 2, 5, :] = g[::-1], 0.1, 0.9
    a[0, :, 6, :] = 0.002 * g                  # very dark: xyz below 0.008856 (linear branch)
    b[0, :, 6, :] = 0.003 * g[::-1]
    a[0, :, 7, :] = 1.0                        # white vs near white
    b[0, :, 7, :] = 1.0 - 0.01 * g
    return torch.from_numpy(a), torch.from_numpy(b)


def gen_color(ref):
    a, b = color_inputs()
    a.requires_grad_(True)
    lab_a = ref.color.rgb2lab_diff(a, 'cpu')
    lab_b = ref.color.rgb2lab_diff(b, 'cpu')
    de = ref.color.ciede2000_diff(lab_a, lab_b, 'cpu')
    de.sum().backward()
    ga = a.grad.clone()
    a2 = a.detach().clone().requires_grad_(True)
    de_o = so.ciede2000_diff(so.rgb2lab_diff(a2), so.rgb2lab_diff(b))
    de_o.sum().backward()
    diff = max((de_o - de).abs().max().item(), (a2.grad - ga).abs().max().item())
    # Sharma pair: reference constant 39 gives 2.0213, textbook 2.0425
    l1 = torch.tensor([50., 2.6772, -79.7751]).view(1, 3, 1, 1)
    l2 = torch.tensor([50., 0., -82.7485]).view(1, 3, 1, 1)
    sharma = ref.color.ciede2000_diff(l1, l2, 'cpu')
    print(f'  colour: oracle maxdiff {diff:.3e}; Sharma pair -> {sharma.item():.4f}')
    save('color_kat', rgb_a=a.detach(), rgb_b=b, lab_a=lab_a.detach(), lab_b=lab_b, de=de.detach(), grad_a=ga,
         sharma=sharma, oracle_maxdiff=diff)


def gen_pcnet(ref, name, prj_sz, cam_sz, mask, seed, bsz=2, use_rough=True):
    sd = syn.pcnet_state_dict(seed, cam_sz=cam_sz, mask=mask)
    if use_rough:
        pc = ref_shims.make_reference_pcnet(ref, sd, prj_sz, cam_sz)
    else:   # models.py:344-345: shading_net(x, s); ShadingNetSPAA(use_rough=False) has a 3-channel conv1_s
        sd['shading_net.conv1_s.weight'] = sd['shading_net.conv1_s.weight'][:, :3].contiguous()
        holder = types.SimpleNamespace
        pc = ref.models.PCNet(sd['mask'], holder(module=ref.models.WarpingNet(out_size=tuple(cam_sz))),
                              holder(module=ref.models.ShadingNetSPAA(use_rough=False)), fix_shading_net=True, use_rough=False)
        pc.load_state_dict(sd)
        pc.eval()
    rng = np.random.default_rng(seed + 100)
    x = torch.from_numpy(rng.random((bsz, 3, *prj_sz)).astype(np.float32)).requires_grad_(True)
    s = syn.scenes(seed + 1, bsz, cam_sz)
    r = torch.from_numpy(rng.standard_normal((bsz, 3, *cam_sz)).astype(np.float32))
    y = pc(x, s)
    (y * r).sum().backward()
    g = x.grad.clone()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = so.pcnet_forward(sd, x2, s, use_rough=use_rough)
    (y2 * r).sum().backward()
    diff = max((y2 - y).abs().max().item(), (x2.grad - g).abs().max().item())
    fine = so.warping_fine_grid(sd, x.shape, cam_sz)
    print(f'  {name}: oracle maxdiff {diff:.3e}')
    save(name, seed=seed, prj_sz=prj_sz, cam_sz=cam_sz, mask=mask, x=x.detach(), s=s, r=r, y=y.detach(), grad_x=g,
         fine_grid=fine, wsum=weights_checksum(sd), oracle_maxdiff=diff, use_rough=use_rough)


def gen_compennet_pp(ref, name, sz, seed):
    sd = syn.compennet_pp_state_dict(seed, out_size=sz)
    holder = types.SimpleNamespace
    net = ref.models.CompenNetPlusplus(holder(module=ref.models.WarpingNet(out_size=tuple(sz))),
                                       holder(module=ref.models.CompenNet()))
    net.load_state_dict(sd)
    net.eval()
    x = syn.scenes(seed + 20, 2, sz)
    s = syn.scenes(seed + 21, 1, sz).expand(2, -1, -1, -1).contiguous()
    with torch.no_grad():
        y = net(x, s)
        y2 = so.compennet_pp_forward(sd, x, s, sz)
    diff = (y2 - y).abs().max().item()
    print(f'  {name}: oracle maxdiff {diff:.3e}')
    save(name, seed=seed, sz=sz, x=x, s=s, y=y, wsum=weights_checksum(sd), oracle_maxdiff=diff)


def gen_metrics(ref, name):
    """calc_img_dists (utils.py:420-491).  utils.py itself cannot be imported (visdom / Qt at import time), so its four
    small metric functions are exec'd from their source text; SSIM and deltaE are the reference's own modules."""
    import ast
    import importlib.util
    import math
    import torch.nn as nn
    ref_root = ref_shims.REF_ROOT
    spec = importlib.util.spec_from_file_location('ref_pytorch_ssim', os.path.join(ref_root, 'pytorch_ssim', '__init__.py'))
    ref_ssim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_ssim)
    src = open(os.path.join(ref_root, 'utils.py')).read()
    ns = dict(torch=torch, nn=nn, math=math)
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ('psnr', 'rmse', 'l2_norm', 'linf_norm'):
            exec(compile(ast.Module([node], []), 'utils.py', 'exec'), ns)
    x = syn.scenes(31, 3, (48, 72))
    rng = np.random.default_rng(5)
    y = (x + torch.from_numpy(rng.normal(0, 0.03, x.shape).astype(np.float32))).clamp(0, 1)
    y[0, :, :8] = x[0, :, :8]  # identical region (dE shortcut, SSIM = 1 there)
    with torch.no_grad():
        want = np.array([ns['psnr'](x, y), ns['rmse'](x, y), ref_ssim.ssim(x, y).item(), ns['l2_norm'](x, y),
                         ns['linf_norm'](x, y), ref.color.deltaE(x, y)])
    got = np.array(so.calc_img_dists(x, y))
    diff = float(np.abs(got - want).max())
    print(f'  {name}: reference {want}, oracle maxdiff {diff:.3e}')
    save(name, x=x, y=y, dists=want, oracle_maxdiff=diff)


class Recorder:
    """Wraps the duck-typed classifier to observe the reference's loop without touching it."""

    def __init__(self, clf):
        self.clf, self.top1, self.p1, self.cam_mean = clf, [], [], []

    def __call__(self, im, cp):
        raw, p, idx = self.clf(im, cp)
        self.top1.append(idx[:, 0].copy())
        self.p1.append(p[:, 0].copy())
        self.cam_mean.append(im.detach().mean(dim=(1, 2, 3)).numpy().copy())
        return raw, p, idx


def near_targets(clf, scene, cp, n, skip=1):
    _, _, idx = clf(scene, cp)
    return [int(i) for i in idx[0, skip:skip + n]]


def gen_spaa(ref, name, sz, targeted, targets, d_thr, stealth, seed=0, mask='rect', keep=None, scene_seed=1):
    t0 = time.time()
    sd = syn.pcnet_state_dict(seed, cam_sz=sz, mask=mask)
    pc = ref_shims.make_reference_pcnet(ref, sd, sz, sz)
    csd = syn.resnet18_state_dict(2, logit_gain=GAIN)
    in_sz = (224, 224) if sz[0] >= 240 else (sz[0] - 8, sz[1] - 8)
    cp = (240, 240) if sz[0] >= 240 else (sz[0] - 4, sz[1] - 4)
    clf = so.OracleClassifier('resnet18', csd, input_sz=in_sz)
    scene = syn.scenes(scene_seed, 1, sz)
    if targets == 'true':
        targets = near_targets(clf, scene, cp, 1, skip=0)
    elif isinstance(targets, tuple) and targets[0] == 'near':
        targets = near_targets(clf, scene, cp, targets[1])
    setup = dict(classifier_crop_sz=cp, prj_brightness=0.5, prj_im_sz=sz)
    rec = Recorder(clf)
    with contextlib.redirect_stdout(io.StringIO()):
        cam_best, prj_best = ref.attack.spaa(pc, rec, LABELS, targets, targeted, scene[0], d_thr, stealth, 'cpu', setup)
    tr = []
    cam_o, prj_o = so.spaa(sd, clf, targets, targeted, scene, d_thr, stealth, setup, trace=tr)
    diff = max((cam_o - cam_best).abs().max().item(), (prj_o - prj_best).abs().max().item())
    top1 = np.stack(rec.top1)
    assert (top1 == np.stack([t['top1'] for t in tr])).all()
    k = slice(None) if keep is None else slice(0, keep)
    print(f'  {name}: B={len(targets)} oracle maxdiff {diff:.3e}; succ/it {[int(t["succ"].sum()) for t in tr][::7]} '
          f'best_adv/it {[int(t["best_adv"].sum()) for t in tr][::7]} ({time.time() - t0:.0f}s)')
    save(name, seed=seed, scene_seed=scene_seed, sz=sz, mask=mask, targeted=targeted, targets=np.array(targets),
         d_thr=d_thr, stealth=stealth, gain=GAIN, crop=cp, input_sz=in_sz,
         cam_infer_best=cam_best.detach()[k], prj_adv_best=prj_best.detach()[k],
         top1=top1, p1=np.stack(rec.p1), cam_mean=np.stack(rec.cam_mean),
         succ=np.stack([t['succ'] for t in tr]), best_adv=np.stack([t['best_adv'] for t in tr]),
         best=np.stack([t['best'] for t in tr]), caml2=np.stack([t['caml2'] for t in tr]),
         camdE=np.stack([t['camdE'] for t in tr]), col_loss=np.stack([t['col_loss'] for t in tr]),
         prj_adv_it0=tr[0]['prj_adv'][k], prj_adv_it9=tr[9]['prj_adv'][k],
         wsum=weights_checksum(sd), oracle_maxdiff=diff)


def gen_sensitivity(ref, name, sz=(64, 64)):
    """How far does the UNMODIFIED reference move its own 50-iteration output under changes that are pure rounding?  Same
    case as spaa_64_near (8 targets, camdE_caml2, d_thr 5); three runs of `spaa()`: (a) 8 CPU threads, (b) 1 thread (another
    summation order inside the convolutions), (c) 8 threads with the start image one ulp above 0.5.  The fixture is what
    DESIGN.md section 4 and the docstrings of tests/test_gpu_parity.py cite; tests/test_oracle_golden.py re-measures the
    thread-count part with the oracle (== reference bit for bit) on the machine that runs the tests."""
    t0 = time.time()
    sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='rect')
    csd = syn.resnet18_state_dict(2, logit_gain=GAIN)
    in_sz, cp = (sz[0] - 8, sz[1] - 8), (sz[0] - 4, sz[1] - 4)
    clf = so.OracleClassifier('resnet18', csd, input_sz=in_sz)
    scene = syn.scenes(1, 1, sz)
    targets = near_targets(clf, scene, cp, 8)
    nthr = torch.get_num_threads()

    def run(threads, brightness):
        torch.set_num_threads(threads)
        pc = ref_shims.make_reference_pcnet(ref, sd, sz, sz)
        setup = dict(classifier_crop_sz=cp, prj_brightness=brightness, prj_im_sz=sz)
        rec = Recorder(clf)
        with contextlib.redirect_stdout(io.StringIO()):
            cam, prj = ref.attack.spaa(pc, rec, LABELS, targets, True, scene[0], 5, 'camdE_caml2', 'cpu', setup)
        torch.set_num_threads(nthr)
        return cam.detach(), prj.detach(), np.stack(rec.top1)

    ulp = float(np.nextafter(np.float32(0.5), np.float32(1.0)))
    cam8, prj8, top8 = run(8, 0.5)
    cam1, prj1, top1 = run(1, 0.5)
    camu, prju, topu = run(8, ulp)

    def rel(a, b):
        return float((a - b).abs().max() / b.abs().max())

    # growth of a 1e-7 perturbation of the start, per iteration (oracle == reference; its trace has every iterate)
    tra, trb = [], []
    setup = dict(classifier_crop_sz=cp, prj_brightness=0.5, prj_im_sz=sz)
    so.spaa(sd, clf, targets, True, scene, 5, 'camdE_caml2', setup, iters=8, trace=tra)
    so.spaa(sd, clf, targets, True, scene, 5, 'camdE_caml2', dict(setup, prj_brightness=0.5 + 1e-7), iters=8, trace=trb)
    growth = np.array([np.abs(a['prj_adv'] - b['prj_adv']).max() / np.abs(a['prj_adv']).max() for a, b in zip(tra, trb)])
    first_div = lambda ta, tb: int(np.argmax((ta != tb).any(axis=1))) if (ta != tb).any() else -1
    print(f'  {name}: prj_adv_best rel Linf  8 vs 1 threads {rel(prj1, prj8):.3f}, one-ulp start {rel(prju, prj8):.3f}; cam_infer_best '
          f'{rel(cam1, cam8):.3f} / {rel(camu, cam8):.3f}; first iteration with another top-1: {first_div(top1, top8)} / '
          f'{first_div(topu, top8)}; growth of 1e-7: {growth} ({time.time() - t0:.0f}s)')
    save(name, sz=sz, targets=np.array(targets), ulp_start=ulp, prj_threads=rel(prj1, prj8), prj_ulp=rel(prju, prj8),
         cam_threads=rel(cam1, cam8), cam_ulp=rel(camu, cam8), top1_div_threads=first_div(top1, top8),
         top1_div_ulp=first_div(topu, top8), growth_1e7=growth,
         mean_prj_threads=float((prj1 - prj8).abs().mean()), mean_prj_ulp=float((prju - prj8).abs().mean()))


def gen_percal(ref, name, sz, targeted, d_thr, confidence):
    csd = syn.resnet18_state_dict(2, logit_gain=GAIN)
    cp, in_sz = (sz[0] - 4, sz[1] - 4), (sz[0] - 8, sz[1] - 8)
    clf = so.OracleClassifier('resnet18', csd, input_sz=in_sz)
    scene = syn.scenes(1, 1, sz)
    targets = near_targets(clf, scene, cp, 8) if targeted else near_targets(clf, scene, cp, 1, skip=0) * 8
    inputs = scene.expand(8, -1, -1, -1).contiguous()
    labels = torch.tensor(targets)
    att = ref.perc_al.PerC_AL(device='cpu', max_iterations=50, alpha_l_init=1, alpha_c_init=0.5, confidence=confidence)
    with contextlib.redirect_stdout(io.StringIO()):
        out = att.adversary_projector(clf, inputs, labels, LABELS, d_thr, targeted, cp)
    out_o = so.perc_al_adversary_projector(clf, inputs, labels, d_thr, targeted, cp, 50, 1., 0.5, confidence)
    diff = (out_o - out).abs().max().item()
    print(f'  {name}: oracle maxdiff {diff:.3e}; changed px {(out != inputs).float().mean().item():.3f}')
    save(name, sz=sz, targeted=targeted, targets=np.array(targets), d_thr=d_thr, confidence=confidence, gain=GAIN,
         crop=cp, input_sz=in_sz, x_adv_best=out.detach(), oracle_maxdiff=diff)


def _exec_defs(path, names, ns):
    """exec single function definitions out of a reference module that cannot be imported as a whole (missing cv2 /
    torchvision at import time); nothing of the source text is kept."""
    import ast
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), os.path.basename(path), 'exec'), ns)
    return ns


def _reference_classifier(body, input_sz):
    """The reference's own `Classifier.classify` / `__call__` (classifier.py:55-75) and `img_proc.expand_4d / resize /
    center_crop` (img_proc.py:110-132), exec'd from their source, around a network body we supply (torchvision and the
    pretrained weights are absent): pins the wrapper contract and the preprocessing, not the body."""
    import ast
    import torch.nn.functional as F
    ns = dict(torch=torch, F=F)
    _exec_defs(os.path.join(ref_shims.REF_ROOT, 'img_proc.py'), ('expand_4d', 'resize', 'center_crop'), ns)
    ns['cc'] = ns['center_crop']
    src = open(os.path.join(ref_shims.REF_ROOT, 'classifier.py')).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == 'Classifier':
            node.body = [n for n in node.body if isinstance(n, ast.FunctionDef) and n.name in ('classify', '__call__')]
            exec(compile(ast.Module([node], []), 'classifier.py', 'exec'), ns)
    clf = object.__new__(ns['Classifier'])
    mean = torch.tensor((0.485, 0.456, 0.406)).view(-1, 1, 1)
    std = torch.tensor((0.229, 0.224, 0.225)).view(-1, 1, 1)

    def normalize(t):  # torchvision.transforms.functional.normalize: tensor.clone().sub_(mean).div_(std)
        return t.clone().sub_(mean).div_(std)

    seen = []

    def model(x):
        seen.append(x)
        return body(x)

    clf.normalize = lambda x: torch.stack([normalize(x[i]) for i in range(x.shape[0])], 0)  # classifier.py:51
    clf.model, clf.device, clf.sort_results, clf.input_sz = model, 'cpu', True, tuple(input_sz)
    return clf, seen, ns


def gen_preproc(name, im_hw, crop, input_sz, seed, bsz=1):
    """a6/a7: Classifier.classify + img_proc helpers of the reference on seeded inputs; expected = the tensor the
    reference hands to the network, its gradient, and the (raw_score, p, idx) triple for the oracle's ResNet-18 body."""
    csd = syn.resnet18_state_dict(2, logit_gain=GAIN)
    clf, seen, ns = _reference_classifier(lambda x: so.resnet18_forward(csd, x), input_sz)
    rng = np.random.default_rng(seed)
    im = torch.from_numpy(rng.random((bsz, 3, *im_hw)).astype(np.float32)).requires_grad_(True)
    r = torch.from_numpy(rng.standard_normal((bsz, 3, *input_sz)).astype(np.float32))
    raw, p, idx = clf(im, crop)
    pre = seen[-1]
    (pre * r).sum().backward()
    g = im.grad.clone()
    # 3-D and uint8 inputs take the same route (classifier.py:56-59, img_proc.py:110-114)
    u8 = (im.detach()[0] * 255).to(torch.uint8)
    raw_u8, _, idx_u8 = clf(u8, crop)
    # oracle restatement on the same inputs
    im2 = im.detach().clone().requires_grad_(True)
    pre_o = so.classifier_preprocess(im2, crop, input_sz)
    (pre_o * r).sum().backward()
    raw_o, p_o, idx_o = so.OracleClassifier('resnet18', csd, input_sz=input_sz)(im.detach(), crop)
    diff = max((pre_o - pre).abs().max().item(), (im2.grad - g).abs().max().item(), (raw_o - raw).abs().max().item(),
               float(np.abs(p_o - p).max()))
    assert (idx_o == idx).all()
    print(f'  {name}: oracle maxdiff {diff:.3e}; crop origin via reference center_crop; up-sampling: {input_sz[0] > crop[0]}')
    save(name, seed=seed, im_hw=im_hw, crop=crop, input_sz=input_sz, bsz=bsz, pre=pre.detach(), grad_im=g,
         raw_score=raw.detach(), p5=p[:, :5], idx5=idx[:, :5], raw_score_u8=raw_u8.detach(), idx5_u8=idx_u8[:, :5],
         gain=GAIN, oracle_maxdiff=diff)


def gen_io(name):
    """f3: the reference's sample image (a data file) decoded independently of any image library; the file itself is
    committed next to the fixture so that spaa_amd.io.torch_imread can be checked on a reference-held PNG."""
    import hashlib
    import shutil
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from png_ref import decode_png
    src = os.path.join(os.path.dirname(os.path.dirname(ref_shims.REF_ROOT)), 'data', 'sample', 'anemone_fish.png')
    dst = os.path.join(HERE, 'anemone_fish.png')
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)
    rgb = decode_png(dst)
    save(name, shape=rgb.shape, sha256=hashlib.sha256(rgb.tobytes()).hexdigest(), mean_rgb=rgb.reshape(-1, 3).mean(0),
         corner=rgb[:4, :4], center=rgb[126:130, 126:130])
    print(f'  {name}: {rgb.shape} mean {rgb.reshape(-1, 3).mean(0)}')


def gen_train(ref, name, sz, bsz, seed):
    """8f-4: the reference's PCNet training iteration.  train_network.py cannot be imported (visdom / Qt at import), so
    `compute_loss` (:367-392) is exec'd from its source with the reference's own pytorch_ssim.SSIM() as `ssim_fun`, and the
    loop body (:306-320) runs on the reference's PCNet module with the optimisers / schedulers of :252-265."""
    import importlib.util
    import torch.nn.functional as F
    spec = importlib.util.spec_from_file_location('ref_pytorch_ssim', os.path.join(ref_shims.REF_ROOT, 'pytorch_ssim', '__init__.py'))
    ref_ssim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_ssim)
    ns = _exec_defs(os.path.join(ref_shims.REF_ROOT, 'train_network.py'), ('compute_loss',), dict(F=F, ssim_fun=ref_ssim.SSIM()))
    sd = syn.pcnet_state_dict(seed, cam_sz=sz, mask='rect')
    wn = ref.models.WarpingNet(out_size=tuple(sz))
    sn = ref.models.ShadingNetSPAA()
    holder = types.SimpleNamespace
    model = ref.models.PCNet(sd['mask'], holder(module=wn), holder(module=sn))
    model.load_state_dict(sd)
    named = [('module.' + k, v) for k, v in model.named_parameters()]       # (the reference wraps the model in DataParallel)
    aff = [v for k, v in named if k in ['module.warping_net.affine_mat', 'module.warping_net.theta']]
    refine = [v for k, v in named if 'module.warping_net.grid_refine_net' in k]
    shading = [v for k, v in named if 'module.warping_net' not in k]
    opts = [torch.optim.Adam([{'params': aff}], lr=1e-2, weight_decay=0), torch.optim.Adam([{'params': refine}], lr=5e-3, weight_decay=0),
            torch.optim.Adam([{'params': shading}], lr=1e-3, weight_decay=1e-4)]
    scene = syn.scenes(seed + 1, 1, sz)
    scene_b = scene.expand(bsz, -1, -1, -1)
    orc = so.PCNetTrainOracle(sd, scene, bsz, l2_reg=1e-4, lr_drop_ratio=0.2)
    out, diff = {}, 0.0
    for it, opt in enumerate(('l1+ssim', 'l1')):
        prj = syn.scenes(seed + 20 + it, bsz, sz)
        cam = syn.scenes(seed + 30 + it, bsz, sz) * 0.8 + 0.05
        model.train()
        infer = model(prj, scene_b)
        loss, l2 = ns['compute_loss'](infer, cam, opt)
        for o in opts:
            o.zero_grad()
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in model.named_parameters()}
        for o in opts:
            o.step()
        lo, l2o = orc.step(prj, cam, opt)
        diff = max(diff, abs(lo - float(loss)), abs(l2o - float(l2)), max(float((orc.grads[k] - g).abs().max()) for k, g in grads.items()),
                   max(float((orc.p[k] - v).abs().max()) for k, v in model.named_parameters()))
        out[f'loss{it}'], out[f'l2_{it}'] = float(loss), float(l2)
        for k in ('warping_net.affine_mat', 'warping_net.theta', 'warping_net.grid_refine_net.6.bias', 'shading_net.conv6.weight',
                  'shading_net.conv1_s.bias', 'shading_net.skipConv1.0.weight', 'shading_net.transConv2.bias'):
            out[f'grad{it}.{k}'] = grads[k].numpy()
            out[f'param{it}.{k}'] = dict(model.named_parameters())[k].detach().numpy().copy()
        out[f'gradnorm{it}'] = np.array([float(grads[k].double().norm()) for k in sorted(grads)])
    print(f'  {name}: oracle maxdiff {diff:.3e}; losses {out["loss0"]:.6f} {out["loss1"]:.6f}')
    save(name, seed=seed, sz=sz, bsz=bsz, names=np.array(sorted(grads)), wsum=weights_checksum(sd), oracle_maxdiff=diff, **out)


CASES = {
    'train_32': lambda r: gen_train(r, 'train_32', (32, 32), 3, 0),
    'preproc_240_224': lambda r: gen_preproc('preproc_240_224', (256, 256), (240, 240), (224, 224), 41),
    'preproc_240_299': lambda r: gen_preproc('preproc_240_299', (256, 256), (240, 240), (299, 299), 42),
    'preproc_nonsq_small': lambda r: gen_preproc('preproc_nonsq_small', (60, 84), (56, 56), (48, 48), 43, bsz=3),
    'io_sample_png': lambda r: gen_io('io_sample_png'),
    'color_kat': lambda r: gen_color(r),
    'pcnet_64': lambda r: gen_pcnet(r, 'pcnet_64', (64, 64), (64, 64), 'rect', 0),
    'pcnet_nonsq': lambda r: gen_pcnet(r, 'pcnet_nonsq', (64, 64), (48, 80), 'ones', 3),
    'pcnet_norough_64': lambda r: gen_pcnet(r, 'pcnet_norough_64', (64, 64), (64, 64), 'rect', 6, use_rough=False),
    'pcnet_256': lambda r: gen_pcnet(r, 'pcnet_256', (256, 256), (256, 256), 'ones', 0, bsz=1),
    'spaa_64_untargeted': lambda r: gen_spaa(r, 'spaa_64_untargeted', (64, 64), False, 'true', 5, 'camdE_caml2'),
    'spaa_64_imagenet10': lambda r: gen_spaa(r, 'spaa_64_imagenet10', (64, 64), True, syn.IMAGENET10_TARGETS, 5,
                                             'camdE_caml2'),
    'spaa_64_near': lambda r: gen_spaa(r, 'spaa_64_near', (64, 64), True, ('near', 8), 5, 'camdE_caml2'),
    'spaa_64_caml2_dthr': lambda r: gen_spaa(r, 'spaa_64_caml2_dthr', (64, 64), True, ('near', 8), 40, 'caml2'),
    'spaa_64_prjl2': lambda r: gen_spaa(r, 'spaa_64_prjl2', (64, 64), True, ('near', 8), 5, 'camdE_caml2_prjl2'),
    'spaa_64_camdE': lambda r: gen_spaa(r, 'spaa_64_camdE', (64, 64), False, 'true', 2, 'camdE', mask='ones'),
    'spaa_256_untargeted': lambda r: gen_spaa(r, 'spaa_256_untargeted', (256, 256), False, 'true', 5, 'camdE_caml2',
                                              mask='ones'),
    'spaa_256_near': lambda r: gen_spaa(r, 'spaa_256_near', (256, 256), True, ('near', 8), 5, 'camdE_caml2',
                                        mask='ones', keep=2),
    'compennet_pp_64': lambda r: gen_compennet_pp(r, 'compennet_pp_64', (64, 64), 5),
    'img_dists': lambda r: gen_metrics(r, 'img_dists'),
    'percal_64_targeted': lambda r: gen_percal(r, 'percal_64_targeted', (64, 64), True, 2, 0),
    'percal_64_untargeted': lambda r: gen_percal(r, 'percal_64_untargeted', (64, 64), False, 2, 40),
    'sensitivity_64': lambda r: gen_sensitivity(r, 'sensitivity_64'),
}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', nargs='*')
    args = ap.parse_args()
    torch.manual_seed(0)
    ref = ref_shims.load_reference()
    for nm, fn in CASES.items():
        if args.only and nm not in args.only:
            continue
        print(nm)
        fn(ref)
