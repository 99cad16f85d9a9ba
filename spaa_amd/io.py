"""On-disk formats either side of the attack path, with the reference's conventions (host side, no GPU work).

Mirrors /root/reference/src/python:
  utils.py:84-167     SimpleDataset / torch_imread / torch_imread_mt / save_imgs  (PNG via OpenCV there: BGR on disk order
                      is an OpenCV-internal detail, files hold ordinary RGB PNGs; float images are written with
                      np.uint8(x * 255), i.e. TRUNCATION, file names img_%04d.png counted from 1 + idx)
  train_network.py:85-95, utils.py:674-675   load_setup_info / save of setup_info.yml (OmegaConf/yaml mapping)
  utils.py:679-680, :717-721                 opt_to_string / save_checkpoint (state_dict in `<dir>/<title>.pth`)
so that a setup directory captured and trained by the reference can be consumed, and results land where its
`summarize_*` functions expect them.  Pillow replaces OpenCV as the codec (OpenCV is not a dependency of this package).
"""
import os
import warnings
from os.path import abspath, join

import numpy as np
import torch
import torch.nn.functional as F
import yaml
from PIL import Image


def _imread_rgb(filename):
    with Image.open(filename) as im:
        return np.asarray(im.convert('RGB'))  # cv.imread(...)[..., ::-1]: 8-bit, 3 channels, alpha dropped


def torch_imread(filename):
    """utils.py:116-117: float tensor [3,H,W] in [0,1]."""
    return torch.from_numpy(_imread_rgb(filename).transpose(2, 0, 1).copy()).float() / 255


def torch_imread_mt(img_dir, size=None, index=None, gray_scale=False, normalize=False):
    """utils.py:120-143: every image of a directory in sorted order -> [N,3,H,W] (or [N,1,H,W]) in [0,1] ([-1,1]).
    `size` is (h, w); resizing is bilinear with half-pixel centres like cv.resize's default (the reference resizes the
    uint8 image in fixed point: results can differ by one grey level)."""
    names = sorted(os.listdir(img_dir))
    if index is not None:
        names = [names[i] for i in index]
    ims = []
    for n in names:
        path = join(img_dir, n)
        assert os.path.isfile(path), path + ' does not exist'
        im = torch.from_numpy(_imread_rgb(path).transpose(2, 0, 1).copy()).float()
        if size is not None and tuple(im.shape[-2:]) != tuple(size):
            im = F.interpolate(im[None], tuple(size), mode='bilinear', align_corners=False)[0].round().clamp(0, 255)
        ims.append(im)
    imgs = torch.stack(ims).div(255)
    if gray_scale:
        imgs = (0.2989 * imgs[:, 0] + 0.5870 * imgs[:, 1] + 0.1140 * imgs[:, 2])[:, None]
    if normalize:
        imgs = (imgs - 0.5) / 0.5
    return imgs


def save_imgs(im_4d, path, idx=0):
    """utils.py:146-167: [N,3,H,W] tensor or [N,H,W,3] array -> path/img_%04d.png numbered from idx + 1; float images are
    scaled by 255 and truncated to uint8 exactly as `np.uint8(x * 255)` does."""
    os.makedirs(path, exist_ok=True)
    if isinstance(im_4d, torch.Tensor):
        imgs = im_4d.detach().cpu().numpy().transpose(0, 2, 3, 1)
    else:
        imgs = np.asarray(im_4d)
    if imgs.dtype == np.float32:
        imgs = np.uint8(imgs * 255)
    for i in range(imgs.shape[0]):
        Image.fromarray(np.ascontiguousarray(imgs[i])).save(join(path, 'img_{:04d}.png'.format(i + 1 + idx)))


class SetupInfo(dict):
    """Mapping with attribute access (the reference passes an OmegaConf DictConfig: both `cfg.key` and `cfg['key']`)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def load_setup_info(setup_path):
    """train_network.py:85-95: `<setup>/setup_info.yml`, else `<setup>/../setup_info_default.yml` with a warning."""
    fn = join(setup_path, 'setup_info.yml')
    if not os.path.exists(fn):
        default = join(setup_path, '../setup_info_default.yml')
        warnings.warn(f'{fn} not found, loading {default} instead')
        fn = default
    with open(fn) as fh:
        cfg = yaml.safe_load(fh)
    return SetupInfo({k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items()})


def save_setup_info(setup_path, cfg):
    """utils.py:672-675."""
    os.makedirs(setup_path, exist_ok=True)
    with open(join(setup_path, 'setup_info.yml'), 'w') as fh:
        yaml.safe_dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(cfg).items()}, fh)


def opt_to_string(opt):
    """utils.py:679-680: the checkpoint / log title of a training configuration."""
    return (f'{opt["setup_name"]}_{opt["model_name"]}_{opt["loss"]}_{opt["num_train"]}_{opt["batch_size"]}_{opt["max_iters"]}_'
            f'{opt["lr"]}_{opt["lr_drop_ratio"]}_{opt["lr_drop_rate"]}_{opt["l2_reg"]}')


def save_checkpoint(checkpoint_dir, model, title):
    """utils.py:717-721."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    fn = abspath(join(checkpoint_dir, title + '.pth'))
    torch.save(model.state_dict(), fn)
    return fn


def load_checkpoint(model, filename, map_location='cpu'):
    """Loads a reference-trained `.pth` (state_dict, possibly with nested DataParallel 'module.' prefixes) into a
    spaa_amd module (PCNet / CompenNetPlusplus strip the prefixes themselves)."""
    model.load_state_dict(torch.load(filename, map_location=map_location))
    return model
