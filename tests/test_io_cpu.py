"""CPU: on-disk conventions of spaa_amd/io.py (reference: utils.py:84-167, :674-680, :717-721; train_network.py:85-95)."""
import os

import numpy as np
import pytest
import torch

from spaa_amd import io as sio
from spaa_amd import synthetic as syn


def test_png_round_trip_truncates_like_np_uint8(tmp_path):
    x = torch.rand(3, 3, 17, 23)
    x[0, :, 0, 0] = torch.tensor([0.999, 0.5, 1.0])      # 254.7 -> 254 (truncation, not rounding); 1.0 -> 255
    sio.save_imgs(x, str(tmp_path / 'out'), idx=4)
    assert sorted(os.listdir(tmp_path / 'out')) == ['img_0005.png', 'img_0006.png', 'img_0007.png']
    y = sio.torch_imread_mt(str(tmp_path / 'out'))
    assert y.shape == x.shape and y.dtype == torch.float32
    assert torch.equal(y, torch.from_numpy(np.uint8(x.numpy() * 255)).float() / 255)
    assert torch.equal(sio.torch_imread(str(tmp_path / 'out' / 'img_0005.png')), y[0])
    # channel order on disk is RGB: a pure-red image reads back red
    red = torch.zeros(1, 3, 4, 4)
    red[:, 0] = 1
    sio.save_imgs(red, str(tmp_path / 'red'))
    assert sio.torch_imread(str(tmp_path / 'red' / 'img_0001.png'))[:, 0, 0].tolist() == [1.0, 0.0, 0.0]
    # uint8 arrays in NHWC are written as they are
    arr = (np.arange(2 * 5 * 6 * 3) % 256).astype(np.uint8).reshape(2, 5, 6, 3)
    sio.save_imgs(arr, str(tmp_path / 'u8'))
    assert np.array_equal((sio.torch_imread_mt(str(tmp_path / 'u8')) * 255).round().byte().numpy().transpose(0, 2, 3, 1), arr)


def test_imread_mt_options(tmp_path):
    x = syn.scenes(3, 4, (32, 48))
    sio.save_imgs(x, str(tmp_path / 'd'))
    sub = sio.torch_imread_mt(str(tmp_path / 'd'), index=[2, 0])
    full = sio.torch_imread_mt(str(tmp_path / 'd'))
    assert torch.equal(sub, full[[2, 0]])
    g = sio.torch_imread_mt(str(tmp_path / 'd'), gray_scale=True, normalize=True)
    want = ((0.2989 * full[:, 0] + 0.5870 * full[:, 1] + 0.1140 * full[:, 2])[:, None] - 0.5) / 0.5
    assert g.shape == (4, 1, 32, 48) and torch.allclose(g, want)
    r = sio.torch_imread_mt(str(tmp_path / 'd'), size=(16, 24))
    assert r.shape == (4, 3, 16, 24) and float((r - torch.nn.functional.avg_pool2d(full, 2)).abs().max()) < 1.5 / 255


def test_setup_info_and_checkpoint_names(tmp_path):
    cfg = dict(prj_screen_sz=(800, 600), prj_im_sz=(256, 256), cam_im_sz=(320, 240), classifier_crop_sz=(240, 240),
               prj_brightness=0.5, delay_frames=13, delay_time=0.02)
    sio.save_setup_info(str(tmp_path / 'setups' / 'lotion'), cfg)
    got = sio.load_setup_info(str(tmp_path / 'setups' / 'lotion'))
    assert dict(got) == cfg and got.prj_im_sz == (256, 256) and got['prj_brightness'] == 0.5
    os.rename(tmp_path / 'setups' / 'lotion' / 'setup_info.yml', tmp_path / 'setups' / 'setup_info_default.yml')
    with pytest.warns(UserWarning):
        assert dict(sio.load_setup_info(str(tmp_path / 'setups' / 'lotion'))) == cfg
    opt = dict(setup_name='lotion', model_name='PCNet', loss='l1+ssim', num_train=500, batch_size=24, max_iters=2000,
               lr=0.001, lr_drop_ratio=0.2, lr_drop_rate=800, l2_reg=0.0005)
    title = sio.opt_to_string(opt)
    assert title == 'lotion_PCNet_l1+ssim_500_24_2000_0.001_0.2_800_0.0005'
    lin = torch.nn.Linear(3, 2)
    fn = sio.save_checkpoint(str(tmp_path / 'checkpoint'), lin, title)
    assert fn.endswith(title + '.pth')
    lin2 = sio.load_checkpoint(torch.nn.Linear(3, 2), fn)
    assert all(torch.equal(a, b) for a, b in zip(lin.state_dict().values(), lin2.state_dict().values()))


def test_reads_the_reference_sample_png(golden_dir):
    """SURVEY §8 f3 with a reference-held file: tests/golden/anemone_fish.png is /root/reference/data/sample/anemone_fish.png
    (a data file, copied by tests/golden/make_golden.py).  torch_imread (Pillow) must return exactly the bytes the file
    holds, as decoded by a codec-independent PNG decoder (tests/png_ref.py) and pinned by the committed hash —
    the reference reads the same bytes through cv.imread + BGR->RGB (utils.py:116-117)."""
    import hashlib
    from png_ref import decode_png
    path = os.path.join(golden_dir, 'anemone_fish.png')
    z = np.load(os.path.join(golden_dir, 'io_sample_png.npz'))
    rgb = decode_png(path)
    assert rgb.shape == tuple(z['shape']) == (256, 256, 3)
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == str(z['sha256'])
    im = sio.torch_imread(path)
    assert im.shape == (3, 256, 256) and im.dtype == torch.float32 and float(im.max()) <= 1.0
    assert torch.equal(im, torch.from_numpy(rgb.transpose(2, 0, 1).copy()).float() / 255)
    assert np.array_equal(rgb[:4, :4], z['corner']) and np.array_equal(rgb[126:130, 126:130], z['center'])
    assert np.allclose(rgb.reshape(-1, 3).mean(0), z['mean_rgb'])


def test_loads_an_omegaconf_style_setup_info(tmp_path):
    """train_network.py:85-95 reads `setup_info.yml` written by OmegaConf.save (utils.py:674-675): block-style YAML with
    tuples as lists.  The values are main.py:19-33's defaults."""
    text = ('prj_screen_sz:\n- 800\n- 600\nprj_im_sz:\n- 256\n- 256\nprj_offset:\n- 3840\n- 0\ncam_raw_sz:\n- 1280\n- 720\n'
            'cam_crop_sz:\n- 960\n- 720\ncam_im_sz:\n- 320\n- 240\nclassifier_crop_sz:\n- 240\n- 240\n'
            'prj_brightness: 0.5\ndelay_frames: 13\ndelay_time: 0.02\n')
    d = tmp_path / 'setups' / 'camera'
    os.makedirs(d)
    (d / 'setup_info.yml').write_text(text)
    cfg = sio.load_setup_info(str(d))
    assert cfg.prj_im_sz == (256, 256) and cfg.cam_im_sz == (320, 240) and cfg.classifier_crop_sz == (240, 240)
    assert cfg['prj_brightness'] == 0.5 and cfg.delay_frames == 13 and cfg.prj_offset == (3840, 0)
