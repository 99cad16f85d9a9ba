"""ORACLE — TEST INFRASTRUCTURE ONLY (this container only; /root/reference does not exist on the GPU box).

Imports the *unmodified* reference modules from /root/reference/src/python so that
`tests/golden/make_golden.py` can run the real `PCNet`, `rgb2lab_diff`, `ciede2000_diff`, `spaa()` and
`PerC_AL.adversary_projector()` and commit their outputs as golden fixtures.  Nothing from the
reference is copied: missing third-party modules that the hot path never touches (cv2, torchvision,
omegaconf, visdom-connecting `utils`, ...) are replaced by inert stand-ins in `sys.modules`, and
`img_proc` is replaced by the three tensor helpers the path uses (restated in oracle/spaa_oracle.py
from img_proc.py:110-132).  No permission denial occurred when importing the reference.
"""
import os
import sys
import types
from unittest.mock import MagicMock

REF_ROOT = '/root/reference/src/python'


def reference_available():
    return os.path.isdir(REF_ROOT)


def load_reference():
    """Returns a namespace with the reference's hot-path symbols."""
    if not reference_available():
        raise RuntimeError('reference sources are not present (only available in the build container)')
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import spaa_oracle as so

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    img_proc = types.ModuleType('img_proc')
    img_proc.expand_4d = so.expand_4d
    img_proc.center_crop = so.center_crop
    img_proc.resize = lambda x, size: so.resize_area(so.expand_4d(x), size)
    img_proc.insert_text = MagicMock()
    img_proc.expand_boarder = MagicMock()
    sys.modules['img_proc'] = img_proc
    for name in ('cv2', 'omegaconf', 'torchvision', 'torchvision.models', 'torchvision.transforms', 'torchvision.utils',
                 'visdom',
                 'train_network', 'utils', 'classifier', 'one_pixel_attacker', 'skimage', 'skimage.filters',
                 'skimage.metrics', 'pytorch_ssim', 'matplotlib', 'matplotlib.pyplot'):
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    import models as ref_models
    import pytorch_tps as ref_tps
    from perc_al import differential_color_functions as ref_color
    import perc_al as ref_perc_al
    import projector_based_attack as ref_attack
    ns = types.SimpleNamespace(models=ref_models, tps=ref_tps, color=ref_color, perc_al=ref_perc_al,
                               attack=ref_attack)
    return ns


def make_reference_pcnet(ref, sd, prj_sz, cam_sz):
    """Reference PCNet with our deterministic weights loaded via load_state_dict."""
    wn = ref.models.WarpingNet(out_size=tuple(cam_sz))
    sn = ref.models.ShadingNetSPAA()
    holder = types.SimpleNamespace  # PCNet deep-copies `.module`
    pc = ref.models.PCNet(sd['mask'], holder(module=wn), holder(module=sn), fix_shading_net=True)
    pc.load_state_dict(sd)
    pc.eval()
    for p in pc.parameters():
        p.requires_grad = False
    return pc
