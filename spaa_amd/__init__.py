"""spaa_amd — the SPAA attack loop (PCNet + frozen classifier + CIEDE2000/L2 stealth loss + alternating
normalised-gradient step) on hand-written HIP kernels for MI355X (gfx950).

Scope: the one hot path of BingyaoHuang/SPAA named in BASELINE.json (`spaa()` and what it calls), behind the
reference's own Python call signatures.  See DESIGN.md and INTEGRATION.md.
"""
from . import synthetic  # noqa: F401  (host-side data generation only)


def __getattr__(name):
    # Lazy: importing the package must not require the GPU library (CPU-side tests import `synthetic`/`convplan`).
    if name in ('PCNet', 'WarpingNet', 'ShadingNetSPAA', 'PCNetEngine'):
        from . import models
        return getattr(models, name)
    if name in ('Classifier', 'ClassifierEngine', 'load_imagenet_labels'):
        from . import classifier
        return getattr(classifier, name)
    if name in ('spaa', 'spaa_attack', 'AttackState'):
        from . import projector_based_attack
        return getattr(projector_based_attack, name)
    if name in ('rgb2lab_diff', 'ciede2000_diff', 'deltaE', 'stealth_loss_with_grad'):
        from . import differential_color_functions
        return getattr(differential_color_functions, name)
    if name in ('CompenNet', 'CompenNetPlusplus'):
        from . import models
        return getattr(models, name)
    if name in ('PerC_AL', 'perc_al_compennet_pp'):
        from . import perc_al
        return getattr(perc_al, name)
    if name == 'calc_img_dists':
        from . import metrics
        return metrics.calc_img_dists
    if name in ('torch_imread', 'torch_imread_mt', 'save_imgs', 'load_setup_info', 'save_checkpoint'):
        from . import io
        return getattr(io, name)
    raise AttributeError(name)
