"""Measures every tapconv launch of one SPAA iteration (bench workload) under each workgroup tile and writes the
fastest tile per layer shape to spaa_amd/tapconv_tune.json.  Run on the GPU box: python tools/autotune.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spaa_amd import convplan  # noqa: E402


def build_reference_shape(batch, classifier, cam, prj):
    """The reference's own call shape (main.py:19-33, projector_based_attack.py:107,120): projector `prj` -> camera `cam`,
    classifier crop 240 x 240, B = 1 (untargeted) or 10 (targeted)."""
    from spaa_amd import synthetic as syn
    from spaa_amd.models import PCNet, WarpingNet
    from spaa_amd.classifier import Classifier
    from spaa_amd.projector_based_attack import AttackState
    dev = 'cuda:0'
    sd = syn.pcnet_state_dict(0, cam_sz=cam, mask='ones')
    pc = PCNet(sd['mask'], WarpingNet(out_size=cam))
    pc.load_state_dict(sd)
    pc = pc.to(dev)
    csd = {'resnet18': syn.resnet18_state_dict, 'vgg16': syn.vgg16_state_dict,
           'inception_v3': syn.inception_v3_state_dict}[classifier](2, logit_gain=20.0)
    clf = Classifier(classifier, dev, state_dict=csd)
    setup = dict(classifier_crop_sz=(240, 240), prj_brightness=0.5, prj_im_sz=prj)
    targets = (syn.IMAGENET10_TARGETS * 8)[:batch]
    return AttackState(pc, clf, targets, syn.scenes(1, 1, cam), 'camdE_caml2', setup, dev)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    classifier = sys.argv[2] if len(sys.argv) > 2 else 'resnet18'  # other classifiers: their layer shapes are MERGED
    shape = sys.argv[3] if len(sys.argv) > 3 else ''                # e.g. 240x320: camera size of the reference's call shape
    if shape:
        cam = tuple(int(v) for v in shape.split('x'))
        st = build_reference_shape(batch, classifier, cam, (256, 256))
    else:
        st, *_ = bench.build_attack(0, batch, 256, 8, 'cuda:0', classifier)
    hp = dict(targeted=True, d_thr=5, adv_lr=2, col_lr=1, p_thresh=0.9)
    st.iteration(**hp)
    torch.cuda.synchronize()
    res = {}
    names = {}
    tiles = [int(t) for t in os.environ.get('SPAA_TUNE_TILES', '').split(',') if t] or (list(range(1, 55)) + [70, 71, 73, 170, 171, 270, 271, 470, 471, 870, 871] + [s * 100 + t for s in (2, 4, 8) for t in (25, 27, 31, 34, 35, 36, 42, 48, 50, 52)] + [900 + t for t in (48, 49, 50, 51, 52, 53, 54)])
    # the current choice (tune table / heuristic) per launch, measured in this same process: a candidate must beat it by 2 %
    cur = {}
    convplan.FORCE_TILE = 0
    st.iteration(**hp)
    convplan.PROFILE = []
    for _ in range(2):
        st.iteration(**hp)
    torch.cuda.synchronize()
    for name, key, flops, e0, e1, used, _nbytes in convplan.PROFILE:
        cur.setdefault(key, []).append(e0.elapsed_time(e1))
    convplan.PROFILE = None
    for tile in tiles:
        convplan.FORCE_TILE = tile
        st.iteration(**hp)  # warm (sets the LDS attribute of a new instantiation)
        convplan.PROFILE = []
        for _ in range(2):
            st.iteration(**hp)
        torch.cuda.synchronize()
        for name, key, flops, e0, e1, used, _nbytes in convplan.PROFILE:
            if (used % 100 not in (70, 71, 73) or (used % 100 == 73) != (tile % 100 == 73)) if tile % 100 in (70, 71, 73) else used != tile:   # (Winograd: the launcher's N tile / K ranges are reported, not the request)
                continue  # this tile is not valid for the layer (ConvPlan.run fell back)
            res.setdefault(key, {}).setdefault(tile, []).append(e0.elapsed_time(e1))
            names.setdefault(key, set()).add(name)
        convplan.PROFILE = None
        print('tile', tile, 'done', flush=True)
    convplan.FORCE_TILE = 0
    tune, total_best, total_t1 = {}, 0.0, 0.0
    rows = []
    for key, per in res.items():
        # a key may be launched several times per iteration; compare mean time per launch
        avg = {t: sum(v) / len(v) for t, v in per.items()}
        # ties (run-to-run noise is ~1-2 %) go to the plainer kernel: the LDS-coalesced-epilogue variants must win by 2 %
        best = min(avg, key=lambda t: avg[t] * (1.02 if t % 100 in (39, 40, 41, 45, 46) else 1.0))
        if key in cur and key in convplan.TUNE and sum(cur[key]) / len(cur[key]) <= 1.02 * avg[best]:
            best = convplan.TUNE[key]      # (not beaten: the entry stays)
            avg.setdefault(best, sum(cur[key]) / len(cur[key]))
            per.setdefault(best, cur[key])
        tune[key] = best
        n = len(per[best]) / 2
        total_best += avg[best] * n
        rows.append((avg[best] * n, key, sorted(names[key])[:3], best, {t: round(a * 1e3) for t, a in avg.items()}))
    for r in sorted(rows, reverse=True)[:40]:
        top = dict(sorted(r[4].items(), key=lambda kv: kv[1])[:6])
        print(f'{r[0]*1e3:8.0f} us/iter  {r[1]:32s} best={r[3]} was={convplan.TUNE.get(r[1])} {top} {r[2]}')
    print(f'sum of best per-launch times: {total_best:.2f} ms/iteration')
    out = os.path.join(ROOT, 'gpurun_out', os.environ.get('SPAA_TUNE_OUT', 'tapconv_tune.json'))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # keep the measured choices of the other workloads.  SPAA_TUNE_UPDATE=1 (and the default workload): re-measured shapes replace
    # their entries where a candidate won by 2 % (after a kernel change); else only new shapes are added
    merged = dict(convplan.TUNE)
    upd = os.environ.get('SPAA_TUNE_UPDATE', '0') == '1' or not (classifier != 'resnet18' or shape or batch != 64)
    merged.update({k: v for k, v in tune.items() if upd or k not in merged})
    tune = merged
    with open(out, 'w') as fh:
        json.dump(tune, fh, indent=0, sort_keys=True)
    print('wrote', out)


if __name__ == '__main__':
    main()
