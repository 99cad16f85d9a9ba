"""Which form of round 5 moved the sub-batch statistic of tests/test_gpu_parity.py::test_perc_al_vgg16_f16_full_batch_properties
(VGG-16 + PerC-AL, fp16 storage, batch 64 at 256 x 256: iteration-0 delta of a sub-batch of 8 against rows 8..15 of the batch of 64)
from 0.046 (rounds 3-4) to 0.10 (round 5)?  One child process per switch setting (the switches are read when the plans are built); every
child prints the statistic, and its batch-64 / batch-8 deltas are compared with the default build's (rel L2 0.0 = arithmetic-preserving form).

    python tools/lab/vgg_f16_bisect.py            # parent: spawns the children one after another (one process on the GPU at a time)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EMPTY_TUNE = '/tmp/spaa_empty_tune.json'

CONFIGS = [
    ('default', {}),
    ('c3h off (first layer: fp32 image operands)', {'SPAA_DEFAULT_DISABLE': 'c3h'}),
    ('h16ppool off (pool not in the conv epilogue)', {'SPAA_DEFAULT_DISABLE': 'h16ppool'}),
    ('h16punp off (un-pool not in the dgrad prologue)', {'SPAA_DEFAULT_DISABLE': 'h16punp'}),
    ('FUSE_POOL=0', {'SPAA_FUSE_POOL': '0'}),
    ('lean-wide off', {'SPAA_H16P_LEAN_WIDE': '0'}),
    ('h16pcv off (no canvas / K ranges in the patch kernel)', {'SPAA_DEFAULT_DISABLE': 'h16pcv'}),
    ('h16splitk off', {'SPAA_DEFAULT_DISABLE': 'h16splitk'}),
    ('h16p64 (round-4 threshold of the patch kernel)', {'SPAA_DEFAULT_DISABLE': 'h16p64'}),
    ('h16p off (no patch-staged kernel at all)', {'SPAA_DEFAULT_DISABLE': 'h16p,h16p2'}),
    ('body byte masks off', {'SPAA_BODY_MASKS': '0'}),
    ('small linear off', {'SPAA_SMALL_LINEAR': '0'}),
    ('no tune table (rules choose every tile)', {'SPAA_TUNE_FILE': EMPTY_TUNE}),
    ('no tune table, c3h off', {'SPAA_TUNE_FILE': EMPTY_TUNE, 'SPAA_DEFAULT_DISABLE': 'c3h'}),
    ('no tune table, no patch-staged kernel, c3h off (round 3-4: the implicit-GEMM tile at both batch sizes)',
     {'SPAA_TUNE_FILE': EMPTY_TUNE, 'SPAA_DEFAULT_DISABLE': 'c3h,h16p,h16p2,h16pcv'}),
    ('the same + no K ranges', {'SPAA_TUNE_FILE': EMPTY_TUNE, 'SPAA_DEFAULT_DISABLE': 'c3h,h16p,h16p2,h16pcv,h16splitk'}),
    ('all round-5 forms off', {'SPAA_DEFAULT_DISABLE': 'c3h,h16ppool,h16punp,h16pcv,h16p64', 'SPAA_FUSE_POOL': '0', 'SPAA_H16P_LEAN_WIDE': '0'}),
]


def child(tag):
    sys.path.insert(0, ROOT)
    import torch
    from spaa_amd import synthetic as syn
    from spaa_amd import classifier as clfm
    from spaa_amd.perc_al import PerC_AL, PerCALState
    dev = torch.device('cuda:0')
    csd = syn.vgg16_state_dict(2, logit_gain=20.0)
    clf = clfm.Classifier('vgg16', dev, state_dict=csd)
    scenes = syn.scenes(11, 8, (256, 256)).repeat_interleave(8, dim=0)
    labels = torch.tensor((syn.IMAGENET10_TARGETS[:8]) * 8)
    att = PerC_AL(device=dev, max_iterations=400, alpha_l_init=1, alpha_c_init=0.5, confidence=0, storage='f16')

    def run(sc, lb):
        with torch.cuda.device(dev):
            st = PerCALState(att, clf, sc, lb, 5.0, True, (240, 240))
        st.iteration(0)
        return st.delta.clone()

    def rel_l2(a, b):
        return float((a - b).norm() / b.norm())

    d64 = run(scenes, labels)
    d8 = run(scenes[8:16].contiguous(), labels[8:16])
    out = {'e8': rel_l2(d8, d64[8:16])}
    ref = '/tmp/vgg_f16_bisect_default.pt'
    if tag == 'default':
        torch.save({'d64': d64.cpu(), 'd8': d8.cpu()}, ref)
    elif os.path.exists(ref):
        r = torch.load(ref)
        out['d64_vs_default'] = rel_l2(d64.cpu(), r['d64'])
        out['d8_vs_default'] = rel_l2(d8.cpu(), r['d8'])
    print('RESULT', tag, out, flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(sys.argv[2])
        return
    with open(EMPTY_TUNE, 'w') as fh:
        fh.write('{}')
    for name, env in CONFIGS:
        e = dict(os.environ)
        e.update(env)
        tag = 'default' if not env else name
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', tag], env=e, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')]
        print(f'{name:55s} {env}: {line[0][7:] if line else "FAILED rc=%d %s" % (r.returncode, r.stderr[-400:])}', flush=True)


if __name__ == '__main__':
    main()
