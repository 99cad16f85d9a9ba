"""VGG-16 + PerC-AL, fp16 storage, batch 64 at 256 x 256 (tests/test_gpu_parity.py::test_perc_al_vgg16_f16_full_batch_properties): the first
iteration's delta under the forms of the patch-staged fp16 kernel -- canvas / K-range form on or off (h16pcv), two workgroups per CU on or
off (h16plean: the SAME arithmetic, must be bitwise equal) -- and the sub-batch statistic of that test under each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import torch
from spaa_amd import synthetic as syn
from spaa_amd import convplan as cp, classifier as clfm
from spaa_amd.perc_al import PerC_AL, PerCALState
DEV = torch.device('cuda:0')
csd = syn.vgg16_state_dict(2, logit_gain=20.0)
clf = clfm.Classifier('vgg16', DEV, state_dict=csd)
scenes = syn.scenes(11, 8, (256, 256)).repeat_interleave(8, dim=0)
labels = torch.tensor((syn.IMAGENET10_TARGETS[:8]) * 8)
att = PerC_AL(device=DEV, max_iterations=400, alpha_l_init=1, alpha_c_init=0.5, confidence=0, storage='f16')
def run(sc, lb):
    with torch.cuda.device(DEV):
        st = PerCALState(att, clf, sc, lb, 5.0, True, (240, 240))
    st.iteration(0)
    return st.delta.clone()
def rel_l2(a, b):
    return float((a - b).norm() / b.norm())
res = {}
for name, dis in [('both', ()), ('no lean', ('h16plean',)), ('no canvas', ('h16pcv',)), ('neither', ('h16plean', 'h16pcv'))]:
    for d in ('h16plean', 'h16pcv'):
        cp.DEFAULT_DISABLE.discard(d)
    for d in dis:
        cp.DEFAULT_DISABLE.add(d)
    d64 = run(scenes, labels)
    d8 = run(scenes[8:16].contiguous(), labels[8:16])
    res[name] = d64
    print(f'{name:10s}: sub-batch of 8 vs rows 8..15 of 64: rel L2 {rel_l2(d8, d64[8:16]):.3e}', flush=True)
print('lean vs no lean (canvas on):  bitwise equal', torch.equal(res['both'], res['no lean']), ' rel L2', rel_l2(res['both'], res['no lean']))
print('lean vs no lean (canvas off): bitwise equal', torch.equal(res['no canvas'], res['neither']), ' rel L2', rel_l2(res['no canvas'], res['neither']))
print('canvas vs no canvas: rel L2', rel_l2(res['both'], res['no canvas']))

# the same statistic in fp32 storage (no fp16 rounding of activations: what is left is summation order) and its robust form in fp16
att32 = PerC_AL(device=DEV, max_iterations=400, alpha_l_init=1, alpha_c_init=0.5, confidence=0, storage='f32')
def run32(sc, lb):
    with torch.cuda.device(DEV):
        st = PerCALState(att32, clf, sc, lb, 5.0, True, (240, 240))
    st.iteration(0)
    return st.delta.clone()
a, b8 = run32(scenes, labels), run32(scenes[8:16].contiguous(), labels[8:16])
print(f'fp32 storage: sub-batch of 8 vs rows 8..15 of 64: rel L2 {rel_l2(b8, a[8:16]):.3e}')
for d in ('h16plean', 'h16pcv'):
    cp.DEFAULT_DISABLE.discard(d)
d64, d8 = run(scenes, labels), run(scenes[8:16].contiguous(), labels[8:16])
diff = (d8 - d64[8:16]).abs()
print(f'fp16 storage: |delta| max {float(d64.abs().max()):.3e}, mean {float(d64.abs().mean()):.3e}; |difference| max {float(diff.max()):.3e}, '
      f'mean {float(diff.mean()):.3e}, fraction above 10 % of max |delta| {float((diff > 0.1 * d64.abs().max()).float().mean()):.3e}')
print(f'fp16 vs fp32 storage, batch 64: rel L2 {rel_l2(d64, a):.3e}')
