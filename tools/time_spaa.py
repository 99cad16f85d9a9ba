"""End-to-end wall time of spaa() (50 iterations, B=64, 256x256): first call (plans built) and repeat calls."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spaa_amd import synthetic as syn
from spaa_amd.models import PCNet, WarpingNet
from spaa_amd.classifier import Classifier
from spaa_amd.projector_based_attack import spaa

dev = 'cuda:0'
sz = (256, 256)
t0 = time.perf_counter()
sd = syn.pcnet_state_dict(0, cam_sz=sz, mask='ones')
pc = PCNet(sd['mask'], WarpingNet(out_size=sz)); pc.load_state_dict(sd); pc = pc.to(dev)
clf = Classifier('resnet18', dev, state_dict=syn.resnet18_state_dict(2, logit_gain=20.0))
torch.cuda.synchronize()
print(f'model construction {time.perf_counter() - t0:.2f} s', flush=True)
setup = dict(classifier_crop_sz=(240, 240), prj_brightness=0.5, prj_im_sz=sz)
scene = syn.scenes(1, 1, sz)
targets = (syn.IMAGENET10_TARGETS * 8)[:64]
for i in range(3):
    t0 = time.perf_counter()
    cam, prj = spaa(pc, clf, None, targets, True, scene[0], 5, 'camdE_caml2', dev, setup)
    torch.cuda.synchronize()
    print(f'spaa() call {i}: {time.perf_counter() - t0:.2f} s for 50 iterations x 64 targets', flush=True)
