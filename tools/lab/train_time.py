"""Time of one PCNet training step (train_network.py:235-363) on the GPU: batch 24 at 256x256 (the reference's batch size)."""
import sys, time, torch
sys.path.insert(0, '.')
from spaa_amd import synthetic as syn, models as m_, _lib
from spaa_amd.train_network import PCNetTrainer
_lib.load()
DEV = 'cuda'
sz, bsz = (256, 256), 24
sd = syn.pcnet_state_dict(1, cam_sz=sz, mask='rect')
pc = m_.PCNet(sd['mask'], m_.WarpingNet(out_size=sz)); pc.load_state_dict(sd); pc = pc.to(DEV)
tr = PCNetTrainer(pc, syn.scenes(2, 1, sz), bsz, device=DEV)
prj, cam = syn.scenes(20, bsz, sz), syn.scenes(30, bsz, sz) * 0.8 + 0.05
for _ in range(3):
    tr.step(prj, cam, 'l1+ssim')
torch.cuda.synchronize()
t0 = time.time()
n = 10
for _ in range(n):
    loss, l2 = tr.step(prj, cam, 'l1+ssim')
torch.cuda.synchronize()
print(f'PCNet training step, batch {bsz} at {sz}: {(time.time() - t0) / n * 1e3:.1f} ms/step  (loss {loss:.4f})')
