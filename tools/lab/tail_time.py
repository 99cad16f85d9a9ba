"""Times the fused tail (forward) and head (backward) kernels inside the loop at batch 64, 256 x 256.

Round 4 (timing-only builds of the forward kernel, same box): full 311 us; ONE conv6 tap instead of nine 228 (the VALU tap loop is
~93 us); no transConv2 MFMAs 165 (the matrix-core phase ~147 us); two X6 rows x two channel blocks per wave instead of one row x
four blocks (half the weight-fragment LDS reads, but every row is split into bf16 planes by four waves instead of two): 355 us
forward / 263 backward against 311 / 229 -- the operand split, not LDS bandwidth, bounds that phase.  Dropped."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from spaa_amd import _lib
st, *_ = bench.build_attack(0, 64, 256, 8, 'cuda:0')
hp = dict(targeted=True, d_thr=5, adv_lr=2, col_lr=1, p_thresh=0.9)
for _ in range(2):
    st.iteration(**hp)
torch.cuda.synchronize()
_lib.PROFILE = []
for _ in range(5):
    st.iteration(**hp)
torch.cuda.synchronize()
acc = {}
for name, e0, e1 in _lib.PROFILE:
    if 'shading' in name:
        acc.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
_lib.PROFILE = None
print({k: round(sum(v) / len(v), 1) for k, v in acc.items()})
