"""A few iterations of the benchmarked loop (batch 64, 256 x 256, ResNet-18) for rocprofv3 --pmc passes over ALL its kernels
(PROF_SCRIPT=lab/loop_one.py PMC_FILTER=shading bash tools/pmc_conv.sh <tag>)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
st, *_ = bench.build_attack(0, 64, 256, 8, 'cuda:0')
for _ in range(3):
    st.step()
torch.cuda.synchronize()
