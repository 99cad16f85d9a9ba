"""Experiment: one attack iteration captured in a HIP graph (torch.cuda.CUDAGraph) vs eager launches."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

st, *_ = bench.build_attack(0, 64, 256, 8, 'cuda:0')
hp = dict(targeted=True, d_thr=5, adv_lr=2, col_lr=1, p_thresh=0.9)
for _ in range(3):
    st.iteration(**hp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    st.iteration(**hp)
torch.cuda.synchronize()
print('eager  ms/step', (time.perf_counter() - t0) / 20 * 1e3, flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    st.iteration(**hp)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st.iteration(**hp)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print('graph  ms/step', (time.perf_counter() - t0) / 20 * 1e3, flush=True)
# host-side cost of one eagerly launched iteration (no sync inside): how far ahead of the GPU the launcher runs
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    st.iteration(**hp)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host launch time per iteration ms', (t1 - t0) / 5 * 1e3, '(GPU time per iteration above)')
