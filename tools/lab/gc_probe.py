"""How long does a full collection of the host's cyclic garbage collector take in a bench process, how often would one fall into a timed
window, and how much does the host's enqueue time per step jitter?  (bench.py switches the collector off between its timing brackets.)

    python tools/lab/gc_probe.py
"""
import sys, time, gc
sys.path.insert(0, '.')
import torch, bench
st, *_ = bench.build_attack(0, 64, 256, 8, 'cuda:0', 'resnet18', 'f16')
for _ in range(3): st.step()
torch.cuda.synchronize()
print('objects tracked', len(gc.get_objects()), 'counts', gc.get_count(), 'thresholds', gc.get_threshold())
for i in range(3):
    t=time.perf_counter(); n=gc.collect(); print('gc.collect()', n, round((time.perf_counter()-t)*1e3,2), 'ms')
# how many allocations per step trigger gen0?
c0=gc.get_count(); st.step(); print('count delta per step', gc.get_count(), c0)
ts=[]
for _ in range(200):
    t=time.perf_counter(); st.step(); ts.append(time.perf_counter()-t)
torch.cuda.synchronize()
ts=sorted(ts); print('host enqueue per step ms: median', round(ts[100]*1e3,3), 'p99', round(ts[197]*1e3,3), 'max', round(ts[-1]*1e3,3))
