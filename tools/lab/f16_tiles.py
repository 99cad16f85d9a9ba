"""fp16-storage SPAA iteration: every tapconv launch under each implicit-GEMM tile (60..65), us per launch; the default choice
(convplan.ConvPlan.run: by GEMM width and pixel count, 68 / 72 where eligible) beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from spaa_amd import convplan
st, *_ = bench.build_attack(0, 64, 256, 8, 'cuda:0', sys.argv[1] if len(sys.argv) > 1 else 'resnet18', 'f16')
hp = dict(targeted=True, d_thr=5, adv_lr=2, col_lr=1, p_thresh=0.9)
st.iteration(**hp)
torch.cuda.synchronize()
res, tiles_used = {}, {}
for tile in (0, 60, 61, 62, 64, 65):
    convplan.FORCE_TILE = tile
    st.iteration(**hp)
    convplan.PROFILE = []
    for _ in range(2):
        st.iteration(**hp)
    torch.cuda.synchronize()
    for name, key, flops, e0, e1, used, _nb in convplan.PROFILE:
        if tile and used != tile:
            continue
        res.setdefault(name, {}).setdefault(tile, []).append(e0.elapsed_time(e1) * 1e3)
        if tile == 0:
            tiles_used[name] = used
    convplan.PROFILE = None
convplan.FORCE_TILE = 0
gain = 0.0
for name, per in sorted(res.items(), key=lambda kv: -sum(kv[1].get(0, [0])) / max(1, len(kv[1].get(0, [0])))):
    avg = {t: sum(v) / len(v) for t, v in per.items()}
    if 0 not in avg:
        continue
    best = min(avg, key=avg.get)
    if avg[best] < 0.95 * avg[0]:
        gain += avg[0] - avg[best]
        print(f'{name:24s} default {tiles_used[name]:3d}: {avg[0]:6.1f} us   best {best}: {avg[best]:6.1f}   ' + ' '.join(f'{t}:{a:.0f}' for t, a in sorted(avg.items())))
print(f'sum of gains over the default choice: {gain:.0f} us per iteration')
