"""Where a workgroup of the patch-staged stride-2 kernel (tile 74, csrc/tapconv_x6p.hip) spends its life: in-kernel s_memtime stamps
of the diagnostic build (`make -C spaa_amd/csrc stamp` -> spaa_amd/libspaa_hip_stamp.so).  Per wave: prologue (second source + first
DMAs), main loop (of which: waiting for vmcnt / lgkmcnt, waiting at the per-step barrier), epilogue; the clock the chip held."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from spaa_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libspaa_hip_stamp.so')
from spaa_amd import convplan as cp
lib = _lib.load()
lib.spaa_x6p_set_stamp_buffer.argtypes = [ctypes.c_void_p]
DEV = torch.device('cuda:0')
torch.manual_seed(0)
B = 64


def t_launch(plan, x, out, n=10, **kw):
    for _ in range(3):
        plan.run(x, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        plan.run(x, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ablate(name, plan, x, out, **kw):
    """timing-only ablations of the diagnostic build (reserved0 bits 8-15 = convplan.DEBUG_PERSIST_CAP): wrong results"""
    cp.FORCE_TILE = 74
    res = []
    for label, bits in (('full', 0), ('no DMA', 1), ('no split', 2), ('no barrier', 4), ('no epilogue', 16), ('no frag loads+split', 34),
                        ('no MFMA', 8), ('no DMA, split, loads', 35), ('MFMA + weight reads only', 1 | 2 | 4 | 32), ('only DMA + barrier', 2 | 8 | 32 | 16),
                        ('nothing but the epilogue', 1 | 2 | 4 | 8 | 32)):
        cp.DEBUG_PERSIST_CAP = bits
        res.append(f'{label}: {t_launch(plan, x, out, **kw):.0f}')
    cp.DEBUG_PERSIST_CAP = 0
    cp.FORCE_TILE = 0
    print(f'{name} ablations (us per launch): ' + '   '.join(res), flush=True)


def run(name, plan, x, out, nwg, **kw):
    cp.FORCE_TILE = 74
    cp.DEBUG_PERSIST_CAP = 128
    for _ in range(5):
        plan.run(x, out, **kw)
    buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=DEV)
    assert lib.spaa_x6p_set_stamp_buffer(buf.data_ptr()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan.run(x, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    assert plan.last_tile == 74
    lib.spaa_x6p_set_stamp_buffer(None)
    s = buf.cpu().numpy().reshape(nwg, 4, 8).astype(np.float64)
    life = s[..., 3] - s[..., 0]
    clk = life / ((s[..., 5] - s[..., 4]) * 10.0)          # cycles per ns (s_memrealtime ticks at 100 MHz)
    ghz = float(np.median(clk))
    us = lambda c: float(np.median(c)) / ghz / 1e3        # noqa: E731
    span = (s[..., 5].max() - s[..., 4].min()) * 10.0 / 1e3
    print(f'{name}: launch {e0.elapsed_time(e1) * 1e3:.0f} us (stamped span {span:.0f} us), {nwg} workgroups = {nwg / 512:.1f} rounds of 2 per CU, '
          f'clock {ghz:.2f} GHz; per wave (median, us): life {us(life):.1f} = prologue {us(s[..., 1] - s[..., 0]):.1f} + main loop '
          f'{us(s[..., 2] - s[..., 1]):.1f} (vmcnt / lgkmcnt waits {us(s[..., 7]):.1f}, barrier waits {us(s[..., 6]):.1f}) + epilogue '
          f'{us(s[..., 3] - s[..., 2]):.1f};  p10 / p90 life {np.percentile(life, 10) / ghz / 1e3:.1f} / {np.percentile(life, 90) / ghz / 1e3:.1f}', flush=True)
    cp.FORCE_TILE = 0
    cp.DEBUG_PERSIST_CAP = 0


wt = torch.randn(128, 64, 3, 3) / 34
x5 = torch.relu(torch.randn(B, 64, 64, 128, device=DEV)); x6 = torch.zeros(B, 128, 128, 64, device=DEV); r2 = torch.randn(B, 128, 128, 64, device=DEV)
m6 = torch.zeros(B, 128, 128, 16, dtype=torch.uint8, device=DEV)
ws = torch.randn(64, 32, 1, 1) / 6; x1 = torch.relu(torch.randn(B, 128, 128, 32, device=DEV))
plan = cp.deconv_fwd_plan(wt, torch.randn(64), 2, 1, DEV, 'transConv1', fold=False)
ablate('transConv1 plain', plan, x5, x6)
ablate('transConv1 +add+relu+mask', plan, x5, x6, add=r2, act=_lib.ACT_RELU, mask_out=m6)
run('transConv1 +add+relu+mask', plan, x5, x6, 2048, add=r2, act=_lib.ACT_RELU, mask_out=m6)
run('transConv1 plain (no epilogue operands)', plan, x5, x6, 2048)
plan.attach_second_source(ws, None)
run('transConv1 + skipConv2 fused +relu+mask', plan, x5, x6, 2048, inp2=x1, act=_lib.ACT_RELU, mask_out=m6)
w2c = torch.randn(64, 32, 3, 3) / 17
p2 = torch.randn(B, 64, 64, 64, device=DEV); p1 = torch.zeros(B, 128, 128, 32, device=DEV); t1 = torch.randn(B, 128, 128, 32, device=DEV)
m1 = (torch.rand(B, 128, 128, 8, device=DEV) * 16).to(torch.uint8)
pl2 = cp.conv_dgrad_plan(w2c, 2, 1, DEV, 'conv2_dgrad', fold=False)
ablate('conv2_dgrad +add+gatebits', pl2, p2, p1, add=t1, gate_bits=m1)
run('conv2_dgrad +add+gatebits', pl2, p2, p1, 2048, add=t1, gate_bits=m1)
run('conv2_dgrad plain', pl2, p2, p1, 2048)
