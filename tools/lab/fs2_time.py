"""Timing of csrc/fs2_h16.hip (fp16 storage, round 6) on the three fractional-stride layers of ShadingNetSPAA at batch 64, 256 x 256:
transConv1 + skipConv2 (128 -> 64 at 64^2 -> 128^2, + 32-channel second source), conv2^T + skipConv2^T (64 -> 32, + 64-channel second source, byte
gate), conv2_s^T (64 -> 32, residual + byte gate).   python tools/lab/fs2_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from spaa_amd import _lib, models as M
_lib.load()
DEV = 'cuda'
B, H, W = 64, 64, 64
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, ci, co, ci2, relu, use_bias, use_add, use_gate, use_mask in (('transConv1+skipConv2', 128, 64, 32, 1, 1, 0, 0, 1), ('conv2^T+skipConv2^T', 64, 32, 64, 0, 0, 0, 1, 0),
                                                                     ('conv2_s^T', 64, 32, 0, 0, 0, 1, 1, 0), ('transConv1 alone', 128, 64, 0, 1, 1, 0, 0, 1),
                                                                     ('128->64 no mask / bias / relu', 128, 64, 0, 0, 0, 0, 0, 0), ('32->64 (one K step), plain', 32, 64, 0, 0, 0, 0, 0, 0),
                                                                     ('64->64 (two K steps), plain', 64, 64, 0, 0, 0, 0, 0, 0), ('64->32 plain', 64, 32, 0, 0, 0, 0, 0, 0),
                                                                     ('32->32 plain', 32, 32, 0, 0, 0, 0, 0, 0)):
    w_img, w2_img = M.pack_fs2(torch.randn(3, 3, co, ci) / (ci * 2.25) ** 0.5, torch.randn(co, ci2) / ci2 ** 0.5 if ci2 else None)
    w_img = w_img.to(DEV)
    w2_img = w2_img.to(DEV) if ci2 else None
    x = torch.randn(B, H, W, ci, device=DEV).half()
    x2 = torch.randn(B, 2 * H, 2 * W, ci2, device=DEV).half() if ci2 else None
    out = torch.zeros(B, 2 * H, 2 * W, co, device=DEV, dtype=torch.float16)
    bias = torch.randn(co, device=DEV) if use_bias else None
    add = torch.randn(B, 2 * H, 2 * W, co, device=DEV).half() if use_add else None
    gate = torch.randint(0, 16, (B, 2 * H, 2 * W, co // 4), device=DEV, dtype=torch.uint8) if use_gate else None
    mask = torch.zeros(B, 2 * H, 2 * W, co // 4, device=DEV, dtype=torch.uint8) if use_mask else None

    def run():
        _lib.call('spaa_fs2_h16', _lib.hptr(x), ci, ci, _lib.hptr(w_img), _lib.hptr(x2) if ci2 else None, ci2, ci2, _lib.hptr(w2_img) if ci2 else None,
                  _lib.ptr(bias) if use_bias else None, _lib.hptr(add) if use_add else None, _lib.ptr(gate) if use_gate else None, relu, _lib.hptr(out),
                  _lib.ptr(mask) if use_mask else None, co, B, H, W)
    us = timeit(run)
    gf = 2 * B * H * W * (9 * ci * co + 4 * ci2 * co) / 1e9
    mb = (x.numel() + (x2.numel() if ci2 else 0) + out.numel() + (add.numel() if use_add else 0)) * 2 / 1e6
    print(f'{name:24s} {us:7.1f} us  {gf / us * 1e-3:7.1f} TF  {mb / us * 1e-6 * 1e6 / 1e6:6.2f} TB/s ({mb:.0f} MB)', flush=True)


print('--- csrc/s2f_h16.hip: stride-2 forward forms')
for name, ci, co, hi, use_add, use_gate, use_mask, relu in (('conv2 (+res2_s, ReLU, mask)', 32, 64, 128, 1, 0, 1, 1), ('conv2_s (ReLU, mask)', 32, 64, 128, 0, 0, 1, 1),
                                                            ('transConv1^T (gate)', 64, 128, 128, 0, 1, 0, 0), ('32->64 plain', 32, 64, 128, 0, 0, 0, 0),
                                                            ('64->128 plain', 64, 128, 128, 0, 0, 0, 0)):
    w_img = M.pack_s2f(torch.randn(3, 3, co, ci) / (ci * 9) ** 0.5).to(DEV)
    x = torch.randn(B, hi, hi, ci, device=DEV).half()
    ho = hi // 2
    out = torch.zeros(B, ho, ho, co, device=DEV, dtype=torch.float16)
    bias = torch.randn(co, device=DEV)
    add = torch.randn(B, ho, ho, co, device=DEV).half() if use_add else None
    gate = torch.randint(0, 16, (B, ho, ho, co // 4), device=DEV, dtype=torch.uint8) if use_gate else None
    mask = torch.zeros(B, ho, ho, co // 4, device=DEV, dtype=torch.uint8) if use_mask else None

    def run():
        _lib.call('spaa_s2f_h16', _lib.hptr(x), ci, ci, _lib.hptr(w_img), _lib.ptr(bias), _lib.hptr(add) if use_add else None,
                  _lib.ptr(gate) if use_gate else None, relu, _lib.hptr(out), _lib.ptr(mask) if use_mask else None, co, B, hi, hi)
    us = timeit(run)
    mb = (x.numel() + out.numel() + (add.numel() if use_add else 0)) * 2 / 1e6
    print(f'{name:28s} {us:7.1f} us  ({mb:.0f} MB)', flush=True)

print('--- csrc/s2f_x6.hip: fp32 stride-2 forward (bf16x6)')
for name, use_add, use_mask, relu in (('conv2 (+res2_s, ReLU, mask)', 1, 1, 1), ('conv2_s (ReLU, mask)', 0, 1, 1), ('32->64 plain', 0, 0, 0)):
    ci, co, hi = 32, 64, 128
    w_img = M.pack_s2f_x6(torch.randn(3, 3, co, ci) / (ci * 9) ** 0.5).to(DEV)
    x = torch.randn(B, hi, hi, ci, device=DEV)
    ho = hi // 2
    out = torch.zeros(B, ho, ho, co, device=DEV)
    bias = torch.randn(co, device=DEV)
    add = torch.randn(B, ho, ho, co, device=DEV) if use_add else None
    mask = torch.zeros(B, ho, ho, co // 4, device=DEV, dtype=torch.uint8) if use_mask else None

    def run():
        _lib.call('spaa_s2f_x6', _lib.ptr(x), ci, ci, M.C_ptr(w_img), _lib.ptr(bias), _lib.ptr(add) if use_add else None, None, relu, _lib.ptr(out),
                  _lib.ptr(mask) if use_mask else None, co, B, hi, hi)
    us = timeit(run)
    mb = (x.numel() + out.numel() + (add.numel() if use_add else 0)) * 4 / 1e6
    print(f'{name:28s} {us:7.1f} us  ({mb:.0f} MB)', flush=True)
