"""What a plain streaming kernel reaches on this box (torch elementwise kernels over 1-4 GB tensors: far larger than the 256 MB MALL):
read-only (sum), copy (read + write), in-place add (read + write of the same lines) -- the practical ceiling behind the 8 TB/s peak
that the `roofline` fractions of bench.py are quoted against."""
import torch
DEV = torch.device('cuda:0')
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for gb in (1, 4):
    n = gb * (1 << 30) // 4
    x = torch.ones(n, device=DEV); y = torch.empty_like(x)
    s = t(lambda: x.sum())
    c = t(lambda: y.copy_(x))
    a = t(lambda: x.add_(1.0))
    h = x.half()
    sh = t(lambda: h.float().sum()) if gb == 1 else 0
    print(f'{gb} GB fp32: read-only sum {gb * 1.0737e-3 / s:.2f} TB/s; copy {2 * gb * 1.0737e-3 / c:.2f} TB/s (read + write); in-place add {2 * gb * 1.0737e-3 / a:.2f} TB/s', flush=True)
    del x, y, h
