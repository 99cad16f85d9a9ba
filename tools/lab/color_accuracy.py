"""Lab / dE2000 / loss-gradient errors of the colour kernels against the reference's fixture (tests/golden/color_kat.npz) -- what
test_color_kernels_vs_reference_golden bounds, printed: run once on the default build (hardware transcendentals) and once after
`make -C spaa_amd/csrc -B color.o COLOR_LIBM=1 && make -C spaa_amd/csrc` (libm forms)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np, torch
from spaa_amd import differential_color_functions as dcf
DEV = torch.device('cuda:0')
z = np.load(os.path.join(R, 'tests', 'golden', 'color_kat.npz'))
a, b = torch.from_numpy(z['rgb_a']).to(DEV), torch.from_numpy(z['rgb_b']).to(DEV)
la, lb = dcf.rgb2lab_diff(a), dcf.rgb2lab_diff(b)
rel = lambda x, y: float((x.cpu() - y).abs().max() / y.abs().max())
print(f'Lab rel Linf: {rel(la, torch.from_numpy(z["lab_a"])):.2e} / {rel(lb, torch.from_numpy(z["lab_b"])):.2e}')
de = dcf.ciede2000_diff(la, lb)
print(f'dE2000 max abs error: {float((de.cpu() - torch.from_numpy(z["de"])).abs().max()):.2e} (dE up to {float(z["de"].max()):.1f})')
l2, dE, g = dcf.stealth_loss_with_grad(a, b, 0.0, 1.0)
g = g.cpu() * (a.shape[2] * a.shape[3])
g_ref = torch.from_numpy(z['grad_a'])
fin = torch.isfinite(g_ref)
chroma = torch.minimum(torch.from_numpy(z['lab_a'])[:, 1:].norm(dim=1), torch.from_numpy(z['lab_b'])[:, 1:].norm(dim=1))
well = (chroma > 1.0)[:, None].expand_as(g_ref) & fin
scale = g_ref[fin].abs().max()
print(f'loss gradient / max |gradient|: well-conditioned pixels {float((g - g_ref)[well].abs().max() / scale):.2e}, all finite {float((g - g_ref)[fin].abs().max() / scale):.2e}')
print(f'mean dE rel error: {float(np.abs(dE.cpu().numpy() - z["de"].mean(axis=(1, 2))).max() / z["de"].mean()):.2e}')
