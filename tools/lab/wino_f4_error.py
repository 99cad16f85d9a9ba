"""Bounded experiment: would Winograd F(4x4,3x3) keep conv4's accuracy?  CPU emulation of the arithmetic the
kernel would run -- filter transform in fp64 rounded to fp32 (host side, as for F(2x2)), data transform, channel sum and output
transform in fp32 -- against fp64, next to F(2x2,3x3) and the direct fp32 sum on the same data (conv4: 128 -> 256 channels,
post-ReLU input, weights ~ N(0, 1/K)).  Kill criterion part 1: error <= 3x the direct kernel's.
python tools/lab/wino_f4_error.py > profiles/r04_wino_experiments.txt"""
import numpy as np, torch
torch.manual_seed(0)
torch.set_num_threads(8)
C, N, H = 128, 256, 32
x = torch.relu(torch.randn(2, C, H + 2, H + 2, dtype=torch.float64))      # (already padded)
w = torch.randn(N, C, 3, 3, dtype=torch.float64) / np.sqrt(9 * C)
ref = torch.nn.functional.conv2d(x, w)


def rel_inf(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


def rel_l2(a, b):
    return float((a.double() - b).norm() / b.norm())


def wino(m, BT, G, AT):
    """F(m x m, 3 x 3): tiles of (m + 2)^2 inputs -> m^2 outputs; fp32 except the filter transform (fp64 -> fp32)."""
    a = m + 2
    U = torch.einsum('ij,ncjk,lk->ncil', G, w, G).float()                    # [N, C, a, a]
    BTf, ATf = BT.float(), AT.float()
    xf = x.float()
    nt = H // m
    d = xf.unfold(2, a, m).unfold(3, a, m)                                   # [B, C, nt, nt, a, a]
    V = torch.einsum('ij,bcyxjk,lk->bcyxil', BTf, d, BTf)                    # data transform, fp32
    M = torch.einsum('ncil,bcyxil->bnyxil', U, V)                            # channel sum, fp32 accumulate
    Y = torch.einsum('ij,bnyxjk,lk->bnyxil', ATf, M, ATf)                    # [B, N, nt, nt, m, m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(2, N, H, H)


# F(2x2,3x3) (the kernel's matrices) and F(4x4,3x3) (Lavin & Gray, points 0, +-1, +-2, inf)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                   [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)

direct = torch.nn.functional.conv2d(x.float(), w.float())
f2 = wino(2, BT2, G2, AT2)
f4 = wino(4, BT4, G4, AT4)
assert rel_inf(wino(2, BT2, G2, AT2).double() * 0 + f2, ref) < 1e-5 and rel_inf(f4, ref) < 1e-3   # (the matrices are right)
print('conv4 (128 -> 256, 3x3) on post-ReLU input, error against fp64: relative L-inf / relative L2')
print(f'  direct fp32 sum        {rel_inf(direct, ref):.2e} / {rel_l2(direct, ref):.2e}')
print(f'  F(2x2,3x3) in fp32     {rel_inf(f2, ref):.2e} / {rel_l2(f2, ref):.2e}   (x{rel_l2(f2, ref) / rel_l2(direct, ref):.1f} the direct sum in L2)')
print(f'  F(4x4,3x3) in fp32     {rel_inf(f4, ref):.2e} / {rel_l2(f4, ref):.2e}   (x{rel_l2(f4, ref) / rel_l2(direct, ref):.1f} the direct sum in L2)')
ok = rel_inf(f4, ref) <= 3 * rel_inf(direct, ref)
print(f'kill criterion part 1 (F(4x4) error <= 3x the direct kernel\'s): {"met" if ok else "NOT met"}')
