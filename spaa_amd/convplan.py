"""Host-side planning for the tap-list convolution kernel (csrc/tapconv.hip).

Every convolution-like op on the SPAA hot path — nn.Conv2d forward, nn.ConvTranspose2d forward and the
input-gradient (`aten::convolution_backward`) of both — is expressed as

    out[b, oy0 + s_out*y, ox0 + s_out*x, n] = sum_t sum_c in[b, s_in*y + dy_t, s_in*x + dx_t, c] * W_t[n, c]

so one MFMA kernel serves all of them.  The weights are frozen during an attack
(/root/reference/src/python/projector_based_attack.py:62-67), so the packing below runs once per model.

This module only rearranges weights (layout plumbing, on the host); the arithmetic happens in the HIP kernel.
"""
import ctypes as C
import os

import torch

from . import _lib

BK = 32
NPAD = 128
PROFILE = None  # set to a list by bench.py to time every tapconv launch with HIP events
PROFILE_ONLY = None  # ... or only the launches of these reported tile ids (events inside bench.py's timed region)
FORCE_TILE = 0  # tools/autotune.py: force one workgroup tile for every launch
TILE_NAMES = {1: '128x128', 2: '256x64', 3: '256x32', 4: '128x64a', 5: '128x32', 6: '64x64', 7: '64x128', 8: '128x64b',
              9: 'direct4', 10: 'direct32', 11: 'thin4', 12: 'x6_64x64', 13: 'x6_128x32', 14: 'x6_32x128',
              15: 'x6v2_128x64g3', 16: 'x6v2_128x64g2', 17: 'x6v2_128x128g1', 18: 'x6v2_64x64g3', 19: 'x6v2_64x128g2',
              20: 'x6v3_128x64g3', 21: 'x6v3_128x64g2', 22: 'x6v3_64x64g3', 23: 'x6v3_128x128g1', 24: 'x6v3_64x128g2',
              25: 'x6d_128x128', 26: 'x6d_256x128', 27: 'x6d_128x64', 28: 'thinpatch32', 29: 'thinpatch16',
              30: 'x6d_128x32', 31: 'x6d_64x64', 32: 'x6d_64x128', 33: 'x6d_256x64',
              34: 'x6d16_128x128', 35: 'x6d16_256x128', 36: 'x6d16_128x64', 37: 'x6d16_128x32', 38: 'smallcin',
              39: 'x6d16co_128x128', 40: 'x6d16co_128x64', 41: 'x6d16co_128x32',
              42: 'x6d16a3_128x64', 43: 'x6d16a3_128x32', 44: 'x6da3_128x64', 45: 'x6d16coa3_128x64', 46: 'x6d16coa3_128x32', 47: 'thinpatch16x2',
              48: 'x6d16p_128x128', 49: 'x6d16p_128x64', 50: 'x6d16a3p_128x64', 51: 'x6da3p_128x64', 52: 'x6d16p_256x128',
              53: 'x6d16p_128x32', 54: 'x6dp_128x128',
              60: 'h16_128x128', 61: 'h16_128x64', 62: 'h16_128x32', 63: 'h16_128x16', 64: 'h16_256x128', 65: 'h16_256x256', 68: 'h16p_16x32x128', 72: 'thinmf_12x32',
              73: 'wino_x6_8x32x64', 74: 'x6p_4x32', 76: 'c3conv_16x32',
              70: 'wino_x6_16x32x128', 71: 'wino_x6_16x32x64'}   # (70: the launcher chooses the N tile -- reported as 71 when it took 64; 71: 64-wide forced)
X6D_TILES = set(range(25, 28)) | set(range(30, 38)) | set(range(39, 47)) | set(range(48, 55))   # DMA-staged bf16x6 kernels (csrc/tapconv_x6d.hip)
X6D_PERSISTENT = set(range(48, 55))      # ... of which the persistent ones (stream-K capable)
H16_TILES = set(range(60, 66))           # fp16 implicit-GEMM kernels (csrc/tapconv_h16.hip); 68 = patch-staged 3x3 (tapconv_h16p.hip)
STORE4_TILES = set(range(15, 28)) | set(range(30, 47)) | set(range(48, 55)) | set(range(60, 66)) | {68, 70, 71, 73, 74, 76}  # shared epilogue (epilogue.hpp)
F16OUT_TILES = set(range(15, 25)) | {38, 76} | set(range(60, 66))  # ... of which these may write fp16 (fp32 image in, fp16 activation out)
DEFAULT_DISABLE = set(os.environ.get('SPAA_DEFAULT_DISABLE', '').split(','))
X6P_STD = os.environ.get('SPAA_X6P_STD', '1') != '0'   # 0: the stride-2 patch kernel's run-time schedule (A/B measurements)
DEBUG_TAPMAJOR = int(os.environ.get('SPAA_X6D_TAPMAJOR', '0'))      # 1: tap-major K order (A/B measurements only)
DEBUG_PERSIST_CAP = int(os.environ.get('SPAA_X6D_PERSIST_CAP', '0'))  # > 0: persistent launches use this many workgroups
FORCE_KSPLIT = int(os.environ.get('SPAA_FORCE_KSPLIT', '0'))        # split-K factor of the fp16 implicit-GEMM kernel (A/B runs, tests)
DEBUG_WINO = int(os.environ.get('SPAA_WINO_DBG', '0'))              # timing experiments of the Winograd kernel
DEBUG_WINO_NOCANVAS = int(os.environ.get('SPAA_WINO_NOCANVAS', '0'))  # 1: small images keep the image-aligned workgroup regions (A/B); 2: canvas wherever it has fewer regions (tests)
WINO_SPLITK = os.environ.get('SPAA_WINO_SPLITK', '1') != '0'        # Winograd layers with few workgroups and long K: K ranges + ordered second pass
# ... with the second pass INSIDE the kernel (the last-arriving workgroup of a tile adds the K ranges; asked for in round 5's review).  Built,
# bitwise the two-pass form -- and SLOWER (profiles/r06_splitk_fixup.txt: 8.94 ms per step against 8.26 with agent-scope stores / loads,
# 9.61 with __threadfence's L2 write-back; fp16 storage 4.34 against 4.13): a layer with few tiles and many K ranges (ResNet layer4: 32 tiles
# x 8 ranges) leaves its whole reduction to 32 workgroups, 2 MB of uncached reads each, where the separate launch spreads it over the chip in
# ~8 us.  Off by default; the tests run both forms
WINO_SPLITK_FIXUP = os.environ.get('SPAA_SPLITK_FIXUP', '0') == '1'
SPLITK_HDR = 4096   # include/spaa_hip.h: SPAA_SPLITK_HDR_FLOATS
DEBUG_THINMF = int(os.environ.get('SPAA_THINMF_DBG', '0'))           # timing experiments of csrc/tapconv_thinmf.hip (builds with -DSPAA_THINMF_ABLATE)
DEBUG_SMALLCIN_NOSLAB = int(os.environ.get('SPAA_SMALLCIN_NOSLAB', '0'))  # 1: stride-2 smallcin layers store from the MFMA layout (A/B)
H16P_CV = tuple(int(v) for v in os.environ.get('SPAA_H16P_CV', '0,0,0').split(','))   # (N tile 0 = chosen / 64 / 128, K ranges 0 = chosen, 1 = canvases wherever they have fewer regions: tests) of the patch-staged fp16 kernel's canvas / K-range form (A/B runs)
H16P_LEAN_WIDE = int(os.environ.get('SPAA_H16P_LEAN_WIDE', '2'))   # 64-wide two-workgroup form of the patch-staged fp16 kernel for wider layers: 0 never, 1 always, 2 by shape
DEBUG_H16_2STAGE = int(os.environ.get('SPAA_H16_2STAGE', '0'))      # 1: the fp16 implicit-GEMM kernel never takes its four-stage form (A/B measurements)
FOLD_K3S2 = os.environ.get('SPAA_FOLD_K3S2', '1') != '0'   # 3x3 / s2 input gradients with few output channels: classes folded
FOLD_K3S2_MAX_COUT = 32
FOLD_DECONV = True  # k2/s2 transposed convs: parity classes folded into GEMM rows (one read of the input)
ENABLE_X6 = True  # build the split-bf16 weight planes (needed by tiles 12-14)
WINOGRAD = os.environ.get('SPAA_WINOGRAD', '1') != '0'  # 3x3/s1 layers: allow the Winograd F(2x2,3x3) kernel (tile 70)


def masked_any(*m):
    return any(x is not None for x in m)


def _load_tune():
    import json
    import os
    path = os.environ.get('SPAA_TUNE_FILE') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tapconv_tune.json')
    if os.path.exists(path):
        with open(path) as fh:
            return {k: int(v) for k, v in json.load(fh).items()}
    return {}


TUNE = _load_tune()  # shape key -> tile id, measured on MI355X by tools/autotune.py (absent key = heuristic)
_NEAREST = {}


def tuned_tile(key):
    """Tile for a layer-shape key `Cin_Cout_taps_sin_sout_M[_fold]`: the measured one, else the measured choice of the SAME layer
    at the nearest pixel count M (ratio at most 4 either way: batch 10 borrows from batch 8 or 16, not from batch 64's 256-row
    persistent tiles), else -1 (the heuristic of `_default_tile`)."""
    t = TUNE.get(key)
    if t is not None:
        return t
    if key in _NEAREST:
        return _NEAREST[key]
    parts = key.split('_')
    fold = parts[-1] == 'fold'
    m = int(parts[5])
    stem = '_'.join(parts[:5]) + '_'
    best, best_r = -1, 4.0001
    for k, v in TUNE.items():
        if not k.startswith(stem) or k.endswith('_fold') != fold:
            continue
        mk = int(k.split('_')[5])
        r = max(mk, m) / max(1, min(mk, m))
        if r < best_r:
            best, best_r = v, r
    _NEAREST[key] = best
    return best


def tune_report():
    """Layer shapes of this process whose kernel was NOT a measured entry of the tune table: borrowed from the same layer at the
    nearest pixel count, or left to the `_default_tile` rules (bench.py reports both counts next to the headline)."""
    return {'borrowed': sorted(k for k, v in _NEAREST.items() if v >= 0), 'untuned': sorted(k for k, v in _NEAREST.items() if v < 0)}


_STREAMK_WS = {}


def _streamk_workspace(device):
    """2 slots x 768 workgroups x (128 x 128) floats (include/spaa_hip.h: ksplit == -1)."""
    key = str(device)
    if key not in _STREAMK_WS:
        _STREAMK_WS[key] = torch.empty(2 * 768 * 128 * 128, device=device, dtype=torch.float32)
    return _STREAMK_WS[key]


def _ceil(a, b):
    return (a + b - 1) // b * b


class TapClassSpec:
    """One output-parity class: output offset and a list of (dy, dx, W[n, c]) taps."""

    def __init__(self, oy0, ox0):
        self.oy0, self.ox0, self.taps = oy0, ox0, []

    def add(self, dy, dx, w):
        self.taps.append((int(dy), int(dx), w))


class ConvPlan:
    """Packed weights + launch template for spaa_tapconv_f32."""

    def __init__(self, classes, cin, cout, s_in, s_out, bias=None, device='cuda', name='', nfold=1, kpad_extra=0):
        assert 1 <= len(classes) <= _lib.MAX_CLASSES
        assert nfold == 1 or (nfold == 4 and len(classes) == 1 and s_out == 2 and cout % 4 == 0 and cin % 32 == 0)
        self.nfold = nfold                  # 4: the parity classes of a k2/s2 deconv folded into GEMM rows
        self.name = name
        self.cin = cin                      # logical input channels (before padding)
        self.cin_p = _ceil(cin, 4)          # K uses the padded count; pad columns are zero
        self.cout = cout
        self.s_in, self.s_out = s_in, s_out
        ngemm = cout * nfold
        npad = _ceil(ngemm, NPAD)
        w_chunks, tap_list, self.cls = [], [], []
        w_off = 0
        for c in classes:
            nt = len(c.taps)
            assert nt <= _lib.MAX_TAPS
            k = nt * self.cin_p
            kpad = _ceil(k, BK) + kpad_extra   # (extra: de-tune the row stride from the L2 channel interleave)
            wp = torch.zeros(npad, kpad, dtype=torch.float32)
            for t, (dy, dx, w) in enumerate(c.taps):
                assert w.shape == (ngemm, cin), (w.shape, ngemm, cin)
                wp[:ngemm, t * self.cin_p:t * self.cin_p + cin] = w
            self.cls.append(dict(oy0=c.oy0, ox0=c.ox0, ntaps=nt, tap_off=len(tap_list), K=k, Kpad=kpad, w_off=w_off))
            tap_list += [(dy, dx) for dy, dx, _ in c.taps]
            w_chunks.append(wp.reshape(-1))
            w_off += npad * kpad
        self.classes_host = classes
        self.weights = torch.cat(w_chunks).to(device) if w_off > 0 else torch.zeros(4, device=device)
        # the same weights as three bf16 planes with w == h + m + l exactly (tapconv_x6.hip)
        self.w_split = None
        if ENABLE_X6 and w_off > 0:
            parts = []
            for wp in w_chunks:
                h = wp.to(torch.bfloat16)
                r1 = wp - h.float()
                m = r1.to(torch.bfloat16)
                lo = (r1 - m.float()).to(torch.bfloat16)
                parts.append(torch.stack([h, m, lo]).view(torch.int16).reshape(-1))
            self.w_split = torch.cat(parts).to(device)
        tl = tap_list if tap_list else [(0, 0)]
        self.tap_range = (min(t[0] for t in tl), max(t[0] for t in tl), min(t[1] for t in tl), max(t[1] for t in tl))
        taps = torch.tensor(tl, dtype=torch.int32).reshape(-1)
        self.taps = taps.to(device)
        self.bias = bias.detach().float().contiguous().to(device) if bias is not None else None
        # algorithmic FLOPs (2*MAC, logical channels, no padding) per pixel of the class grid
        self.flops_per_pixel = 2 * sum(c['ntaps'] for c in self.cls) * cin * ngemm
        self.ntaps_total = sum(c['ntaps'] for c in self.cls)
        self._ws = None  # split-K workspace, allocated on first use
        self._ws_fix = None  # ... with the arrival-counter header (K ranges combined inside the kernel)
        self._npad, self._dev = npad, device
        self.w_half = None  # fp16 plane for the fp16-storage kernels, packed on first use (half_plane())
        self.wino = None    # the same layer in Winograd F(2x2,3x3) form (attach_winograd), run as tile 70
        self.fixed_tile = 0
        self._wino_plans = {}   # spaa_tapconv_wino_plan results per launch shape (run)
        self.alg_taps = self.ntaps_total
        self._thin = {}     # folded weight layouts of the thin-output matrix-core kernel (thin_fold), packed on first use

    def thin_ok(self):
        """Thin output on the matrix cores (csrc/tapconv_thinmf.hip, tile 72): at most 4 output channels, stride-1 input sampling,
        the S x S output-parity classes in row-major order, a tap box of at most 4 x 4."""
        s = self.s_out
        if not (self.cout <= 4 and self.cin_p % 32 == 0 and self.s_in == 1 and self.nfold == 1 and s in (1, 2) and len(self.cls) == s * s):
            return False
        if any((c['oy0'], c['ox0']) != (i // s, i % s) for i, c in enumerate(self.cls)):
            return False
        dy0, dy1, dx0, dx1 = self.tap_range
        return dy1 - dy0 + 1 <= 4 and dx1 - dx0 + 1 <= 4

    def x6p_ok(self):
        """Stride-2 fractional layer for the patch-staged bf16x6 kernel (csrc/tapconv_x6p.hip, tile 74): four output-parity classes in
        row-major order, every tap inside one 2 x 2 window, stride-1 input sampling, Cin % 32 == 0, unfolded."""
        if not (len(self.cls) == 4 and self.s_in == 1 and self.s_out == 2 and self.nfold == 1 and self.cin_p % 32 == 0
                and self.cin == self.cin_p and self.w_split is not None):
            return False
        if any((c['oy0'], c['ox0']) != (i // 2, i % 2) or not 1 <= c['ntaps'] <= 4 or c['Kpad'] != c['K'] for i, c in enumerate(self.cls)):
            return False
        dy0, dy1, dx0, dx1 = self.tap_range
        if not (dy1 - dy0 == 1 and dx1 - dx0 == 1):
            return False
        # (the kernel fetches the next window position's pixels under the products of class (1, 1): it must have a tap at each)
        return sorted((dy, dx) for dy, dx, _ in self.classes_host[3].taps) == [(dy0, dx0), (dy0, dx1), (dy1, dx0), (dy1, dx1)]

    def x6p_canonical(self):
        """The (class, tap) structure csrc/tapconv_x6p.hip has compiled in (its STD instantiation): tap window (0..1) x (0..1), class
        (0,0) = [(0,0)], (0,1) = [(0,1), (0,0)], (1,0) = [(1,0), (0,0)], (1,1) = [(1,1), (1,0), (0,1), (0,0)] -- what a k3 / s2 / p1
        transposed convolution and the input gradient of a k3 / s2 / p1 convolution give."""
        want = [[(0, 0)], [(0, 1), (0, 0)], [(1, 0), (0, 0)], [(1, 1), (1, 0), (0, 1), (0, 0)]]
        return (self.x6p_ok() and tuple(self.tap_range) == (0, 1, 0, 1)
                and [[(dy, dx) for dy, dx, _ in c.taps] for c in self.classes_host] == want)

    def attach_second_source(self, weight2, bias2=None):
        """Fuse a 1 x 1 convolution of a tensor at OUTPUT resolution into this stride-2 layer (tile 74): `weight2` [Cout, Cin2] (Cin2 =
        32 or 64).  `run(..., inp2=...)` then adds it before bias / residual / activation; the bias becomes the sum of both."""
        assert self.x6p_ok()
        w2 = weight2.detach().float().reshape(weight2.shape[0], -1).cpu()
        assert w2.shape[0] == self.cout and w2.shape[1] in (32, 64)
        wp = torch.zeros(self._npad, w2.shape[1])
        wp[:self.cout] = w2
        self.w2_split = split_planes(wp).reshape(-1).to(self._dev)
        self.cin2 = w2.shape[1]
        b = self.bias.clone() if self.bias is not None else torch.zeros(self.cout, device=self._dev)
        if bias2 is not None:
            b = b + bias2.detach().float().to(b.device)
        self.bias2 = b          # bias of the fused launch
        return self

    def attach_second_source_h16(self, weight2, bias2=None):
        """fp16 storage: the same fusion for a FOLDED stride-2 transposed layer on the patch-staged fp16 kernel (tile 68, nfold = 4):
        `weight2` [Cout, Cin2] rounded to fp16 (as `w_half`).  `run(..., inp2=...)` (fp16 tensors) then adds the 1 x 1 convolution."""
        assert self.nfold == 4 and len(self.cls) == 1 and self.cout % 16 == 0
        w2 = weight2.detach().float().reshape(weight2.shape[0], -1).cpu()
        assert w2.shape[0] == self.cout and w2.shape[1] in (32, 64)
        self.w2_half = w2.half().contiguous().to(self._dev)
        self.cin2 = w2.shape[1]
        b = self.bias.clone() if self.bias is not None else torch.zeros(self.cout, device=self._dev)
        if bias2 is not None:
            b = b + bias2.detach().float().to(b.device)
        self.bias2 = b
        return self

    def c3_ok(self):
        """First layer of a network for csrc/tapconv_c3.hip (tile 76): a convolution over a 3-channel image (NHWC4, zero lane), one
        class, stride 1 or 2, at most 64 output channels, 3 x taps products per output within one or five 32-deep steps."""
        return (len(self.cls) == 1 and self.cin == 3 and self.cin_p == 4 and self.s_out == 1 and self.s_in in (1, 2) and self.cout <= 64
                and self.nfold == 1 and 3 * self.ntaps_total <= 160)

    def c3_pack(self, half=False):
        """The weights as tile 76 stages them: K = (tap, channel of 3) products only, [NK][3 bf16 planes: w == h + m + l][BN rows][32],
        16-byte chunk c of row n stored at chunk c ^ ((n >> 3 & 1) << 1) (conflict-free fragment reads); packed on first use.
        `half` (fp16-storage mode): ONE plane of the weights rounded to fp16, same rows and swizzle."""
        if half:
            if getattr(self, '_c3h', None) is None:
                nt, c = self.ntaps_total, self.cls[0]
                nk, bn = (1 if 3 * nt <= 32 else 5), (32 if self.cout <= 32 else 64)
                w = self.weights[c['w_off']:c['w_off'] + self._npad * c['Kpad']].view(self._npad, c['Kpad'])[:bn, :4 * nt]
                w = w.reshape(bn, nt, 4)[:, :, :3].reshape(bn, 3 * nt)
                w = torch.nn.functional.pad(w, (0, 32 * nk - 3 * nt)).float().cpu().half().view(torch.int16)
                pl = w.view(bn, nk, 4, 8).permute(1, 0, 2, 3).contiguous()                              # [nk][row][chunk][8]
                n = torch.arange(bn)
                idx = (torch.arange(4)[None, :] ^ (((n >> 3) & 1) << 1)[:, None])
                pl = torch.gather(pl, 2, idx[None, :, :, None].expand(nk, bn, 4, 8))
                self._c3h = pl.contiguous().reshape(-1).to(self._dev)
            return self._c3h
        if getattr(self, '_c3', None) is None:
            nt, c = self.ntaps_total, self.cls[0]
            nk, bn = (1 if 3 * nt <= 32 else 5), (32 if self.cout <= 32 else 64)
            w = self.weights[c['w_off']:c['w_off'] + self._npad * c['Kpad']].view(self._npad, c['Kpad'])[:bn, :4 * nt]
            w = w.reshape(bn, nt, 4)[:, :, :3].reshape(bn, 3 * nt)
            w = torch.nn.functional.pad(w, (0, 32 * nk - 3 * nt)).float().cpu()
            pl = split_planes(w).view(3, bn, nk, 4, 8).permute(2, 0, 1, 3, 4).contiguous()        # [nk][plane][row][chunk][8]
            n = torch.arange(bn)
            idx = (torch.arange(4)[None, :] ^ (((n >> 3) & 1) << 1)[:, None])                       # logical chunk held by each slot
            pl = torch.gather(pl, 3, idx[None, None, :, :, None].expand(nk, 3, bn, 4, 8))
            self._c3 = pl.contiguous().reshape(-1).to(self._dev)
        return self._c3

    def thin_fold(self, half):
        """The weights in the folded layout of the thin-output matrix-core kernel: GEMM rows = class * 4 + channel (16 rows; rows of
        taps a class does not have, of absent channels and classes: zero), [channel block][tap column][plane][tap row][16][32
        channels], 16-byte chunks of a 64-byte row swapped in pairs for rows 8-15 (conflict-free fragment reads: swz64).  Planes:
        fp16 (`half`) or three bf16 with w == h + m + l exactly."""
        key = 'h' if half else 's'
        if key not in self._thin:
            s, dev = self.s_out, self.weights.device
            dy0, dy1, dx0, dx1 = self.tap_range
            tbh, tbw, nkb = max(dy1 - dy0 + 1, 2), dx1 - dx0 + 1, self.cin_p // 32
            wf = torch.zeros(16, tbh, tbw, self.cin_p, device=dev)
            for i, (c, spec) in enumerate(zip(self.cls, self.classes_host)):
                wp = self.weights[c['w_off']:c['w_off'] + self._npad * c['Kpad']].view(self._npad, c['Kpad'])
                for t, (dy, dx, _w) in enumerate(spec.taps):
                    wf[4 * i:4 * i + self.cout, dy - dy0, dx - dx0] = wp[:self.cout, t * self.cin_p:(t + 1) * self.cin_p]
            w = wf.view(16, tbh, tbw, nkb, 4, 8).permute(3, 2, 1, 0, 4, 5).contiguous()   # [kb][dxi][dyi][row][chunk][8]
            sw = w.clone()
            sw[..., 8:, 0, :], sw[..., 8:, 2, :] = w[..., 8:, 2, :], w[..., 8:, 0, :]
            sw[..., 8:, 1, :], sw[..., 8:, 3, :] = w[..., 8:, 3, :], w[..., 8:, 1, :]
            sw = sw.reshape(nkb, tbw, tbh, 16, 32)
            if half:
                planes = sw.half().unsqueeze(2).view(torch.int16)
            else:
                h = sw.to(torch.bfloat16)
                r1 = sw - h.float()
                m = r1.to(torch.bfloat16)
                lo = (r1 - m.float()).to(torch.bfloat16)
                planes = torch.stack([h, m, lo], dim=2).view(torch.int16)               # [kb][dxi][plane][dyi][16][32]
            self._thin[key] = planes.contiguous().reshape(-1)
        return self._thin[key]

    def half_plane(self):
        """The weights rounded to fp16, per class [Npad][K rounded up to 64] (zero padded), classes back to back: the
        operand of the fp16-storage kernels (csrc/tapconv_h16.hip)."""
        if self.w_half is None:   # (from the device copy of the weights: what `refresh` keeps up to date)
            parts = []
            for c in self.cls:
                k64 = _ceil(c['K'], 64)
                wp = self.weights[c['w_off']:c['w_off'] + self._npad * c['Kpad']].view(self._npad, c['Kpad'])
                wh = torch.zeros(self._npad, k64, dtype=torch.float16, device=self.weights.device)
                wh[:, :c['K']] = wp[:, :c['K']].to(torch.float16)
                parts.append(wh.reshape(-1))
            self.w_half = torch.cat(parts) if parts else torch.zeros(8, dtype=torch.float16, device=self.weights.device)
        return self.w_half

    def run(self, inp, out, add=None, gate=None, gate_mode=_lib.GATE_POS, act=_lib.ACT_NONE, aux_out=None,
            gate2=None, in_coff=0, out_coff=0, add_coff=0, gate_coff=0, mask_out=None, gate_bits=None, gate2_bits=None,
            inp2=None, in2_coff=0, _wino=None, pool_adjoint=None, pool=None, unpool=None):
        """inp: [B,Hin,Win,Cs_in], out: [B,Hout,Wout,Cs_out] NHWC float32 CUDA tensors.
        `_wino` (internal): (tile, K ranges) of this launch when the plan is the Winograd form of another plan.
        `mask_out` / `gate_bits` / `gate2_bits`: uint8 [B,Hout,Wout,Cs/4] ReLU-gate masks (one byte per 4 channels,
        include/spaa_hip.h): written for this launch's output resp. read instead of a float `gate` / `gate2`.
        `pool_adjoint` = (arg-max bytes uint8 [B,Hp,Wp,C], (Hin, Win), relu_gate): `inp` is then the gradient w.r.t. the OUTPUT of a
        3 x 3 / stride 2 / padding 1 max-pool [B,Hp,Wp,C] whose input (Hin x Win) is this layer's input: the pool's adjoint runs as the
        prologue of the thin-output matrix-core kernel (tile 72, fp32) instead of as a launch of its own.
        `pool` = (pooled [B,Hout/2,Wout/2,C], arg-max bytes uint8 [B,Hout/2,Wout/2,C], want_arg): this layer's ReLU is followed by a 2 x 2 /
        stride-2 max-pool (torchvision VGG-16): where the patch-staged fp16 kernel serves the layer (tile 68, image-aligned regions) the pool
        runs in its epilogue and `out` is NOT written; anywhere else the layer runs as usual and spaa_maxpool_fwd follows -- the same pooled
        values and arg-max bytes either way.
        `unpool` = (arg-max bytes uint8 [B,Hin/2,Win/2,C], full-size gradient buffer [B,Hin,Win,Cs]): `inp` is the gradient w.r.t. the OUTPUT of
        the 2 x 2 / stride-2 max-pool that followed this layer's input (an input-gradient plan of torchvision VGG-16): where the
        two-workgroup form of the patch-staged fp16 kernel serves the layer the pool's adjoint (with its ReLU gate) runs as the patch
        prologue and the full-size gradient is never written; anywhere else spaa_maxpool_bwd fills the buffer first."""
        _lib.check_dev(inp, out, add, gate, aux_out, gate2, inp2, half_ok=True)
        _lib.check_mask(mask_out, gate_bits, gate2_bits)
        in_f16, out_f16 = inp.dtype == torch.float16, out.dtype == torch.float16
        for t in (add, gate, aux_out, gate2):
            if t is not None and t.dtype != out.dtype:
                raise ValueError(f'{self.name}: add / gate / aux_out / gate2 must have the storage type of `out` ({out.dtype})')
        b, hin, win, cs_in = inp.shape
        if unpool is not None:
            assert in_f16 and inp2 is None and in_coff == 0 and unpool[0].shape[:3] == inp.shape[:3] and unpool[0].shape[3] == self.cin_p == cs_in
            assert unpool[1].shape == (b, 2 * hin, 2 * win, cs_in) and unpool[1].dtype == inp.dtype
            hin, win = 2 * hin, 2 * win       # (the layer's input grid; `inp` is at the pooled resolution)
        if pool_adjoint is not None:
            parg, (hin, win), pgate = pool_adjoint
            if (in_f16 or out_f16 or inp2 is not None or parg.dtype != torch.uint8 or tuple(parg.shape) != tuple(inp.shape) or cs_in != self.cin_p
                    or in_coff or inp.shape[1:3] != ((hin - 1) // 2 + 1, (win - 1) // 2 + 1)):
                raise ValueError(f'{self.name}: pool_adjoint needs fp32 tensors, arg-max bytes of the pooled gradient\'s shape and a 3/2/1 pool geometry')
        b2, hout, wout, cs_out = out.shape
        cin2k = getattr(self, 'cin2_k', 0)   # (two-source Winograd plan: the last cin2_k input channels come from `inp2`)
        if cin2k and (inp2 is None or inp2.shape[:3] != inp.shape[:3] or inp2.dtype != inp.dtype or in2_coff + cin2k > inp2.shape[3]):
            raise ValueError(f'{self.name}: a two-source plan needs `inp2` [B, H, W, >= {cin2k} channels] of the storage type of `inp`')
        assert b == b2 and cs_in % 4 == 0 and in_coff % 4 == 0 and in_coff + self.cin_p - cin2k <= cs_in
        assert out_coff + self.cout <= cs_out
        d = _lib.TapConv()
        d.inp, d.Hin, d.Win, d.Cin, d.in_cstride, d.in_coff = inp.data_ptr(), hin, win, self.cin_p, cs_in, in_coff
        d.out, d.Hout, d.Wout, d.Cout, d.out_cstride, d.out_coff = out.data_ptr(), hout, wout, self.cout, cs_out, out_coff
        d.B = b
        if self.s_out == 1:
            d.Hm, d.Wm = hout, wout
        else:
            d.Hm, d.Wm = (hout + self.s_out - 1) // self.s_out, (wout + self.s_out - 1) // self.s_out
        d.s_in, d.s_out = self.s_in, self.s_out
        d.weights, d.taps = self.weights.data_ptr(), self.taps.data_ptr()
        d.io_dtype = (_lib.IO_IN_F16 if in_f16 else 0) | (_lib.IO_OUT_F16 if out_f16 else 0)
        if in_f16:
            if self.cin_p % 32 or any(c['Kpad'] != c['K'] for c in self.cls):
                raise ValueError(f'{self.name}: fp16-storage input needs Cin % 32 == 0 (got {self.cin_p})')
            d.w_half = self.half_plane().data_ptr()
        d.w_split = self.w_split.data_ptr() if self.w_split is not None else None
        d.bias = self.bias.data_ptr() if self.bias is not None else None
        if add is not None:
            assert add.shape[:3] == out.shape[:3] and add_coff + self.cout <= add.shape[3]
            d.add, d.add_cstride, d.add_coff = add.data_ptr(), add.shape[3], add_coff
        if gate is not None:
            assert gate.shape[:3] == out.shape[:3] and gate_coff + self.cout <= gate.shape[3]
            d.gate, d.gate_cstride, d.gate_coff, d.gate_mode = gate.data_ptr(), gate.shape[3], gate_coff, gate_mode
        d.act = act
        if aux_out is not None:
            assert aux_out.shape == out.shape
            d.aux_out = aux_out.data_ptr()
        if gate2 is not None:
            assert aux_out is not None and gate2.shape[:3] == out.shape[:3] and self.cout <= gate2.shape[3]
            d.gate2, d.gate2_cstride, d.gate2_coff = gate2.data_ptr(), gate2.shape[3], 0
        masked = mask_out is not None or gate_bits is not None or gate2_bits is not None
        if mask_out is not None:
            assert mask_out.shape == out.shape[:3] + (cs_out // 4,), (mask_out.shape, out.shape)
            d.mask_out = mask_out.data_ptr()
        if gate_bits is not None:
            assert gate is None and gate_bits.shape[:3] == out.shape[:3] and gate_coff + self.cout <= 4 * gate_bits.shape[3]
            d.gate_bits, d.gate_cstride, d.gate_coff = gate_bits.data_ptr(), 4 * gate_bits.shape[3], gate_coff
        if gate2_bits is not None:
            assert gate2 is None and aux_out is not None and gate2_bits.shape[:3] == out.shape[:3] and self.cout <= 4 * gate2_bits.shape[3]
            d.gate2_bits, d.gate2_cstride, d.gate2_coff = gate2_bits.data_ptr(), 4 * gate2_bits.shape[3], 0
        key = f'{self.cin_p}_{self.cout}_{self.alg_taps}_{self.s_in}_{self.s_out}_{b * d.Hm * d.Wm}' + ('_fold' if self.nfold > 1 else '')
        forced = FORCE_TILE
        if forced == 9 and self.cout > 4:
            forced = 0
        if forced == 10 and (self.cout > 32 or self.ntaps_total * self.cin_p > 512):
            forced = 0
        if forced == 11 and (self.cout > 4 or self.cin_p > 256 or (self.cin_p & (self.cin_p - 1))):
            forced = 0
        if forced == 38 and (len(self.cls) != 1 or self.cout > 64 or self.cin_p not in (4, 8) or self.ntaps_total > 9
                             or self.s_in > 2):
            forced = 0
        if forced == 76 and not self.c3_ok():
            forced = 0
        if forced in (28, 29, 47) and (self.cout > 4 or self.s_in != 1 or self.cin_p % (32 if forced == 28 else 16)):
            forced = 0
        if forced in X6D_TILES and (self.cin_p % 32 or any(c['Kpad'] != c['K'] for c in self.cls)):
            forced = 0
        if (forced in H16_TILES or forced == 68) and not in_f16:   # fp16 kernels forced (A/B runs) on a layer with fp32 input
            forced = 0
        tile = forced if forced else tuned_tile(key)
        if tile == 76 and not forced and (not self.c3_ok() or in_f16 or 'c3' in DEFAULT_DISABLE):   # (the key of a first layer, but not a 3-channel image: the rules decide)
            tile = -1
        if tile < 0:
            tile = self._default_tile(b * d.Hm * d.Wm)
        if out_f16 and not in_f16 and not forced and self.c3_ok() and not ({'c3', 'c3h'} & DEFAULT_DISABLE):
            # fp16-storage mode: a first layer over the 3-channel image multiplies fp16 operands (image rounded in registers, one fp16
            # weight plane) "like every other layer of the mode" -- at EVERY batch size: until round 6 only the pixel counts with a tune
            # entry (batch 64) took tile 76, a batch of 8 ran the same layer on fp32-exact operands, and a sub-batch did not reproduce its
            # rows of the full batch to the mode's own noise (profiles/r06_vgg_f16_bisect.txt)
            tile = 76
        if cin2k and self.wino is not None and tile % 100 not in (70, 71, 73) and not in_f16:
            tile = 70      # (two sources: only the Winograd kernel reads them -- and, in fp16 storage, the patch-staged fp16 kernel below)
        thin_mf = (forced in (0, 72) and 'thinmf' not in DEFAULT_DISABLE and not out_f16 and not masked and self.thin_ok())
        if thin_mf and not in_f16:
            # fp32 input (bf16x6: the 3-way operand split costs as much as the products): the stride-2 layers, whose four classes fill
            # the tile's 16 rows (ResNet stem 7 x 7: 230 against 325 us on the VALU kernel; conv1 3 x 3: 52 against 61; Inception 1a
            # 74 against 82); VGG-16's stride-1 first layer fills 4 of 16 rows: 465 against 265 (tools/lab/thin_time.py).  fp16 input:
            # always (55 against 306 us).
            if forced == 72 or self.s_out == 2:
                tile = 72   # thin output: the parity classes folded into the N dimension of a matrix-core tile (csrc/tapconv_thinmf.hip)
        if tile % 100 in (70, 71, 73):   # Winograd form of a 3x3 / stride-1 layer (csrc/tapconv_wino.hip): fp32 storage, same-size output
            # (tune values: 70 = the launcher's choice of N tile and K ranges, 71 = 64-wide N tile, 73 = 64-wide, four-wave workgroups; + 100 k = k K ranges, k = 1: none)
            if self.wino is not None and WINOGRAD and not (in_f16 or out_f16) and (hout, wout) == (hin + 2 * self.wino_pad - 2, win + 2 * self.wino_pad - 2):
                # (tile and K ranges travel as arguments: the shared Winograd plan keeps no per-call state)
                return self.wino.run(inp, out, add, gate, gate_mode, act, aux_out, gate2, in_coff, out_coff, add_coff, gate_coff,
                                     mask_out, gate_bits, gate2_bits, inp2, in2_coff, _wino=(tile % 100, tile // 100))
            tile = 0 if forced else self._default_tile(b * d.Hm * d.Wm, winograd=False)
        h16p_cv = False
        if in_f16:    # fp16 activations: the h16 kernels, N tile by the GEMM's width
            ngemm = self.cout * self.nfold
            same_ok = ((hin, win) == (d.Hm, d.Wm) and self.tap_range[0] >= -1 and self.tap_range[1] <= 1 and self.tap_range[2] >= -1
                       and self.tap_range[3] <= 1)
            # (unfolded stride-1 layers: any 3 x 3 tap window -- the unpadded 3 x 3 layers of Inception-v3's stem and their input gradients)
            span_ok = (self.nfold == 1 and self.s_out == 1 and self.tap_range[1] - self.tap_range[0] <= 2 and self.tap_range[3] - self.tap_range[2] <= 2
                       and 'h16pvalid' not in DEFAULT_DISABLE)
            patch_ok = (len(self.cls) == 1 and 4 <= self.ntaps_total <= 9 and self.s_in == 1 and (same_ok or span_ok)
                        and ((self.nfold == 1 and self.s_out == 1) or (self.nfold == 4 and self.s_out == 2)))
            # the forward form of a 3 x 3 / stride-2 convolution on the same kernel (S = 2: 8 x 32-pixel tiles, one patch buffer)
            patch2_ok = (len(self.cls) == 1 and 4 <= self.ntaps_total <= 9 and self.s_in == 2 and self.s_out == 1 and self.nfold == 1
                         and not cin2k and inp2 is None and self.tap_range[1] - self.tap_range[0] <= 2 and self.tap_range[3] - self.tap_range[2] <= 2)
            if forced in H16_TILES or (forced == 68 and (patch_ok or patch2_ok)):
                tile = forced
            elif thin_mf:
                tile = 72
            elif (not out_f16 and self.cout <= 4 and self.nfold == 1 and self.s_in == 1 and self.cin_p % 32 == 0 and not masked_any(mask_out, gate_bits, gate2_bits)
                  and forced in (0, 29) and 'thin' not in DEFAULT_DISABLE):
                tile = 29   # thin fp32 output from an fp16 activation (image-side input gradients): the patch-staged VALU kernel
            else:
                tile = 60 if ngemm > 64 else 61 if ngemm > 32 else 62 if ngemm > 16 else 63
                m_all = b * d.Hm * d.Wm
                # (32 GEMM columns leave half of the 64-wide tile empty and still win: Inception-v3 Conv2d_2a 138 -> 125 us, its input gradient 148 -> 128,
                # Conv2d_2b's 203 -> 170; `h16p64` in SPAA_DEFAULT_DISABLE: the round-4 threshold)
                if (patch_ok and ngemm >= (64 if 'h16p64' in DEFAULT_DISABLE else 32) and 'h16p' not in DEFAULT_DISABLE and forced == 0
                        and b * ((d.Hm + 15) // 16) * ((d.Wm + 31) // 32) * ((ngemm + 127) // 128) >= 256
                        and d.Hm * d.Wm >= 0.6 * ((d.Hm + 15) // 16 * 16) * ((d.Wm + 31) // 32 * 32)):   # (16 x 32-pixel tiles)
                    tile = 68   # 3x3 / stride 1 (or a folded stride-2 transposed layer): the input patch staged once for all taps (csrc/tapconv_h16p.hip)
                elif (patch2_ok and ngemm >= 64 and 'h16p2' not in DEFAULT_DISABLE and forced == 0
                      and b * ((d.Hm + 7) // 8) * ((d.Wm + 31) // 32) * ((ngemm + 127) // 128) >= 192
                      and d.Hm * d.Wm >= 0.6 * ((d.Hm + 7) // 8 * 8) * ((d.Wm + 31) // 32 * 32)):
                    tile = 68   # 3x3 / stride 2 forward: conv2 / conv2_s, transConv1's input gradient, the classifiers' stride-2 layers
                elif (patch_ok and same_ok and self.nfold == 1 and not cin2k and inp2 is None and ngemm >= 64 and self.cin_p >= 64 and forced == 0
                      and 'h16pcv' not in DEFAULT_DISABLE and m_all < (1 << 24) and max(d.Hm, d.Wm) <= 254):
                    # small images / few regions with long K (ResNet-18 layer3 / layer4, VGG-16's 14 x 14 block at batch 64): the
                    # patch-staged kernel's canvas / K-range form (the launcher's plan, asked for below: it needs the workspace)
                    tile, h16p_cv = 68, True
                elif tile == 60 and 'h16n64' not in DEFAULT_DISABLE and (
                        (m_all + 127) // 128 * ((ngemm + 127) // 128) < 256 or
                        (0 < ngemm % 128 <= 64 and (m_all + 127) // 128 * ((ngemm + 127) // 128) < 1024 and 'h16waste' not in DEFAULT_DISABLE)):
                    # too few 128 x 128 tiles for 256 CUs (ResNet layer3 / layer4 at batch 64): twice as many of 128 x 64; or a last
                    # 128-wide tile at most half full on a small grid (Inception-v3's 192-channel layers on 17 x 17 maps: 31 -> 26 us;
                    # Inception f16 73.2 -> 75.0 it/s.  Preferring 128 x 64 for every small grid costs VGG-16 3 %: tools/lab/f16_tiles.py)
                    tile = 61
                # (the 256-row tiles 64 / 65 paid on the 64 x 64 3x3 layers, which the patch-staged kernel serves now; for what is
                # left -- strided and folded layers -- 128 x 128 is as good or better: tools/lab/f16_tiles.py)
        elif out_f16:  # fp32 image in, fp16 activation out: any kernel built on the shared epilogue, without split-K
            tile %= 100
            # (the patch kernel's two-half form, Cout <= 64, is NOT preferred for VGG-16's first layer: measured on one box, VGG-16
            # PerC-AL fp16 storage 88.9 it/s with the layer on the register-staged bf16x6 tile against 86.4 on tile 38 -- 72 fp32 MFMAs
            # of 64 cycles per 64 pixels; tools/lab/body_masks_ab.sh, profiles/r05_body_masks_ab.txt)
            sc_ok = len(self.cls) == 1 and self.cin_p in (4, 8) and self.cout <= 64 and self.ntaps_total <= 9 and self.s_in <= 2
            if tile not in F16OUT_TILES or tile in H16_TILES:   # (the fp16 implicit-GEMM tiles need an fp16 input as well)
                tile = 38 if (sc_ok and self.cout <= 32) else (18 if self.cout > 32 else 16)
        # tune values >= 100 encode split-K: tile + 100 * ksplit (x6d tiles, one class, enough K-steps per split)
        ksplit, tile = (tile // 100, tile % 100) if tile >= 100 else (1, tile)
        if self.nfold > 1:  # only the DMA-staged kernels know the folded epilogue
            ksplit = 1
            if not (tile in X6D_TILES or tile in H16_TILES or tile == 68):
                tile = 34
            d.nfold = self.nfold
        if ksplit == 9:  # stream-K (persistent x6d tiles, one class): workspace shared by all plans (one stream)
            if len(self.cls) != 1 or self.nfold > 1 or tile not in X6D_PERSISTENT or self.cin_p % 32:
                ksplit, tile = 1, (0 if forced else tile)
            else:
                d.splitk_ws, d.ksplit = _streamk_workspace(inp.device).data_ptr(), -1
                ksplit = 1
        if in_f16 or out_f16:
            ksplit = 1
            if (in_f16 and tile in (60, 61, 62, 63) and len(self.cls) == 1 and self.nfold == 1 and 'h16splitk' not in DEFAULT_DISABLE
                    and (forced == 0 or FORCE_KSPLIT)):
                # skinny GEMMs of the fp16 path (VGG-16's fully connected layers at batch 64: ONE row tile, K = 25088; ResNet layer4):
                # split K until the grid covers the chip about twice, at least eight 64-deep steps per split
                bn = {60: 128, 61: 64, 62: 32, 63: 16}[tile]
                wgs = (b * d.Hm * d.Wm + 127) // 128 * ((self.cout + bn - 1) // bn)
                nk = (self.cls[0]['K'] + 63) // 64
                ksplit = FORCE_KSPLIT if FORCE_KSPLIT else max(1, min(16, 512 // max(wgs, 1), nk // 8))
        if ksplit > 1:
            nk = self.cls[0]['Kpad'] // BK
            if len(self.cls) != 1 or nk < 2 * ksplit or (tile not in X6D_TILES and not (in_f16 and tile in (60, 61, 62, 63))) or self.cin_p % 32:
                ksplit, tile = 1, (0 if forced else tile)
        if ksplit > 1:
            need = ksplit * b * d.Hm * d.Wm * ((self.cout + 127) // 128 * 128)
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, device=inp.device, dtype=torch.float32)
            d.splitk_ws, d.ksplit = self._ws.data_ptr(), ksplit
        if gate is not None and gate_mode == _lib.GATE_MUL and tile < 25:  # multiplicative gate: newer epilogues only
            tile = self._default_tile(b * d.Hm * d.Wm, winograd=False) % 100
            d.ksplit, d.splitk_ws = 0, None
            if tile < 25:
                raise ValueError('GATE_MUL needs a layer shape served by the DMA-staged kernels')
        if masked and tile not in STORE4_TILES:
            # byte masks live in the shared 4-channel epilogue (epilogue.hpp): thin / fp32-MFMA kernels do not have it
            tile = self._default_tile(b * d.Hm * d.Wm, winograd=False) % 100
            d.ksplit, d.splitk_ws = 0, None
            if tile not in STORE4_TILES:
                raise ValueError(f'{self.name}: gate masks need a layer shape served by the bf16x6 / smallcin kernels')
        if self.fixed_tile:
            tile, d.ksplit, d.splitk_ws = (_wino[0] if _wino else self.fixed_tile), 0, None
        if tile == 76:
            if not self.c3_ok() or in_f16:
                raise ValueError(f'{self.name}: tile 76 serves 3-channel-image convolutions only')
            if out_f16 and 'c3h' not in DEFAULT_DISABLE:
                # fp16-storage mode: fp16 operands like every other layer of the mode (the image rounded in registers, one fp16 weight plane)
                d.w_split = self.c3_pack(True).data_ptr()
                d.reserved1 |= 1
            else:
                d.w_split = self.c3_pack().data_ptr()
        if tile == 72:
            if in_f16:
                d.w_half = self.thin_fold(True).data_ptr()
            else:
                d.w_split = self.thin_fold(False).data_ptr()
        if cin2k:
            if in_f16 and self.nfold == 1:
                tile, d.ksplit, d.splitk_ws = 68, 0, None      # (fp16 storage: the patch-staged fp16 kernel's two-source form)
            elif tile not in (70, 71, 73):
                raise ValueError(f'{self.name}: a two-source plan runs on the Winograd kernel (fp32 storage, same-size output) or the patch-staged fp16 kernel')
            d.in2, d.in2_cstride, d.in2_coff, d.Cin2 = inp2.data_ptr(), inp2.shape[3], in2_coff, cin2k
        elif inp2 is not None and in_f16:
            # second source of a folded fp16 layer (attach_second_source_h16): the patch-staged fp16 kernel only
            if getattr(self, 'w2_half', None) is None or not out_f16 or self.nfold != 4:
                raise ValueError(f'{self.name}: an fp16 second source needs attach_second_source_h16() on a folded stride-2 layer')
            assert inp2.dtype == torch.float16 and inp2.shape[:3] == out.shape[:3] and in2_coff + self.cin2 <= inp2.shape[3]
            tile, d.ksplit, d.splitk_ws = 68, 0, None
            d.in2, d.in2_cstride, d.in2_coff, d.Cin2 = inp2.data_ptr(), inp2.shape[3], in2_coff, self.cin2
            d.w2_split = self.w2_half.data_ptr()
            d.bias = self.bias2.data_ptr()
        elif inp2 is not None:
            # second source (attach_second_source): the patch-staged stride-2 kernel only, fp32 storage
            if getattr(self, 'w2_split', None) is None or in_f16 or out_f16 or not self.x6p_ok():
                raise ValueError(f'{self.name}: a second source needs attach_second_source() on an fp32 stride-2 layer')
            assert inp2.dtype == torch.float32 and inp2.shape[:3] == out.shape[:3] and in2_coff + self.cin2 <= inp2.shape[3]
            tile, d.ksplit, d.splitk_ws = 74, 0, None
            d.in2, d.in2_cstride, d.in2_coff, d.Cin2 = inp2.data_ptr(), inp2.shape[3], in2_coff, self.cin2
            d.w2_split = self.w2_split.data_ptr()
            d.bias = self.bias2.data_ptr()
        if pool_adjoint is not None:
            if tile != 72:
                raise ValueError(f'{self.name}: pool_adjoint is served by the thin-output matrix-core kernel only (tile 72; got {tile})')
            d.in2, d.in2_cstride, d.in2_coff, d.Cin2 = parg.data_ptr(), inp.shape[1], inp.shape[2], int(bool(pgate))
        if tile == 74 and not (self.x6p_ok() and not (in_f16 or out_f16)):
            tile = 0 if forced else self._default_tile(b * d.Hm * d.Wm, winograd=False) % 100
        if tile == 74 and X6P_STD and self.x6p_canonical():
            d.reserved2 = 1      # (the kernel's compile-time schedule: x6p_canonical)
        d.tile = self.last_tile = tile     # (last_tile: for tests and reports)
        d.reserved0 = (DEBUG_TAPMAJOR | (DEBUG_PERSIST_CAP << 8) | (DEBUG_WINO << 16) | (DEBUG_H16_2STAGE << 25) | (DEBUG_SMALLCIN_NOSLAB << 26)
                       | (((DEBUG_THINMF & 7) << 27) if tile == 72 else 0)
                       | (((DEBUG_WINO_NOCANVAS & 1) << 30 | (DEBUG_WINO_NOCANVAS >> 1 & 1) << 29 | {1: 0, 0: 1, 2: 2}[getattr(self, 'wino_pad', 1)] << 27) if tile in (70, 71, 73) else 0))  # measurement / test switches; bits 27-28 of a Winograd launch: its zero padding (1 / 0 / 2)
        d.nclass = len(self.cls)
        d.tap_range[:] = self.tap_range
        for i, c in enumerate(self.cls):
            for k, v in c.items():
                setattr(d.cls[i], k, v)
        wino_bn = 0
        if tile in (70, 71, 73):
            # the launcher's plan (csrc/tapconv_wino.hip: N tile, canvas layout for small images, K ranges for few workgroups with
            # long K) -- asked for here because the K ranges need a workspace; its K-range count is then passed back explicitly
            want = _wino[1] if _wino else getattr(self, 'wino_ksplit', 0)
            d.ksplit = want if WINO_SPLITK else 1
            # (the plan depends on the shape, the tile, the K ranges asked for, a second source and the switches in reserved0 -- not
            # on pointers: asked once per such key, the launcher's own canvas search is the only one left per launch)
            pkey = (b, hin, win, hout, wout, cs_in, tile, d.ksplit, inp2 is not None, d.reserved0)
            wp = self._wino_plans.get(pkey)
            if wp is None:
                wpc = (C.c_int32 * 8)()
                rc = _lib.load().spaa_tapconv_wino_plan(C.byref(d), wpc)
                if rc != 0:
                    raise RuntimeError(f'{self.name}: spaa_tapconv_wino_plan failed with HIP error {rc}')
                wp = self._wino_plans[pkey] = tuple(wpc)
            wino_bn, d.ksplit = wp[0], wp[1]
            self.last_wino_plan = wp
            if d.ksplit > 1:
                need = d.ksplit * b * hout * wout * ((self.cout + 127) // 128 * 128)
                if WINO_SPLITK_FIXUP:
                    # round 6: the K ranges meet inside the kernel (the last-arriving workgroup of a tile adds them in fixed order): the
                    # workspace starts with a header of arrival counters, zero before and after every launch (include/spaa_hip.h)
                    if self._ws_fix is None or self._ws_fix.numel() < need + SPLITK_HDR:
                        self._ws_fix = torch.zeros(need + SPLITK_HDR, device=inp.device, dtype=torch.float32)
                    d.splitk_ws = self._ws_fix.data_ptr()
                    d.reserved1 |= 256
                else:
                    if self._ws is None or self._ws.numel() < need:
                        self._ws = torch.empty(need, device=inp.device, dtype=torch.float32)
                    d.splitk_ws = self._ws.data_ptr()
        if tile == 68 and h16p_cv:
            # the launcher's plan of the canvas / K-range form (csrc/tapconv_h16p.hip), cached per launch shape like the Winograd plans
            d.reserved1 = 4 | {0: 0, 64: 1, 128: 2}[H16P_CV[0]] | (8 if len(H16P_CV) > 2 and H16P_CV[2] else 0)
            d.ksplit, d.splitk_ws = H16P_CV[1], None
            pkey = ('h16p', b, hin, win, cs_in, d.reserved1, d.ksplit)
            wp = self._wino_plans.get(pkey)
            if wp is None:
                wpc = (C.c_int32 * 8)()
                rc = _lib.load().spaa_tapconv_h16p_plan(C.byref(d), wpc)
                if rc != 0:
                    raise RuntimeError(f'{self.name}: spaa_tapconv_h16p_plan failed with HIP error {rc}')
                wp = self._wino_plans[pkey] = tuple(wpc)
            d.ksplit = wp[1]
            d.reserved1 = (d.reserved1 & 8) | 4 | {64: 1, 128: 2}[wp[0]]
            self.last_h16p_plan = wp
            if d.ksplit > 1:
                need = d.ksplit * b * hout * wout * ((self.cout + 127) // 128 * 128)
                if WINO_SPLITK_FIXUP:   # (the K ranges meet inside the kernel: workspace with the arrival-counter header, as the Winograd form)
                    if self._ws_fix is None or self._ws_fix.numel() < need + SPLITK_HDR:
                        self._ws_fix = torch.zeros(need + SPLITK_HDR, device=inp.device, dtype=torch.float32)
                    d.splitk_ws = self._ws_fix.data_ptr()
                    d.reserved1 |= 256
                else:
                    if self._ws is None or self._ws.numel() < need:
                        self._ws = torch.empty(need, device=inp.device, dtype=torch.float32)
                    d.splitk_ws = self._ws.data_ptr()
        if tile == 68 and 'h16plean' in DEFAULT_DISABLE:
            d.reserved1 |= 16    # (A/B runs: the 64-wide stride-1 form as one workgroup per compute unit)
        if tile == 68 and not h16p_cv and self.s_in == 1 and self.cout * self.nfold > 64 and self.h16p_lean_wide(b, d.Hm, d.Wm):
            d.reserved1 |= 32    # (wider layers as 64-wide N tiles, two workgroups per compute unit)
        unpool_fused = False
        if unpool is not None:
            parg, gfull = unpool
            if (tile == 68 and not h16p_cv and self.s_in == 1 and self.nfold == 1 and (self.cout <= 64 or (d.reserved1 & 32)) and not (d.reserved1 & 16)
                    and self.tap_range[0] >= -1 and self.tap_range[1] <= 1 and self.tap_range[2] >= -1 and self.tap_range[3] <= 1
                    and self.cin_p % 8 == 0 and 'h16punp' not in DEFAULT_DISABLE):
                d.in2, d.in2_cstride, d.in2_coff, d.Cin2 = parg.data_ptr(), parg.shape[3], 0, 0
                d.reserved1 |= 128
                unpool_fused = True
            else:
                _lib.call('spaa_maxpool_bwd_f16', _lib.hptr(inp), _lib.ptr(parg), 1, _lib.hptr(gfull), b, hin, win, self.cin_p, hin // 2, win // 2,
                          2, 2, 0, cs_in, 0)
                d.inp, d.in_cstride = gfull.data_ptr(), gfull.shape[3]
        self.last_unpool_fused = unpool_fused
        pool_fused = False
        if pool is not None:
            assert (act == _lib.ACT_RELU and add is None and gate is None and gate2 is None and aux_out is None and mask_out is None
                    and gate_bits is None and gate2_bits is None and out_coff == 0)
            pooled, parg, want_arg = pool
            assert pooled.shape == (b, hout // 2, wout // 2, cs_out) and pooled.dtype == out.dtype and parg.shape == (b, hout // 2, wout // 2, self.cout)
            if (tile == 68 and not h16p_cv and self.s_in == 1 and self.nfold == 1 and hout % 2 == 0 and wout % 2 == 0 and self.cout % 4 == 0
                    and cs_out == self.cout and 'h16ppool' not in DEFAULT_DISABLE):
                d.out, d.mask_out = pooled.data_ptr(), (parg.data_ptr() if want_arg else None)
                d.reserved1 |= 64
                pool_fused = True
        self.last_pool_fused = pool_fused
        tid = 0
        if PROFILE is not None:
            tid = d.tile + 100 * (d.ksplit if d.ksplit > 1 else (9 if d.ksplit == -1 else 0))
            if d.tile == 68 and h16p_cv:
                tid += 1000
            if d.tile == 70 and wino_bn == 64:   # the launcher's choice of the N tile: a kernel of its own for rocprofv3
                tid += 1
            if d.tile in (70, 71, 73):           # ... and so are the canvas / K-range form (+ 1000) and the two-source form (+ 2000)
                tid += 2000 if cin2k else (1000 if (self.last_wino_plan[2] or d.ksplit > 1) else 0)
        if PROFILE is None or (PROFILE_ONLY is not None and tid not in PROFILE_ONLY):
            _lib.call('spaa_tapconv_f32', C.byref(d))
        else:  # bench.py's instrumented pass: HIP events on the launch stream around this one kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.call('spaa_tapconv_f32', C.byref(d))
            e1.record()
            # algorithmic bytes: every operand read / result written once (logical channels, fp32)
            npx = b * hout * wout
            bi, bo = (2 if in_f16 else 4), (2 if out_f16 else 4)   # bytes per element: fp16 storage or fp32
            nbytes = (bi * b * hin * win * self.cin_p + bo * npx * self.cout * (1 + (add is not None) + (gate is not None)
                                                                               + (aux_out is not None) + (gate2 is not None))
                      + bi * self.alg_taps * self.cin_p * self.cout)
            nbytes += npx * self.cout // 4 * ((mask_out is not None) + (gate_bits is not None) + (gate2_bits is not None))
            fl = self.flops(b, hout, wout)
            if inp2 is not None and not cin2k:   # (second source of the stride-2 kernel; a two-source plan's channels are in cin_p)
                nbytes += 4 * npx * self.cin2
                fl += 2 * npx * self.cin2 * self.cout
            if unpool_fused:   # (the pooled gradient and the arg-max bytes instead of the full-size gradient)
                nbytes -= bi * b * hin * win * self.cin_p - (bi + 1) * b * (hin // 2) * (win // 2) * self.cin_p
            if pool_fused:   # (the pooled tensor and its arg-max bytes instead of the full-size activation)
                nbytes -= bo * npx * self.cout - (bo + 1) * (npx // 4) * self.cout
            PROFILE.append((self.name, key, fl, e0, e1, tid, nbytes))
        if pool is not None and not pool_fused:
            pooled, parg, _want = pool
            _lib.call('spaa_maxpool_fwd_f16' if out_f16 else 'spaa_maxpool_fwd', _lib.hptr(out), _lib.hptr(pooled), _lib.ptr(parg), b, hout, wout,
                      self.cout, hout // 2, wout // 2, 2, 2, 0, pooled.shape[3], 0)
        return out

    def h16p_lean_wide(self, b, hm, wm):
        """Patch-staged fp16 kernel, stride-1 form, more than 64 GEMM columns: 64-wide N tiles with two workgroups per compute unit
        (csrc/tapconv_h16p.hip LEAN) instead of one 128-wide workgroup?  SPAA_H16P_LEAN_WIDE: 0 never, 1 always (A/B runs), 2 (default) by
        the layer's shape."""
        if H16P_LEAN_WIDE != 2:
            return H16P_LEAN_WIDE == 1
        # measured per layer in both loops (profiles/r05_h16p_lean_wide.txt): up to eight 32-channel blocks the overlap of one
        # workgroup's patch loads / epilogue with the other's products wins (conv4 195 -> 174 us, transConv1 + skipConv2 166 -> 148, VGG-16
        # features.5 217 -> 184); with sixteen blocks the 128-wide tile's halved patch traffic does (features.19 / 21: + 4-6 us)
        return self.cin_p <= 256

    def wgrad(self, inp, gout, dbias=True, nchunk=None, out_coff=0, in_coff=0):
        """Weight (and bias) gradient of this layer in its forward form (csrc/tapconv_wgrad.hip): `inp` = the layer's input
        activation [B,Hin,Win,Cs], `gout` = gradient w.r.t. its pre-activation [B,Hout,Wout,Cs'].  Returns (dW in the packed
        layout of `self.weights`, dbias [cout] or None); `unpack_grad` maps dW back to the parameter's shape."""
        _lib.check_dev(inp, gout)
        assert self.nfold == 1, 'weight gradients use the unfolded transposed-convolution plan'
        b, hin, win, cs_in = inp.shape
        b2, hout, wout, cs_out = gout.shape
        assert b == b2 and in_coff + self.cin_p <= cs_in and out_coff + self.cout <= cs_out
        d = _lib.TapConv()
        d.inp, d.Hin, d.Win, d.Cin, d.in_cstride, d.in_coff = inp.data_ptr(), hin, win, self.cin_p, cs_in, in_coff
        d.Hout, d.Wout, d.Cout, d.out_cstride, d.out_coff = hout, wout, self.cout, cs_out, out_coff
        d.B = b
        if self.s_out == 1:
            d.Hm, d.Wm = hout, wout
        else:
            d.Hm, d.Wm = (hout + self.s_out - 1) // self.s_out, (wout + self.s_out - 1) // self.s_out
        d.s_in, d.s_out = self.s_in, self.s_out
        d.taps = self.taps.data_ptr()
        d.nclass = len(self.cls)
        for i, c in enumerate(self.cls):
            for k, v in c.items():
                setattr(d.cls[i], k, v)
        m = b * d.Hm * d.Wm
        if nchunk is None:   # enough waves to fill the chip: (taps x n tiles x c groups) x chunks ~ 4096
            blocks = self.ntaps_total * ((self.cout + 31) // 32) * ((self.cin_p + 127) // 128)
            nchunk = max(1, min(1024, (4096 + blocks - 1) // blocks, max(1, m // 64)))
        wtotal = self.weights.numel()
        need = nchunk * max(wtotal, self.cout)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=inp.device, dtype=torch.float32)
        dw = torch.empty(wtotal, device=inp.device)
        db = torch.empty(self.cout, device=inp.device) if dbias else None
        _lib.call('spaa_tapconv_wgrad', C.byref(d), _lib.ptr(gout), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(self._ws), int(nchunk))
        return dw, db

    def refresh(self, weight, bias=None):
        """Re-pack a parameter that has changed (training): fp32 matrix and, for the bf16x6 kernels, its three bf16
        planes — on the device (needs the maps of `attach_maps`).  Exactly what the constructor does on the host."""
        if getattr(self, 'w2_split', None) is not None or getattr(self, 'w2_half', None) is not None or getattr(self, 'cin2_k', 0):
            # a fused second source (attach_second_source / the two-source Winograd plans) keeps packed copies of ANOTHER layer's
            # parameters and a summed bias: re-packing only this layer's would leave them stale -- build the plan again instead
            raise RuntimeError(f'{self.name}: refresh() on a plan with a fused second source; rebuild it (PCNetTrainer builds its '
                               'engine with fuse_skip2=False for this reason)')
        w = weight.detach().float().reshape(-1)
        self.weights[self.repack_pos] = w[self.repack_src]
        if self.w_split is not None:
            parts = []
            for c in self.cls:
                wp = self.weights[c['w_off']:c['w_off'] + self._npad * c['Kpad']]
                h = wp.to(torch.bfloat16)
                r1 = wp - h.float()
                m = r1.to(torch.bfloat16)
                lo = (r1 - m.float()).to(torch.bfloat16)
                parts.append(torch.stack([h, m, lo]).view(torch.int16).reshape(-1))
            self.w_split.copy_(torch.cat(parts))
        self.w_half = None
        self._thin = {}
        self._c3 = self._c3h = None
        if self.wino is not None:
            _winograd_weights(self, self.wino)
        if bias is not None:
            self.bias.copy_(bias.detach().float())

    def unpack_grad(self, dw_packed):
        """Packed dW -> gradient in the shape of the parameter this plan was built from (needs `unpack_idx`, attached by
        `attach_unpack`)."""
        return dw_packed[self.unpack_idx].view(self.param_shape)

    def _default_tile(self, m, winograd=True):
        """Kernel choice for a layer shape that tools/autotune.py has not measured: the family that wins for the
        measured shapes of the same kind."""
        one = len(self.cls) == 1
        off = DEFAULT_DISABLE  # debugging aid: families to leave out of the default choice
        if winograd and self.wino is not None and 'wino' not in off and self.cout >= 128 and m >= 50176 and m * self.cout >= 50176 * 256:
            return 70                                                   # big 3x3 / s1 layer: Winograd F(2x2,3x3)
        if 'thin' not in off and self.cout <= 4 and self.s_in == 1 and self.cin_p % 16 == 0:
            return 29                                                   # thin output: patch-staged VALU kernel
        if 'smallcin' not in off and one and self.cin_p in (4, 8) and self.cout <= 32 and self.ntaps_total <= 9 and self.s_in <= 2:
            return 38                                                   # few input channels: fp32 MFMA from an LDS patch
        if self.w_split is None:
            return 0
        if 'x6d' not in off and self.cin_p % 32 == 0:                   # DMA-staged bf16x6 kernels
            ngemm = self.cout * self.nfold
            tile = 34 if ngemm > 64 else (36 if ngemm > 32 else 37)
            nk = self.cls[0]['Kpad'] // BK
            wgs = (m + 127) // 128 * ((ngemm + 127) // 128 if tile == 34 else 1)
            if 'splitk' not in off and one and self.nfold == 1 and wgs < 256 and nk >= 16:      # few pixels, long K: split K to fill the chip
                tile += 100 * (4 if wgs < 128 and nk >= 32 else 2)
            return tile
        return 18 if self.cout > 32 else 16                             # register-staged bf16x6 kernels

    def flops(self, b, hout, wout):
        hm = hout if self.s_out == 1 else (hout + 1) // 2
        wm = wout if self.s_out == 1 else (wout + 1) // 2
        return b * hm * wm * self.flops_per_pixel


_WINO_G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]], dtype=torch.float64)


def _winograd_weights(plan, wino):
    """U = G g G^T of every (output, input) channel pair of `plan` (fp64 on the device, rounded once to fp32) into `wino`'s
    packed 16-'tap' matrix W[n][pos * Cin + c] and its three bf16 planes."""
    c = plan.cls[0]
    g = plan.weights[:plan._npad * c['Kpad']].view(plan._npad, c['Kpad'])[:, :9 * plan.cin_p]
    g = g.reshape(plan._npad, 9, plan.cin_p).double()
    u = torch.einsum('pt,ntc->npc', wino._wino_t.to(g.device), g).float().reshape(plan._npad, -1)
    u = torch.nn.functional.pad(u, (0, wino.cls[0]['Kpad'] - u.shape[1])).reshape(-1)
    wino.weights.copy_(u)
    h = u.to(torch.bfloat16)
    r1 = u - h.float()
    m = r1.to(torch.bfloat16)
    lo = (r1 - m.float()).to(torch.bfloat16)
    wino.w_split.copy_(torch.stack([h, m, lo]).view(torch.int16).reshape(-1))


def attach_winograd(plan):
    """Give a 3x3 / stride-1 / pad-1 convolution plan (forward or input gradient) its Winograd F(2x2,3x3) form: the filter
    transform U = G g G^T as a 16-'tap' ConvPlan run by csrc/tapconv_wino.hip (tile 70).  Returns the plan; plans of any other
    shape are returned untouched."""
    if (len(plan.cls) != 1 or plan.ntaps_total != 9 or plan.s_in != 1 or plan.s_out != 1 or plan.nfold != 1
            or plan.cin_p % 32 or plan.cin != plan.cin_p or plan.w_split is None or plan.cout < 64):
        return plan
    taps = [(dy, dx) for dy, dx, _ in plan.classes_host[0].taps]
    # pad 1 (taps -1..1: same-size output), pad 0 (taps 0..2: the unpadded layer, output 2 smaller: Inception-v3's Conv2d_2a / 4a) or
    # pad 2 (taps -2..0: that layer's input gradient, output 2 larger): the kernel's patch origin moves, nothing else
    sh = min(dy for dy, _ in taps) + 1
    if sh not in (-1, 0, 1) or sorted(taps) != [(dy + sh, dx + sh) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]:
        return plan
    taps = [(dy - sh, dx - sh) for dy, dx in taps]
    t = torch.zeros(16, 9, dtype=torch.float64)  # U[pos] = sum_t T[pos, t] g_t,  g_t = the tap at offset (dy, dx)
    for i, (dy, dx) in enumerate(taps):
        for xi in range(4):
            for nu in range(4):
                t[4 * xi + nu, i] = _WINO_G[xi, dy + 1] * _WINO_G[nu, dx + 1]
    c = TapClassSpec(0, 0)
    zero = torch.zeros(plan.cout, plan.cin)
    for xi in range(4):
        for nu in range(4):
            c.add(xi, nu, zero)
    # rows of 16 * Cin bf16 are a multiple of 4 KiB apart for Cin = 128, 256: every row of a DMA piece would hit the same L2
    # channel (measured 16x slower); 128 extra elements (256 B) walk the channels
    wino = ConvPlan([c], plan.cin, plan.cout, 1, 1, None, plan._dev, plan.name, kpad_extra=128)
    wino.bias = plan.bias            # (shared tensor: refresh() updates both)
    wino._wino_t = t
    wino.fixed_tile = 70
    wino.wino_pad = plan.wino_pad = 1 - sh
    wino.flops_per_pixel, wino.alg_taps = plan.flops_per_pixel, 9   # algorithmic work = the direct convolution's
    _winograd_weights(plan, wino)
    plan.wino = wino
    return plan


def conv_fwd_plan_2src(weight_a, weight_b, bias, device='cuda', name=''):
    """conv(a, Wa) + conv(b, Wb) + bias for two 3x3 / s1 / p1 convolutions of tensors a, b of the same size (`conv5(x4) +
    skipConv3(x2)`, /root/reference/src/python/models.py:294,298) as ONE convolution over the channel-concatenated input, run by
    the Winograd kernel whose patch DMA reads the channel blocks of `a` from `inp` and those of `b` from `inp2`: no concatenated
    tensor, no second launch, one accumulation chain.  None when the layer has no Winograd form."""
    w = torch.cat([_w2(weight_a), _w2(weight_b)], 1)
    plan = conv_fwd_plan(w, bias, 1, 1, device, name)
    if plan.wino is None or weight_a.shape[1] % 32 or weight_b.shape[1] % 32:
        return None
    plan.cin2_k = plan.wino.cin2_k = weight_b.shape[1]
    return plan


def conv_dgrad_plan_2src(weight_a, weight_b, device='cuda', name=''):
    """conv_a^T(ga) + conv_b^T(gb): the input gradients of two 3x3 / s1 / p1 convolutions that read the SAME tensor (conv3 and
    skipConv3 both read x2), as one launch (see conv_fwd_plan_2src); `inp` = ga, `inp2` = gb."""
    w = torch.cat([_w2(weight_a), _w2(weight_b)], 0)
    plan = conv_dgrad_plan(w, 1, 1, device, name)
    if plan.wino is None or weight_a.shape[0] % 32 or weight_b.shape[0] % 32:
        return None
    plan.cin2_k = plan.wino.cin2_k = weight_b.shape[0]
    return plan


def split_planes(w):
    """fp32 matrix -> int16 tensor [3, *w.shape]: the bf16 planes h, m, l with w == h + m + l exactly."""
    w = w.detach().float().contiguous()
    h = w.to(torch.bfloat16)
    r1 = w - h.float()
    m = r1.to(torch.bfloat16)
    lo = (r1 - m.float()).to(torch.bfloat16)
    return torch.stack([h, m, lo]).view(torch.int16).contiguous()


def _w2(w):
    return w.detach().float().cpu()


def conv_fwd_plan(weight, bias, stride, pad, device='cuda', name=''):
    """nn.Conv2d forward. weight [co, ci, kh, kw]."""
    w = _w2(weight)
    co, ci, kh, kw = w.shape
    ph, pw = _pair(pad)
    c = TapClassSpec(0, 0)
    for ky in range(kh):
        for kx in range(kw):
            c.add(ky - ph, kx - pw, w[:, :, ky, kx])
    return attach_winograd(ConvPlan([c], ci, co, stride, 1, bias, device, name))


def _pair(p):
    return (p, p) if isinstance(p, int) else (int(p[0]), int(p[1]))


def _fractional_classes(wsel, kh, kw, pad):
    """Stride-2 transposed structure: output (2y+py, 2x+px) receives taps with (py+pad-ky) even."""
    ph, pw = _pair(pad)
    classes = []
    for py in range(2):
        for px in range(2):
            c = TapClassSpec(py, px)
            for ky in range(kh):
                if (py + ph - ky) % 2:
                    continue
                for kx in range(kw):
                    if (px + pw - kx) % 2:
                        continue
                    c.add((py + ph - ky) // 2, (px + pw - kx) // 2, wsel(ky, kx))
            classes.append(c)
    return classes


def _fold_classes(classes, cout, cin):
    """Four output-parity classes of a stride-2 fractional layer as ONE class over the union of their taps, the classes stacked
    in the GEMM rows (row c * cout + n, c = 2 py + px); a class without a tap gets zero weights there (3x3 / s2: 9 of the 16
    (class, tap) pairs are real).  One pass over the input instead of four, N = 4 cout."""
    by_par = {(c.oy0, c.ox0): c for c in classes}
    taps = sorted({(dy, dx) for c in classes for dy, dx, _ in c.taps})
    zero = torch.zeros(cout, cin)
    f = TapClassSpec(0, 0)
    for dy, dx in taps:
        rows = []
        for par in ((0, 0), (0, 1), (1, 0), (1, 1)):
            w = [w_ for dy_, dx_, w_ in by_par[par].taps if (dy_, dx_) == (dy, dx)]
            rows.append(w[0] if w else zero)
        f.add(dy, dx, torch.cat(rows, 0))
    return f


def conv_dgrad_plan(weight, stride, pad, device='cuda', name='', in_ch=None, fold=None):
    """Input gradient of nn.Conv2d: consumes grad_out [.., co], produces grad_in [.., ci].
    `in_ch=(lo, hi)` restricts the produced input channels (used for conv1_s, whose first 3 inputs are constant)."""
    w = _w2(weight)
    co, ci, kh, kw = w.shape
    lo, hi = in_ch if in_ch is not None else (0, ci)

    def wsel(ky, kx):
        return w[:, lo:hi, ky, kx].t().contiguous()

    if stride == 1:
        ph, pw = _pair(pad)
        c = TapClassSpec(0, 0)
        for ky in range(kh):
            for kx in range(kw):
                c.add(ph - ky, pw - kx, wsel(ky, kx))
        return attach_winograd(ConvPlan([c], co, hi - lo, 1, 1, None, device, name))
    assert stride == 2
    classes = _fractional_classes(wsel, kh, kw, pad)
    if (FOLD_K3S2 if fold is None else fold) and kh == 3 and kw == 3 and co % 32 == 0 and (hi - lo) % 4 == 0 and ENABLE_X6 and (fold or hi - lo <= FOLD_K3S2_MAX_COUT):
        return ConvPlan([_fold_classes(classes, hi - lo, co)], co, hi - lo, 1, 2, None, device, name, nfold=4)
    return ConvPlan(classes, co, hi - lo, 1, 2, None, device, name)


def deconv_fwd_plan(weight, bias, stride, pad, device='cuda', name='', fold=None):
    """nn.ConvTranspose2d forward (stride 2). weight [ci, co, kh, kw].  `fold=False`: keep the four parity classes
    separate (the training path: weight gradients are taken per class)."""
    assert stride == 2
    w = _w2(weight)
    ci, co, kh, kw = w.shape

    def wsel(ky, kx):
        return w[:, :, ky, kx].t().contiguous()

    if (FOLD_DECONV if fold is None else fold) and kh == 2 and kw == 2 and pad == 0 and ci % 32 == 0 and co % 4 == 0 and ENABLE_X6:
        # the four output-parity classes share the tap (0, 0): fold them into the GEMM rows (row c*co + n, c = 2 dy + dx)
        c = TapClassSpec(0, 0)
        c.add(0, 0, torch.cat([wsel(dy, dx) for dy in (0, 1) for dx in (0, 1)], 0))
        return ConvPlan([c], ci, co, 1, 2, bias, device, name, nfold=4)
    classes = _fractional_classes(wsel, kh, kw, pad)
    if fold and kh == 3 and kw == 3 and ci % 32 == 0 and co % 4 == 0 and ENABLE_X6:   # (measured slower for transConv1: opt-in)
        return ConvPlan([_fold_classes(classes, co, ci)], ci, co, 1, 2, bias, device, name, nfold=4)
    return ConvPlan(classes, ci, co, 1, 2, bias, device, name)


def deconv_dgrad_plan(weight, stride, pad, device='cuda', name=''):
    """Input gradient of nn.ConvTranspose2d: a stride-`stride` convolution of grad_out."""
    w = _w2(weight)
    ci, co, kh, kw = w.shape
    c = TapClassSpec(0, 0)
    for ky in range(kh):
        for kx in range(kw):
            c.add(ky - pad, kx - pad, w[:, :, ky, kx].contiguous())
    return ConvPlan([c], co, ci, stride, 1, None, device, name)


def attach_maps(plan, builder, weight):
    """Index plumbing for training: where does each element of the parameter tensor sit in the packed weight matrix?
    `builder(w)` is the plan constructor that built `plan` (as a function of the weight alone); it is run once more on a
    tensor of element indices.  Keeps on the device
      repack_pos / repack_src : packed[repack_pos] = param.flatten()[repack_src]   (every used packed position;
                                `ConvPlan.refresh` re-packs a CHANGED parameter with it, no host round trip)
      unpack_idx              : param_grad.flatten() = dW_packed[unpack_idx]       (only when every element is used once,
                                i.e. for forward plans)."""
    n = weight.numel()
    assert n < (1 << 24)   # indices travel through the fp32 packing exactly
    idx_w = (torch.arange(n, dtype=torch.float32) + 1).view(weight.shape)
    ip = builder(idx_w)
    packed = ip.weights.cpu().round().long()
    assert packed.numel() == plan.weights.numel()
    pos = torch.nonzero(packed, as_tuple=False).view(-1)
    dev = plan.weights.device
    plan.repack_pos, plan.repack_src = pos.to(dev), (packed[pos] - 1).to(dev)
    plan.param_shape = tuple(weight.shape)
    if pos.numel() == n and torch.unique(packed[pos]).numel() == n:
        unpack = torch.empty(n, dtype=torch.long)
        unpack[packed[pos] - 1] = pos
        plan.unpack_idx = unpack.to(dev)
    return plan


SMALL_LINEAR = os.environ.get('SPAA_SMALL_LINEAR', '1') != '0'   # 0: the classifiers' last layer on the implicit-GEMM tiles (A/B measurements)


class SmallLinearPlan:
    """nn.Linear on at most 64 rows as a wave-per-output kernel (csrc/linear_small.hip); more rows, fp16 tensors or a fused
    epilogue go to the 1 x 1 convolution plan it wraps.  `w_rows` [N, K]: row n holds the weights of output n."""

    def __init__(self, conv_plan, w_rows, bias, device, transposed=False):
        self.conv, self.name, self.transposed = conv_plan, conv_plan.name, transposed
        self.w = w_rows.detach().float().contiguous().to(device)
        self.bias = bias.detach().float().contiguous().to(device) if bias is not None else None
        self.n, self.k = self.w.shape
        self.last_tile = -1

    def applies(self, inp, out, kw):
        m = inp.shape[0] * inp.shape[1] * inp.shape[2]
        return (SMALL_LINEAR and not kw and m <= 64 and inp.dtype == out.dtype == torch.float32 and inp.shape[3] == self.k
                and out.shape[3] == self.n and self.k % 4 == 0 and 4 <= self.k <= 4096 and inp.is_contiguous() and out.is_contiguous())

    def run(self, inp, out, **kw):
        if not self.applies(inp, out, kw):
            r = self.conv.run(inp, out, **kw)
            self.last_tile = self.conv.last_tile
            return r
        _lib.check_dev(inp, out)
        m = inp.shape[0] * inp.shape[1] * inp.shape[2]
        _lib.call('spaa_linear_small', _lib.ptr(inp), _lib.ptr(self.w), _lib.ptr(self.bias) if self.bias is not None else None, _lib.ptr(out),
                  m, self.k, self.n, self.k, self.k, self.n)
        self.last_tile = 75
        return out

    def refresh(self, weight, bias=None):
        """A changed parameter (training): the wrapped plan's packed matrices AND this plan's own row-major copy."""
        self.conv.refresh(weight, bias)
        w = _w2(weight).to(self.w.device)
        self.w.copy_(w.t() if self.transposed else w)
        if bias is not None and self.bias is not None:
            self.bias.copy_(bias.detach().float())

    def __getattr__(self, item):      # (flops(), tune keys ...: the wrapped plan's)
        return getattr(self.conv, item)


def linear_fwd_plan(weight, bias, device='cuda', name=''):
    """nn.Linear as a 1x1 convolution over a [B,1,1,C] activation."""
    w = _w2(weight)
    c = TapClassSpec(0, 0)
    c.add(0, 0, w)
    plan = ConvPlan([c], w.shape[1], w.shape[0], 1, 1, bias, device, name)
    return SmallLinearPlan(plan, w, bias, device) if w.numel() <= (1 << 22) and w.shape[1] % 4 == 0 else plan


def linear_dgrad_plan(weight, device='cuda', name=''):
    w = _w2(weight)
    c = TapClassSpec(0, 0)
    c.add(0, 0, w.t().contiguous())
    plan = ConvPlan([c], w.shape[0], w.shape[1], 1, 1, None, device, name)
    return SmallLinearPlan(plan, w.t().contiguous(), None, device, transposed=True) if w.numel() <= (1 << 22) and w.shape[0] % 4 == 0 else plan


def fold_bn(weight, bn_w, bn_b, bn_mean, bn_var, eps=1e-5):
    """Eval-mode BatchNorm folded into the preceding bias-free convolution."""
    scale = bn_w.double() / torch.sqrt(bn_var.double() + eps)
    w = (weight.double() * scale.view(-1, 1, 1, 1)).float()
    b = (bn_b.double() - bn_mean.double() * scale).float()
    return w, b
