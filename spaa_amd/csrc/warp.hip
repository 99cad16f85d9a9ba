// warp.hip — WarpingNet: sampling-grid construction and bilinear grid_sample forward / backward.
//
// Replaces (paths relative to /root/reference/src/python):
//   F.affine_grid + pytorch_tps.tps_grid + F.grid_sample(affine, tps)        models.py:168-172, pytorch_tps.py:54-106
//   clamp(refine + tps, -1, 1)                                               models.py:176
//   F.grid_sample(clamp(x,0,1), fine_grid, align_corners=True) * mask ; x*s  models.py:184,340,342;
//                                                                            projector_based_attack.py:265
//   grid_sampler_2d_backward (w.r.t. the image only; the grid is frozen)     autograd of the above
// The fine grid does not depend on the optimised image, so it is built once per attack (batch 1) instead of once
// per iteration replicated B times as the reference does.  All images are NHWC4 (float4 per pixel): one 16-byte
// load per bilinear tap instead of three scalar gathers.  HBM-bound: algorithmic traffic per scene-iteration is
// read x (Hp*Wp*16 B) + write xw + cat8 (Hc*Wc*48 B) forward.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "warp_common.hpp"

namespace {

__global__ void coarse_grid_kernel(const float* __restrict__ affine6, const float* __restrict__ theta,
                                   const float* __restrict__ ctrl, int T, int Hin, int Win, int Hout, int Wout,
                                   float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hout * Wout) return;
    const int oy = idx / Wout, ox = idx - oy * Wout;
    // pytorch_tps.tps_grid: homogeneous grid (1, x, y) over [0,1]
    const float px = linspace_at(0.f, 1.f, Wout, ox);
    const float py = linspace_at(0.f, 1.f, Hout, oy);
    // reduced form: theta has T+2 rows: T-1 free weights, then 3 affine rows; w_0 = -sum(others)   (:65-69)
    float bx = 0.f, by = 0.f, wsx = 0.f, wsy = 0.f, u0 = 0.f;
    for (int t = 0; t < T; ++t) {
        const float dx = px - ctrl[2 * t], dy = py - ctrl[2 * t + 1];
        const float d = sqrtf(dx * dx + dy * dy);
        const float u = (d * d) * logf(d + 1e-6f);
        if (t == 0) {
            u0 = u;
        } else {
            const float wx = theta[2 * (t - 1)], wy = theta[2 * (t - 1) + 1];
            bx += u * wx;
            by += u * wy;
            wsx += wx;
            wsy += wy;
        }
    }
    bx += u0 * (-wsx);
    by += u0 * (-wsy);
    const float* a = theta + 2 * (T - 1);
    const float zx = (a[0] + px * a[2] + py * a[4]) + bx;
    const float zy = (a[1] + px * a[3] + py * a[5]) + by;
    const float tx = (px + zx) * 2.f - 1.f;
    const float ty = (py + zy) * 2.f - 1.f;
    // sample the affine grid (evaluated analytically at the four taps; zeros outside)
    const Bilinear bl = bilinear_setup(tx, ty, Win, Hin);
    float gx = 0.f, gy = 0.f;
    auto tap = [&](int yy, int xx, float wgt) {
        if ((unsigned)yy < (unsigned)Hin && (unsigned)xx < (unsigned)Win) {
            const float bxn = linspace_at(-1.f, 1.f, Win, xx);
            const float byn = linspace_at(-1.f, 1.f, Hin, yy);
            gx += (bxn * affine6[0] + byn * affine6[1] + affine6[2]) * wgt;
            gy += (bxn * affine6[3] + byn * affine6[4] + affine6[5]) * wgt;
        }
    };
    tap(bl.y0, bl.x0, bl.nw);
    tap(bl.y0, bl.x0 + 1, bl.ne);
    tap(bl.y0 + 1, bl.x0, bl.sw);
    tap(bl.y0 + 1, bl.x0 + 1, bl.se);
    reinterpret_cast<float4*>(out)[idx] = make_float4(gx, gy, 0.f, 0.f);
}

__global__ void finish_grid_kernel(const float4* __restrict__ coarse, const float4* __restrict__ refine,
                                   float4* __restrict__ fine, int npix) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const float4 c = coarse[idx];
    float gx = c.x, gy = c.y;
    if (refine != nullptr) {
        const float4 r = refine[idx];
        gx = r.x + c.x;
        gy = r.y + c.y;
    }
    fine[idx] = make_float4(fminf(fmaxf(gx, -1.f), 1.f), fminf(fmaxf(gy, -1.f), 1.f), 0.f, 0.f);
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// (Kept in this form: restructuring the tap loads changes how the compiler contracts bilinear_setup's coordinate arithmetic at
// this call site -- coordinates move by ~W * 2^-24 pixels, outputs by up to 1e-5 -- and with it the validated trajectories of
// tests/test_gpu_parity.py::test_free_running_drift_vs_fp64_oracle; the 10 us it would save are not worth re-validating.)
__global__ void warp_fwd_kernel(const float4* __restrict__ x, const float4* __restrict__ grid,
                                const float* __restrict__ mask, const float4* __restrict__ s,
                                float4* __restrict__ xw, float4* __restrict__ cat8, int B, int Hp, int Wp, int HWc,
                                int clamp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HWc) return;
    const int b = idx / HWc, pix = idx - b * HWc;
    const float4 g = grid[pix];
    const Bilinear bl = bilinear_setup(g.x, g.y, Wp, Hp);
    const float4* xb = x + (size_t)b * Hp * Wp;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    auto tap = [&](int yy, int xx, float wgt) {
        if ((unsigned)yy < (unsigned)Hp && (unsigned)xx < (unsigned)Wp) {
            float4 v = xb[yy * Wp + xx];
            if (clamp) {
                v.x = clamp01(v.x);
                v.y = clamp01(v.y);
                v.z = clamp01(v.z);
            }
            r0 += v.x * wgt;
            r1 += v.y * wgt;
            r2 += v.z * wgt;
        }
    };
    tap(bl.y0, bl.x0, bl.nw);
    tap(bl.y0, bl.x0 + 1, bl.ne);
    tap(bl.y0 + 1, bl.x0, bl.sw);
    tap(bl.y0 + 1, bl.x0 + 1, bl.se);
    const float m = (mask != nullptr) ? mask[pix] : 1.f;
    r0 *= m;
    r1 *= m;
    r2 *= m;
    xw[idx] = make_float4(r0, r1, r2, 0.f);
    if (cat8 != nullptr) {
        const float4 sv = s[idx];
        cat8[2 * (size_t)idx] = make_float4(sv.x, sv.y, sv.z, r0 * sv.x);
        cat8[2 * (size_t)idx + 1] = make_float4(r1 * sv.y, r2 * sv.z, 0.f, 0.f);
    }
}

// The same forward pass from the per-attack TAP TABLE (round 6): the four source pixels and weights (x mask) of every camera pixel are
// computed ONCE by warp_taps_kernel -- the table the deterministic backward pass is built from, so forward and backward use the same
// weights to the bit (an exact adjoint pair) and no call site re-derives the coordinate arithmetic.  A workgroup owns a 32 x 8 tile of
// camera pixels (its taps fall into a ~30 x 9 box of the projector image: the 1-D blocks of warp_fwd_kernel fetched every projector row
// from two workgroups on different XCDs: 217 MB of fabric traffic for 100 MB of algorithmic bytes, profiles/r05_pmc_traffic.json) and
// FB images, for which the table entry is read once; consecutive tiles run on the same XCD and share their halo in its L2.
constexpr int FT_W = 32, FT_H = 8, FB = 4;
__global__ __launch_bounds__(256) void warp_fwd_taps_kernel(const float4* __restrict__ x, const int4* __restrict__ src, const float4* __restrict__ wgt,
                                                            float4* __restrict__ xw, int B, int HWp, int Hc, int Wc, int ntx, int ntile,
                                                            int clamp) {
    int t;
    {
        const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tile = t % ntile, b0 = (t / ntile) * FB;
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int cy = ty * FT_H + (threadIdx.x >> 5), cx = tx * FT_W + (threadIdx.x & 31);
    if (cy >= Hc || cx >= Wc) return;
    const int pix = cy * Wc + cx;
    const int4 s = src[pix];
    const float4 w = wgt[pix];
    const int si[4] = {s.x, s.y, s.z, s.w};
    const float wi[4] = {w.x, w.y, w.z, w.w};
    float4 v[FB][4];
#pragma unroll
    for (int k = 0; k < FB; ++k) {
        const float4* xb = x + (size_t)(b0 + k < B ? b0 + k : B - 1) * HWp;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[k][q] = si[q] != 0x7fffffff ? xb[si[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < FB; ++k) {
        if (b0 + k >= B) break;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 u = v[k][q];
            if (clamp) {
                u.x = clamp01(u.x);
                u.y = clamp01(u.y);
                u.z = clamp01(u.z);
            }
            r0 += u.x * wi[q];
            r1 += u.y * wi[q];
            r2 += u.z * wi[q];
        }
        xw[(size_t)(b0 + k) * Hc * Wc + pix] = make_float4(r0, r1, r2, 0.f);
    }
}

// Backward w.r.t. the projector image: scatter-add of the four bilinear taps.  The clamp(x,0,1) of the forward
// passes gradient where 0 <= x <= 1 (ATen clamp_backward).
__global__ void warp_bwd_kernel(const float4* __restrict__ g_xw, const float4* __restrict__ g_xs,
                                const float4* __restrict__ x, const float4* __restrict__ grid,
                                const float* __restrict__ mask, const float4* __restrict__ s, float* __restrict__ g_x,
                                int B, int Hp, int Wp, int HWc, int clamp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HWc) return;
    const int b = idx / HWc, pix = idx - b * HWc;
    float4 g = g_xw[idx];
    if (g_xs != nullptr) {
        const float4 gs = g_xs[idx];
        const float4 sv = s[idx];
        g.x += gs.x * sv.x;
        g.y += gs.y * sv.y;
        g.z += gs.z * sv.z;
    }
    const float m = (mask != nullptr) ? mask[pix] : 1.f;
    g.x *= m;
    g.y *= m;
    g.z *= m;
    const float4 gr = grid[pix];
    const Bilinear bl = bilinear_setup(gr.x, gr.y, Wp, Hp);
    const size_t base = (size_t)b * Hp * Wp;
    auto tap = [&](int yy, int xx, float wgt) {
        if ((unsigned)yy < (unsigned)Hp && (unsigned)xx < (unsigned)Wp) {
            const size_t o = base + (size_t)(yy * Wp + xx);
            bool p0 = true, p1 = true, p2 = true;
            if (clamp) {
                const float4 v = x[o];
                p0 = (v.x >= 0.f && v.x <= 1.f);
                p1 = (v.y >= 0.f && v.y <= 1.f);
                p2 = (v.z >= 0.f && v.z <= 1.f);
            }
            float* dst = g_x + 4 * o;
            if (p0) atomicAdd(dst + 0, g.x * wgt);
            if (p1) atomicAdd(dst + 1, g.y * wgt);
            if (p2) atomicAdd(dst + 2, g.z * wgt);
        }
    };
    tap(bl.y0, bl.x0, bl.nw);
    tap(bl.y0, bl.x0 + 1, bl.ne);
    tap(bl.y0 + 1, bl.x0, bl.sw);
    tap(bl.y0 + 1, bl.x0 + 1, bl.se);
}

// Per (camera pixel, bilinear tap): the projector pixel it reads (or -1) and its weight.  The grid is constant during an
// attack, so the transposed (projector pixel -> list of contributions) structure is built ONCE from this table
// (stable sort by source index; spaa_amd/models.py) and the backward pass becomes a deterministic gather.
__global__ void warp_taps_kernel(const float4* __restrict__ grid, int Hp, int Wp, int HWc, int32_t* __restrict__ src,
                                 float* __restrict__ wgt) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HWc) return;
    const float4 g = grid[pix];
    const Bilinear bl = bilinear_setup(g.x, g.y, Wp, Hp);
    const int ys[4] = {bl.y0, bl.y0, bl.y0 + 1, bl.y0 + 1};
    const int xs[4] = {bl.x0, bl.x0 + 1, bl.x0, bl.x0 + 1};
    const float ws[4] = {bl.nw, bl.ne, bl.sw, bl.se};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool ok = (unsigned)ys[t] < (unsigned)Hp && (unsigned)xs[t] < (unsigned)Wp;
        src[4 * pix + t] = ok ? ys[t] * Wp + xs[t] : 0x7fffffff;
        wgt[4 * pix + t] = ok ? ws[t] : 0.f;
    }
}

// g_x[b, s] = clampgate(x) * sum_{e in list(s)} w_e * ((g_xw + g_xs * scene) * mask)[b, campix_e]
// order[e] = 4*campix + tap (entries sorted by source pixel, ties in camera-pixel order), off[s]..off[s+1] the list of s.
// A thread owns source pixel s of GB consecutive images: the tap list (identical for every image of the batch) is read once
// for them, and the GB gathers of an entry are independent loads in flight.
constexpr int GB = 4;
__global__ void warp_bwd_gather_kernel(const float4* __restrict__ g_xw, const float4* __restrict__ g_xs,
                                       const float4* __restrict__ x, const float* __restrict__ mask,
                                       const float4* __restrict__ s, const int32_t* __restrict__ off,
                                       const int32_t* __restrict__ order, const float* __restrict__ wgt,
                                       float4* __restrict__ g_x, int B, int HWp, int HWc, int clamp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nbg = (B + GB - 1) / GB;
    if (idx >= nbg * HWp) return;
    const int bg = idx / HWp, sp = idx - bg * HWp;
    const int b0 = bg * GB;
    const int e0 = off[sp], e1 = off[sp + 1];
    float a0[GB], a1[GB], a2[GB];
#pragma unroll
    for (int k = 0; k < GB; ++k) a0[k] = a1[k] = a2[k] = 0.f;
    for (int e = e0; e < e1; ++e) {
        const int ent = order[e];
        const int cp = ent >> 2;
        const float w = wgt[ent] * ((mask != nullptr) ? mask[cp] : 1.f);
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            if (b0 + k >= B) break;
            const size_t ci = (size_t)(b0 + k) * HWc + cp;
            float4 g = g_xw[ci];
            if (g_xs != nullptr) {
                const float4 gs = g_xs[ci];
                const float4 sv = s[ci];
                g.x += gs.x * sv.x;
                g.y += gs.y * sv.y;
                g.z += gs.z * sv.z;
            }
            a0[k] += g.x * w;
            a1[k] += g.y * w;
            a2[k] += g.z * w;
        }
    }
#pragma unroll
    for (int k = 0; k < GB; ++k) {
        if (b0 + k >= B) break;
        const size_t o = (size_t)(b0 + k) * HWp + sp;
        float r0 = a0[k], r1 = a1[k], r2 = a2[k];
        if (clamp) {
            const float4 v = x[o];
            r0 = (v.x >= 0.f && v.x <= 1.f) ? r0 : 0.f;
            r1 = (v.y >= 0.f && v.y <= 1.f) ? r1 : 0.f;
            r2 = (v.z >= 0.f && v.z <= 1.f) ? r2 : 0.f;
        }
        g_x[o] = make_float4(r0, r1, r2, 0.f);
    }
}

// The same gather with the camera-side operand staged through LDS.  A workgroup owns a TS x TS tile of projector pixels and
// WB images; the camera pixels its tap lists touch lie in a bounding box (the warp is smooth: ~21 x 21 pixels for a
// 16 x 16 tile at the bench's 0.9 scale) that is read ONCE per image with coalesced rows, instead of every camera pixel
// being fetched by the four projector pixels that sample it (PMC r02: 557 MB of fabric traffic for 100 MB of algorithmic
// bytes).  Per tap-list entry the host stores the LDS index inside the tile's box and the weight (x mask) in entry order.
// Same summation order per projector pixel as warp_bwd_gather_kernel: results are bitwise equal.
constexpr int TS = 16, WB = 4;
__global__ __launch_bounds__(256) void warp_bwd_tiled_kernel(const float4* __restrict__ g_xw, const float4* __restrict__ x,
                                                             const int32_t* __restrict__ off, const int32_t* __restrict__ lidx,
                                                             const float* __restrict__ w_e, const int32_t* __restrict__ tbox,
                                                             float4* __restrict__ g_x, int B, int Hp, int Wp, int Hc, int Wc,
                                                             int ntx, int box_cap, int clamp, float* __restrict__ partial_ss,
                                                             const int32_t* __restrict__ state, float gray, float prjl2_scale,
                                                             const uint8_t* __restrict__ clamp_bits) {
    extern __shared__ __attribute__((aligned(16))) float4 box[];   // [WB][box_cap]
    const int tile = blockIdx.x, b0 = blockIdx.y * WB;
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int cy0 = tbox[4 * tile], cx0 = tbox[4 * tile + 1], ch = tbox[4 * tile + 2], cw = tbox[4 * tile + 3];
    // (a tile whose box does not fit -- clamped grids pile camera pixels of a whole border strip onto the projector's border
    // pixels -- has rows = -1: its entries carry the camera pixel itself and are gathered from global memory)
    const bool direct = ch < 0;
    const int npx = direct ? 0 : ch * cw;
    const size_t HWc = (size_t)Hc * Wc, HWp = (size_t)Hp * Wp;
    // Every load that does not depend on the staged box is requested BEFORE the box is staged: the list bounds, the first four
    // list entries and the clamp gate's operand -- four dependent round trips per workgroup (box, bounds, entries, gate) become two
    // (bounds, then box + entries + gate together).  Same arithmetic in the same order.
    const int sy = ty * TS + (threadIdx.x >> 4), sx = tx * TS + (threadIdx.x & 15);
    const bool live = sy < Hp && sx < Wp;
    const int sp = live ? sy * Wp + sx : 0;
    const int e0 = live ? off[sp] : 0, e1 = live ? off[sp + 1] : 0;
    float4 xv[WB];
    // (`clamp_bits`: the gate's three comparisons as one byte per pixel from the kernel that wrote x -- spaa_step_and_track_n --: 1 byte
    // per pixel instead of x's 16; x itself only for the prjl2 term)
    const bool need_x = (clamp && clamp_bits == nullptr) || (partial_ss != nullptr && prjl2_scale != 0.f);
    unsigned int cb[WB];
#pragma unroll
    for (int k = 0; k < WB; ++k) {
        const size_t o = (size_t)(b0 + k < B ? b0 + k : B - 1) * HWp + sp;
        xv[k] = (need_x && live) ? x[o] : make_float4(0.f, 0.f, 0.f, 0.f);
        cb[k] = (clamp && clamp_bits != nullptr && live) ? clamp_bits[o] : 7u;
    }
    int li0[4];
    float w0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ee = e0 + q < e1 ? e0 + q : (e1 > e0 ? e1 - 1 : 0);
        li0[q] = e1 > e0 ? lidx[ee] : 0;
        w0[q] = e1 > e0 ? w_e[ee] : 0.f;
    }
    for (int i = threadIdx.x; i < npx; i += 256) {
        const int r = i / cw, c = i - r * cw;
        const size_t cp = (size_t)(cy0 + r) * Wc + (cx0 + c);
#pragma unroll
        for (int k = 0; k < WB; ++k)
            box[k * box_cap + i] = g_xw[(size_t)(b0 + k < B ? b0 + k : B - 1) * HWc + cp];   // (images past B: a copy of the last, never stored)
    }
    __syncthreads();
    if (!live && partial_ss == nullptr) return;
    float a0[WB], a1[WB], a2[WB];
#pragma unroll
    for (int k = 0; k < WB; ++k) a0[k] = a1[k] = a2[k] = 0.f;
    // four list entries per round, their (index, weight) pairs loaded together (entries past the list: the last one again with
    // weight 0 -- a dependent global round trip per entry otherwise; 3.2 entries per pixel on average); the first round's were
    // requested above
    for (int e = e0; e < e1; e += 4) {
        int li[4];
        float w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (e == e0) {
                li[q] = li0[q];
                w[q] = w0[q];
            } else {
                const int ee = e + q < e1 ? e + q : e1 - 1;
                li[q] = lidx[ee];
                w[q] = w_e[ee];
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (e + q >= e1) break;
#pragma unroll
            for (int k = 0; k < WB; ++k) {
                float4 g;
                if (direct) g = g_xw[(size_t)(b0 + k < B ? b0 + k : b0) * HWc + li[q]];
                else g = box[k * box_cap + li[q]];   // (images past B: stale LDS, never stored)
                a0[k] += g.x * w[q];
                a1[k] += g.y * w[q];
                a2[k] += g.z * w[q];
            }
        }
    }
    float ss[WB];
#pragma unroll
    for (int k = 0; k < WB; ++k) {
        ss[k] = 0.f;
        if (b0 + k >= B || !live) continue;
        const size_t o = (size_t)(b0 + k) * HWp + sp;
        float r0 = a0[k], r1 = a1[k], r2 = a2[k];
        const float4 v = xv[k];
        if (clamp && clamp_bits != nullptr) {
            r0 = (cb[k] & 1u) ? r0 : 0.f;
            r1 = (cb[k] & 2u) ? r1 : 0.f;
            r2 = (cb[k] & 4u) ? r2 : 0.f;
        } else if (clamp) {
            r0 = (v.x >= 0.f && v.x <= 1.f) ? r0 : 0.f;
            r1 = (v.y >= 0.f && v.y <= 1.f) ? r1 : 0.f;
            r2 = (v.z >= 0.f && v.z <= 1.f) ? r2 : 0.f;
        }
        if (partial_ss != nullptr) {
            // spaa_grad_sumsq folded in (round 6): the prjl2 term's gradient for samples taking the colour step
            // (projector_based_attack.py:275,310: d ||gray - x|| / dx = -(gray - x) / n, zero where the norm is zero), then this
            // pixel's share of ||g_b||^2 -- the same arithmetic, one launch and one pass over g_x less per iteration
            if (prjl2_scale != 0.f && state[4 * (b0 + k) + 1] != 0) {
                const float d0 = gray - v.x, d1 = gray - v.y, d2 = gray - v.z;
                const float n = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
                if (n != 0.f) {
                    const float kk = -prjl2_scale / n;
                    r0 += kk * d0;
                    r1 += kk * d1;
                    r2 += kk * d2;
                }
            }
            ss[k] = r0 * r0 + r1 * r1 + r2 * r2;
        }
        g_x[o] = make_float4(r0, r1, r2, 0.f);
    }
    if (partial_ss != nullptr) {
        // per (image, tile) partial sums in a fixed order: wave shuffles, then the four waves' sums through LDS (the box is free)
        __syncthreads();
        float* red = reinterpret_cast<float*>(box);
#pragma unroll
        for (int k = 0; k < WB; ++k) {
            float v = ss[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * WB + k] = v;
        }
        __syncthreads();
        if (threadIdx.x < WB && b0 + threadIdx.x < B)
            partial_ss[(size_t)(b0 + threadIdx.x) * gridDim.x + tile] =
                red[threadIdx.x] + red[WB + threadIdx.x] + red[2 * WB + threadIdx.x] + red[3 * WB + threadIdx.x];
    }
}

__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ src, float4* __restrict__ dst, int B, int HW,
                                     int clamp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW, pix = idx - b * HW;
    const float* p = src + (size_t)b * 3 * HW + pix;
    float r = p[0], g = p[HW], bl = p[2 * (size_t)HW];
    if (clamp) {
        r = clamp01(r);
        g = clamp01(g);
        bl = clamp01(bl);
    }
    dst[idx] = make_float4(r, g, bl, 0.f);
}

__global__ void nhwc4_to_nchw_kernel(const float4* __restrict__ src, float* __restrict__ dst, int B, int HW,
                                     int clamp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW, pix = idx - b * HW;
    float4 v = src[idx];
    if (clamp) {
        v.x = clamp01(v.x);
        v.y = clamp01(v.y);
        v.z = clamp01(v.z);
    }
    float* p = dst + (size_t)b * 3 * HW + pix;
    p[0] = v.x;
    p[HW] = v.y;
    p[2 * (size_t)HW] = v.z;
}

inline int blocks_for(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

}  // namespace

extern "C" {

int spaa_warp_coarse_grid(const float* affine6, const float* theta, const float* ctrl, int T, int Hin, int Win,
                          int Hout, int Wout, float* out, spaa_stream_t stream) {
    if (!affine6 || !theta || !ctrl || !out || T < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(coarse_grid_kernel, dim3(blocks_for((int64_t)Hout * Wout, 256)), dim3(256), 0,
                       (hipStream_t)stream, affine6, theta, ctrl, T, Hin, Win, Hout, Wout, out);
    return (int)hipGetLastError();
}

int spaa_warp_finish_grid(const float* coarse, const float* refine, float* fine, int npix, spaa_stream_t stream) {
    if (!coarse || !fine || npix < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(finish_grid_kernel, dim3(blocks_for(npix, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)coarse, (const float4*)refine, (float4*)fine, npix);
    return (int)hipGetLastError();
}

int spaa_warp_fwd(const float* x, const float* grid, const float* mask, const float* s, float* xw, float* cat8, int B,
                  int Hp, int Wp, int Hc, int Wc, int clamp, spaa_stream_t stream) {
    if (!x || !grid || !xw || (cat8 && !s) || B < 1 || Hp < 1 || Wp < 1 || Hc < 1 || Wc < 1)
        return hipErrorInvalidValue;
    if ((int64_t)B * Hc * Wc >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_fwd_kernel, dim3(blocks_for((int64_t)B * Hc * Wc, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)x, (const float4*)grid, mask, (const float4*)s,
                       (float4*)xw, (float4*)cat8, B, Hp, Wp, Hc * Wc, clamp);
    return (int)hipGetLastError();
}

int spaa_warp_bwd(const float* g_xw, const float* g_xs, const float* x, const float* grid, const float* mask,
                  const float* s, float* g_x, int B, int Hp, int Wp, int Hc, int Wc, int clamp, spaa_stream_t stream) {
    if (!g_xw || !x || !grid || !g_x || (g_xs && !s) || B < 1 || Hp < 1 || Wp < 1 || Hc < 1 || Wc < 1)
        return hipErrorInvalidValue;
    if ((int64_t)B * Hc * Wc >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(blocks_for((int64_t)B * Hc * Wc, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)g_xw, (const float4*)g_xs, (const float4*)x,
                       (const float4*)grid, mask, (const float4*)s, g_x, B, Hp, Wp, Hc * Wc, clamp);
    return (int)hipGetLastError();
}

int spaa_warp_taps(const float* grid, int Hp, int Wp, int Hc, int Wc, int32_t* src, float* wgt, spaa_stream_t stream) {
    if (!grid || !src || !wgt || Hp < 1 || Wp < 1 || Hc < 1 || Wc < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_taps_kernel, dim3(blocks_for((int64_t)Hc * Wc, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)grid, Hp, Wp, Hc * Wc, src, wgt);
    return (int)hipGetLastError();
}

int spaa_warp_bwd_gather(const float* g_xw, const float* g_xs, const float* x, const float* mask, const float* s,
                         const int32_t* off, const int32_t* order, const float* wgt, float* g_x, int B, int Hp, int Wp,
                         int Hc, int Wc, int clamp, spaa_stream_t stream) {
    if (!g_xw || !x || !off || !order || !wgt || !g_x || (g_xs && !s) || B < 1 || Hp < 1 || Wp < 1 || Hc < 1 || Wc < 1)
        return hipErrorInvalidValue;
    if ((int64_t)B * Hp * Wp >= ((int64_t)1 << 31) || (int64_t)B * Hc * Wc >= ((int64_t)1 << 31))
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_bwd_gather_kernel, dim3(blocks_for((int64_t)((B + GB - 1) / GB) * Hp * Wp, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)g_xw, (const float4*)g_xs, (const float4*)x, mask,
                       (const float4*)s, off, order, wgt, (float4*)g_x, B, Hp * Wp, Hc * Wc, clamp);
    return (int)hipGetLastError();
}

static int launch_warp_bwd_tiled(const float* g_xw, const float* x, const int32_t* off, const int32_t* lidx, const float* w_e,
                                 const int32_t* tbox, int box_cap, float* g_x, int B, int Hp, int Wp, int Hc, int Wc, int clamp,
                                 float* partial_ss, const int32_t* state, float gray, float prjl2_scale, const uint8_t* clamp_bits,
                                 spaa_stream_t stream) {
    if (!g_xw || !x || !off || !lidx || !w_e || !tbox || !g_x || B < 1 || Hp < 1 || Wp < 1 || Hc < 1 || Wc < 1 || box_cap < 1 ||
        (size_t)box_cap * WB * 16 > 64 * 1024)
        return hipErrorInvalidValue;
    if ((int64_t)B * Hp * Wp >= ((int64_t)1 << 31) || (int64_t)B * Hc * Wc >= ((int64_t)1 << 31))
        return hipErrorInvalidValue;
    const int ntx = (Wp + TS - 1) / TS, nty = (Hp + TS - 1) / TS;
    size_t smem = (size_t)box_cap * WB * 16;
    if (partial_ss != nullptr && smem < 4 * WB * sizeof(float)) smem = 4 * WB * sizeof(float);
    hipLaunchKernelGGL(warp_bwd_tiled_kernel, dim3(ntx * nty, (B + WB - 1) / WB), dim3(256), smem,
                       (hipStream_t)stream, (const float4*)g_xw, (const float4*)x, off, lidx, w_e, tbox, (float4*)g_x, B, Hp, Wp,
                       Hc, Wc, ntx, box_cap, clamp, partial_ss, state, gray, prjl2_scale, clamp_bits);
    return (int)hipGetLastError();
}

int spaa_warp_bwd_tiled(const float* g_xw, const float* x, const int32_t* off, const int32_t* lidx, const float* w_e,
                        const int32_t* tbox, int box_cap, float* g_x, int B, int Hp, int Wp, int Hc, int Wc, int clamp,
                        spaa_stream_t stream) {
    return launch_warp_bwd_tiled(g_xw, x, off, lidx, w_e, tbox, box_cap, g_x, B, Hp, Wp, Hc, Wc, clamp, nullptr, nullptr, 0.f, 0.f, nullptr, stream);
}

int spaa_warp_bwd_tiled_sumsq(const float* g_xw, const float* x, const int32_t* off, const int32_t* lidx, const float* w_e,
                              const int32_t* tbox, int box_cap, float* g_x, int B, int Hp, int Wp, int Hc, int Wc, int clamp,
                              float gray, float prjl2_scale, const int32_t* state, float* partial_ss, const uint8_t* clamp_bits,
                              spaa_stream_t stream) {
    if (!partial_ss || (prjl2_scale != 0.f && !state)) return hipErrorInvalidValue;
    return launch_warp_bwd_tiled(g_xw, x, off, lidx, w_e, tbox, box_cap, g_x, B, Hp, Wp, Hc, Wc, clamp, partial_ss, state, gray, prjl2_scale,
                                 clamp_bits, stream);
}

int spaa_warp_fwd_taps(const float* x, const int32_t* tap_src, const float* tap_wgt, float* xw, int B, int Hp, int Wp, int Hc, int Wc,
                       int clamp, spaa_stream_t stream) {
    if (!x || !tap_src || !tap_wgt || !xw || B < 1 || Hp < 1 || Wp < 1 || Hc < 1 || Wc < 1) return hipErrorInvalidValue;
    if ((int64_t)B * Hc * Wc >= ((int64_t)1 << 31) || (int64_t)B * Hp * Wp >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    const int ntx = (Wc + FT_W - 1) / FT_W, nty = (Hc + FT_H - 1) / FT_H;
    const int64_t nwg = (int64_t)ntx * nty * ((B + FB - 1) / FB);
    if (nwg > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_fwd_taps_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       (const int4*)tap_src, (const float4*)tap_wgt, (float4*)xw, B, Hp * Wp, Hc, Wc, ntx, ntx * nty, clamp);
    return (int)hipGetLastError();
}

int spaa_nchw_to_nhwc4(const float* src, float* dst, int B, int H, int W, int clamp, spaa_stream_t stream) {
    if (!src || !dst || B < 1 || H < 1 || W < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(blocks_for((int64_t)B * H * W, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, (float4*)dst, B, H * W, clamp);
    return (int)hipGetLastError();
}

int spaa_nhwc4_to_nchw(const float* src, float* dst, int B, int H, int W, int clamp, spaa_stream_t stream) {
    if (!src || !dst || B < 1 || H < 1 || W < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nhwc4_to_nchw_kernel, dim3(blocks_for((int64_t)B * H * W, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)src, dst, B, H * W, clamp);
    return (int)hipGetLastError();
}

int spaa_zero(void* p, int64_t bytes, spaa_stream_t stream) {
    if (!p || bytes < 0) return hipErrorInvalidValue;
    return (int)hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
}

const char* spaa_version(void) { return "spaa_hip 0.6 (gfx950)"; }

}  // extern "C"
