// tapconv_wino.hip — 3x3 / stride-1 / pad-1 convolutions (and their input gradients) by Winograd F(2x2, 3x3) on the bf16x6
// matrix-core arithmetic of tapconv_x6d.hip: 16 multiplications per 2x2 output tile instead of 36.
//
// The six 64x64 x (128 <-> 256 channel) layers of ShadingNetSPAA (conv4, conv5, conv4_s and their input gradients: 69 % of
// PCNet's MACs, /root/reference/src/python/models.py:286-298) sit at the chip's power wall in the direct form (DESIGN.md
// section 3): fewer products is the only lever left there.
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        g: 3x3 filter, d: 4x4 input patch (stride 2), Y: 2x2 outputs
//   * U = G g G^T is computed in fp64, rounded ONCE to fp32 and split exactly into three bf16 planes, packed like a 16-tap
//     weight matrix  W[n][pos * Cin + c],  pos = 4 xi + nu  (spaa_amd/convplan.py: attach_winograd); its rows are de-tuned by
//     256 B: rows a multiple of 4 KiB apart all map to one L2 channel (measured: 16x slower);
//   * V = B^T d B (entries are sums / differences of four inputs: +-1 coefficients, fp32 adds) is formed in registers from a
//     patch of the input staged ONCE per 32-channel block in LDS by LDS-DMA (out-of-image pixels: the out-of-range offset,
//     zeros = the convolution's zero padding; pixel-pair / chunk swizzle: conflict-free reads), split into three bf16
//     fragments like every other activation operand;
//   * per position one 32-deep product  M_pos = U_pos . V_pos  (6 bf16 MFMAs per 16x16 block, fp32 accumulation) is folded
//     into the four output accumulators with A^T . A's +-1 coefficients, in place between the MFMAs of the next block (the
//     accumulators are pinned to fixed registers; the xi / nu loops are expanded so that the coefficients are immediates);
//   * a workgroup (8 waves) owns 8 x 16 Winograd tiles (16 x 32 output pixels) of one image x BN = 128 or 64 output channels;
//     wave w owns tile row w (16 tiles = the 16 columns of the MFMA's B operand); the U planes of a (position, channel block)
//     are shared by the workgroup: 24 KB per step, THREE LDS stages (DMA two steps ahead), one barrier per step; the next
//     block's patch is requested as soon as the last row combination of the current one has been read;
//   * epilogue through LDS (a store instruction writes whole 512-byte channel rows) and the shared store4 (bias, residual,
//     activation, gates, byte masks), the residual / gate operands of eight pixels in flight.
// Accuracy: measured against fp64 the error is BELOW that of the direct kernels (2.6e-7 vs 1.0e-6 relative L-inf on conv4: 128
// instead of 1152 terms per fp32 accumulation chain); gated by tests/test_gpu_parity.py::test_winograd_*.
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ unsigned int cvt2(float a, float b) {
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned int p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned int p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 fp32 -> three bf16x8 with x == h + m + l exactly
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u4 hh, mm, ll;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned int ph = cvt2(x[2 * i], x[2 * i + 1]);
        const float r0 = x[2 * i] - lo_f(ph), r1 = x[2 * i + 1] - hi_f(ph);
        const unsigned int pm = cvt2(r0, r1);
        const float s0 = r0 - lo_f(pm), s1 = r1 - hi_f(pm);
        hh[i] = ph;
        mm[i] = pm;
        ll[i] = cvt2(s0, s1);
    }
    h = __builtin_bit_cast(bf16x8, hh);
    m = __builtin_bit_cast(bf16x8, mm);
    l = __builtin_bit_cast(bf16x8, ll);
}

constexpr int TY = 8, TX = 16;                 // Winograd tiles per workgroup (rows x columns): 16 x 32 output pixels
constexpr int PH = 2 * TY + 2, PW = 2 * TX + 2;  // input patch 18 x 34 pixels
constexpr int NPX = PH * PW;                   // 612
constexpr int NPIECE = (NPX + 7) / 8;          // 1-KiB pieces of 8 pixels x 32 channels (fp32): 77
constexpr int PPW = (NPIECE + 7) / 8;          // pieces per wave (8 waves): 10
constexpr int PATCH_BYTES = PPW * 8 * 1024;    // 81920 (pieces 77..79: pad)

// 64-byte weight rows (32 bf16), chunk swizzle as tapconv_x6d.hip swz_w<16>
__device__ __forceinline__ int swz_w16(int n) { return ((n >> 3) & 1) << 1; }

// workgroup barrier that leaves LDS-DMA in flight: __syncthreads() drains vmcnt to 0 while a buffer_load ... lds is pending
// (cdna_hip_programming.md section 5, "Pipelining across barriers"), which defeats the counted waits of the weight pipeline
template <bool RAW>
__device__ __forceinline__ void wg_barrier() {
    if constexpr (RAW) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
        __syncthreads();
    }
}

// Small images (ResNet layer3 / layer4: 14 x 14 and 7 x 7; Inception-v3 35 x 35; VGG-16 14 x 14) would leave most of a 16 x 32-pixel
// workgroup region empty.  CANVAS mode lays the images of the batch out on virtual canvases -- gy x gx images with periods
// (py, px) >= (H + 1, W + 1): the gap row / column between two images is the zero padding of both -- and the workgroup regions tile
// the CANVAS: Winograd tiles that straddle a gap compute a discarded output next to a real one, nothing else changes (the patch
// DMA maps a canvas pixel to (image, y, x) or to the out-of-range offset, the epilogue maps it back).  Few workgroups with long K
// (layer4: 8 regions x 4 N tiles, 16 channel blocks) are cut along K: split ks computes channel blocks [ks * kb_per, ...) into
// the fp32 workspace [split][pixel][Npad], wino_splitk_reduce_kernel adds the splits in fixed order and applies the epilogue.
struct wino_geo_t {
    int ksplit, kb_per;        // K ranges (1 = off) and 32-channel blocks per range
    int gy, gx, py, px;        // canvas: images per canvas (rows x columns) and their periods in pixels
    unsigned int my, mx;       // v / py == (v * my) >> 20 for every canvas coordinate v (launcher: canvas sides <= 4095, periods <= 255)
    int nsp;                   // workgroup regions of all canvases together
    int order;                 // 1: regions fastest in the workgroup order, 0: N tiles fastest (as in the image-aligned form)
    int fix;                   // K ranges: 1 = the last-arriving workgroup of a (region, N tile) sums the partial sums and applies the
                               // epilogue itself (arrival counters at the head of the workspace), 0 = wino_splitk_reduce_kernel does
};

// TWO: the layer's input channels come from TWO tensors of the same B x H x W (channel blocks [0, Cin - Cin2) from `in`, the rest from
// `in2`): conv(a, Wa) + conv(b, Wb) as ONE convolution over the concatenated channels without the concatenated tensor --
// ShadingNetSPAA's `conv5(x4) + skipConv3(x2)` (models.py:294,298) and its mirror image in the backward pass.
// NWT = 4 (tile 73, 64-wide N tile): a workgroup of FOUR waves (4 x 16 tiles = 8 x 32 output pixels, 80 KB of LDS), TWO workgroups per
// compute unit -- the two waves of a SIMD belong to different workgroups, so one's prologue / epilogue runs under the other's main
// loop (short-K layers: ResNet-18 layer1 spends a third of a launch in them).
__host__ __device__ __forceinline__ int wino_pad(const int reserved0) {
    const int c = (reserved0 >> 27) & 3;
    return c == 0 ? 1 : (c == 1 ? 0 : 2);
}
template <int BN, int VAR, int DBG = 0, bool CV = false, bool TWO = false, int NWT = 8>   // DBG: timing-only ablations (wrong results): 1 no epilogue, 2 no fold, 4 no V, 8 no barrier, 16 no DMA; 32: raw barrier.  CV: canvas / split-K form
__global__ __launch_bounds__(64 * NWT, NWT == 4 ? 2 : 1) void wino_x6_kernel(const spaa_tapconv_t p, const int wg_y, const int wg_x, const int n_tiles, const wino_geo_t geo) {
    constexpr int TJ = BN / 16;
    constexpr int W_PLANE = BN * 64;
    constexpr int NW = NWT;
    // (these shadow the 8-wave constants of the same names above)
    constexpr int TY = NWT, PH = 2 * TY + 2, NPX = PH * PW, NPIECE = (NPX + 7) / 8, PPW = (NPIECE + NW - 1) / NW, PATCH_BYTES = PPW * NW * 1024;
    static_assert(NWT == 8 || (NWT == 4 && !(VAR & 1)), "the late-V variants pair waves w and w + 4");
    constexpr int W_PIECES = 3 * BN / 16;
    constexpr int WPW = (W_PIECES + NW - 1) / NW;
    constexpr int WS_BYTES = WPW * NW * 1024;   // (every wave issues WPW DMAs per step: pieces past the planes land in the pad)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* wsm = smem + PATCH_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const spaa_tapclass_t cl = p.cls[0];
    const int Cin = p.Cin, H = p.Hin, W = p.Win;
    // zero padding of the layer: 1 (same size), 0 (unpadded: the output is 2 smaller) or 2 (the unpadded layer's input gradient: 2 larger);
    // `reserved0` bits 27-28 = 0 / 1 / 2.  The canvas / K-range form (CV) shares its gaps as padding: pad 1 only (the launcher sees to it).
    const int pad_ = CV ? 1 : wino_pad(p.reserved0);

    // XCD-aware order over (image, patch row, patch column, n tile): an XCD takes a contiguous range
    int n_blk, img, oy0, ox0, ks = 0;   // (CV: img = the canvas, (oy0, ox0) = the region's origin on it)
    int region = 0;                     // (CV: index of the region among all canvases' regions)
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        if constexpr (CV) {
            if (geo.order) {
                // regions fastest: the workgroups of an XCD share ONE (N tile, K range) slice of the weights (layer4: 25 MB of planes
                // against 4 MB of L2 per XCD, 6 MB of activations)
                const int combo = t / geo.nsp;
                t -= combo * geo.nsp;
                n_blk = (combo % n_tiles) * BN;
                ks = combo / n_tiles;
            } else {
                // (N tile, K range) fastest: the workgroups of a region, which read the same patch, sit next to each other
                const int nc = n_tiles * geo.ksplit, combo = t % nc;
                t /= nc;
                n_blk = (combo % n_tiles) * BN;
                ks = combo / n_tiles;
            }
        } else {
            n_blk = (t % n_tiles) * BN;
            t /= n_tiles;
        }
        region = t;
        ox0 = (t % wg_x) * (2 * TX);
        t /= wg_x;
        oy0 = (t % wg_y) * (2 * TY);
        img = t / wg_y;
    }
    const int row_bytes = p.in_cstride * 4;
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(H * W) * (uint32_t)row_bytes;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int kb1 = TWO ? (Cin - p.Cin2) >> 5 : 0x7fffffff;   // first channel block of the second source
    const int row_bytes2 = TWO ? p.in2_cstride * 4 : 0;
    const auto rsrc_in2 = TWO ? rsrc_or_empty(p.in2, (int64_t)p.B * (H * W) * row_bytes2) : rsrc_in;
    const int npad = (p.Cout + 127) & ~127;
    const int plane_bytes = npad * cl.Kpad * 2;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_split);
    const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane(3u * (uint32_t)plane_bytes), 0x00020000);

    // ---- patch staging: piece i (8 consecutive pixel SLOTS) -> wave i % 8; lane -> (slot lane >> 3, physical chunk lane & 7).
    // LDS layout (bank-conflict-free for a wave's reads: 16 tiles = patch columns 2 apart, 4 channel chunks):
    //   slot of patch pixel p = row * PW + column:  p ^ ((p >> 1) & 1)   (neighbouring pixel pairs alternate their order, so
    //   pixels 2 apart alternate between the two 128-byte halves of the 64 banks);
    //   16-byte chunk L of the pixel's 32 channels sits at chunk  L ^ ((column >> 2) & 7).
    // Every wave issues exactly PPW DMAs per block (s_waitcnt counts): pieces past the patch read the out-of-range offset
    // into the pad at the end of the patch buffer.
    const int pa_base = p.in_coff * 4 + (lane & 7) * 16;   // (chunk swizzle applied per pixel below)
    // ---- weight staging (as tapconv_x6d.hip): piece q = wave + 8 i -> (plane, 16-row block); lane -> (row, physical chunk)
    int w_goff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int q = wave + NW * i;
        const int pl = q / (BN / 16), rb = q % (BN / 16);
        const int n = 16 * rb + (lane >> 2);
        const int c = (lane & 3) ^ swz_w16(n);
        w_goff[i] = q < W_PIECES ? pl * plane_bytes + (n_blk + n) * cl.Kpad * 2 + c * 16 : (int)0x80000000;
    }
    const int w_addr_l = (lane & 15) * 64 + (((lane >> 4) ^ swz_w16(lane & 15)) * 16);

    // ---- this lane's tile: row `wave`, column lane & 15, channels 8 (lane >> 4) .. + 7 of the block
    const int tx = lane & 15, q8 = lane >> 4;

    f32x4 Y[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) Y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 32-channel blocks of this workgroup: all of them, or (CV) the K range [kb0, kb0 + nkb)
    const int kb0 = CV ? ks * geo.kb_per : 0;
    const int nkb = CV ? ((Cin >> 5) - kb0 < geo.kb_per ? (Cin >> 5) - kb0 : geo.kb_per) : (Cin >> 5);
    const int nsteps = nkb * 16;        // (block, position) steps
    const int ipc = CV ? geo.gy * geo.gx : 1;   // images per canvas
    // canvas pixel (vy, vx) of canvas `img` -> image pixel index; false: a gap, past the last image, off the canvas
    auto canvas_pixel = [&](const int vy, const int vx, int& o) -> bool {
        // (24-bit multiplications, full rate: coordinates <= 4095, multipliers < 2^19, B * H * W < 2^24 -- the launcher checks)
        const int sy = (int)(__umul24((unsigned int)vy, geo.my) >> 20), sx = (int)(__umul24((unsigned int)vx, geo.mx) >> 20);
        const int iy = vy - (int)__umul24(sy, geo.py), ix = vx - (int)__umul24(sx, geo.px);
        const int im = img * ipc + (int)__umul24(sy, geo.gx) + sx;
        o = (int)__umul24(__umul24(im, H) + iy, W) + ix;
        return vy >= 0 && vx >= 0 && iy < H && ix < W && sy < geo.gy && sx < geo.gx && im < p.B;
    };
#define WINO_DMA_W(st, step)                                                                                      \
    {                                                                                                              \
        const int kb_ = kb0 + ((step) >> 4), pos_ = (step) & 15;                                                   \
        const int soff_ = (pos_ * Cin + kb_ * 32) * 2;                                                             \
        _Pragma("unroll") for (int i = 0; i < WPW; ++i)                                                             \
            dma16(rsrc_w, wsm + (st) * WS_BYTES + (wave + NW * i) * 1024, w_goff[i], soff_);                       \
    }
    // pipeline: weight stages three deep (the DMA of step t+2 is issued at step t); the patch of the next channel block is
    // requested as soon as the last reads of the current one (the row combination of xi = 3) are done.
    // Waves w and w+4 share a SIMD: waves 0-3 form V(t) at the START of step t, waves 4-7 form V(t+1) at the END of step t, so
    // that one wave's transform (VALU) runs under its partner's MFMAs.
#define WINO_DMA_PATCH(kb_)                                                                                        \
    if constexpr (CV) {                                                                                              \
        /* (canvas form: the pixel -> address map of a lane's PPW pieces does not depend on the channel block) */       \
        _Pragma("unroll") for (int i = 0; i < PPW; ++i)                                                               \
            dma16(rsrc_in, smem + (wave + NW * i) * 1024, poff[i], (kb0 + (kb_)) * 128);                             \
    } else                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                                \
        const int slot_ = (wave + NW * i) * 8 + (lane >> 3);                                                         \
        const int pix_ = slot_ ^ ((slot_ >> 1) & 1);                                                                 \
        const int pr_ = pix_ / PW, pc_ = pix_ - pr_ * PW;                                                            \
        const int iy_ = oy0 - pad_ + pr_, ix_ = ox0 - pad_ + pc_;                                                    \
        int pxi_ = (img * H + iy_) * W + ix_;                                                                        \
        bool ok_ = pix_ < NPX && (unsigned)iy_ < (unsigned)H && (unsigned)ix_ < (unsigned)W;                         \
        if constexpr (CV) ok_ = canvas_pixel(iy_, ix_, pxi_) && pix_ < NPX;                                          \
        const int u_ = pc_ >> 1;                                                                                     \
        const int lc_ = ((((lane & 7) >> 1) - 2 * (u_ >> 2)) & 3) * 2 + ((lane & 1) ^ ((u_ >> 1) & 1));               \
        if constexpr (TWO) {                                                                                         \
            const bool s2_ = (kb_) >= kb1;   /* (uniform) */                                                           \
            const int off2_ = ok_ ? pxi_ * (s2_ ? row_bytes2 : row_bytes) + lc_ * 16 +                                \
                                        (s2_ ? p.in2_coff * 4 + ((kb_) - kb1) * 128 : p.in_coff * 4 + (kb_) * 128)   \
                                  : (int)0x80000000;                                                                 \
            if (s2_) dma16(rsrc_in2, smem + (wave + NW * i) * 1024, off2_, 0);                                       \
            else dma16(rsrc_in, smem + (wave + NW * i) * 1024, off2_, 0);                                            \
            continue;                                                                                                \
        }                                                                                                            \
        const int off_ = ok_ ? pxi_ * row_bytes + p.in_coff * 4 + lc_ * 16 + (kb0 + (kb_)) * 128                     \
                             : (int)0x80000000;                                                                      \
        dma16(rsrc_in, smem + (wave + NW * i) * 1024, off_, 0);                                                      \
    }
    int poff[CV ? PPW : 1];
    if constexpr (CV) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int slot_ = (wave + NW * i) * 8 + (lane >> 3);
            const int pix_ = slot_ ^ ((slot_ >> 1) & 1);
            const int pr_ = pix_ / PW, pc_ = pix_ - pr_ * PW;
            int pxi_;
            const bool ok_ = canvas_pixel(oy0 - 1 + pr_, ox0 - 1 + pc_, pxi_) && pix_ < NPX;
            const int u_ = pc_ >> 1;
            const int lc_ = ((((lane & 7) >> 1) - 2 * (u_ >> 2)) & 3) * 2 + ((lane & 1) ^ ((u_ >> 1) & 1));
            poff[i] = ok_ ? pxi_ * row_bytes + p.in_coff * 4 + lc_ * 16 : (int)0x80000000;
        }
    }
    constexpr bool LATE = VAR & 1, XIU = (VAR >> 1) & 1;   // measurement variants (default 0)
    const bool late = LATE && wave >= 4;
    f32x4 rc[4][2];   // s1 * d[a1][b] + s2 * d[a2][b] for the four patch columns b of the current xi (8 channels: two vectors)
    bf16x8 pf[3];     // V(xi, nu) of the current step, split
    // rows of B^T for xi: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    auto combine_rows = [&](const int xi) {
        const int a1 = xi == 0 ? 0 : 1, a2 = xi == 3 ? 3 : 2;
        const float s1 = xi == 2 ? -1.f : 1.f, s2 = (xi == 1 || xi == 2) ? 1.f : -1.f;
        const f32x4 s1v = {s1, s1, s1, s1}, s2v = {s2, s2, s2, s2};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            // pixel (row 2 wave + a, column 2 tx + b): p = row * PW + column, (p >> 1) & 1 = (a + u) & 1 with u = column >> 1
            const int u = tx + (b >> 1);
            const int p1 = (2 * wave + a1) * PW + 2 * tx + b, p2 = (2 * wave + a2) * PW + 2 * tx + b;
            const int o1 = (p1 ^ ((a1 + u) & 1)) << 7, o2 = (p2 ^ ((a2 + u) & 1)) << 7;
            const int pair = ((q8 + 2 * (u >> 2)) & 3) << 5, hb = ((u >> 1) & 1) << 4;   // chunk 2 pair + (h ^ bit)
            const f32x4 d1l = *reinterpret_cast<const f32x4*>(smem + o1 + pair + hb);
            const f32x4 d1h = *reinterpret_cast<const f32x4*>(smem + o1 + pair + (hb ^ 16));
            const f32x4 d2l = *reinterpret_cast<const f32x4*>(smem + o2 + pair + hb);
            const f32x4 d2h = *reinterpret_cast<const f32x4*>(smem + o2 + pair + (hb ^ 16));
            // (+-1 coefficients: the products are exact, one rounding per addition)
            rc[b][0] = __builtin_elementwise_fma(d2l, s2v, d1l * s1v);
            rc[b][1] = __builtin_elementwise_fma(d2h, s2v, d1h * s1v);
        }
    };
    // row nu of B^T over the four row-combined columns: 0: c0 - c2, 1: c1 + c2, 2: c2 - c1, 3: c1 - c3 (nu is a constant
    // wherever this is expanded)
    auto form_v = [&](const int nu) {
        f32x4 v0, v1;
        if (nu == 0) { v0 = rc[0][0] - rc[2][0]; v1 = rc[0][1] - rc[2][1]; }
        else if (nu == 1) { v0 = rc[1][0] + rc[2][0]; v1 = rc[1][1] + rc[2][1]; }
        else if (nu == 2) { v0 = rc[2][0] - rc[1][0]; v1 = rc[2][1] - rc[1][1]; }
        else { v0 = rc[1][0] - rc[3][0]; v1 = rc[1][1] - rc[3][1]; }
        const float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        split8(vv, pf[0], pf[1], pf[2]);
    };
    WINO_DMA_PATCH(0)
    WINO_DMA_W(0, 0)
    if (nsteps > 1) WINO_DMA_W(1, 1)
    int st = 0;   // weight stage of the current step
    // one xi group (four positions = four steps).  XIU: expanded for each xi (coefficients and row choices become immediates:
    // no multiplications by 0 / +-1 in the fold and the row combination), else one body with run-time xi
    auto xi_group = [&](const int kb, const int xi) __attribute__((always_inline)) {
        {
            // A^T = [[1,1,1,0],[0,1,-1,-1]]: coefficient of M(xi, nu) in output (i, j) = At[i][xi] * At[j][nu]
            const float ci0 = xi < 3 ? 1.f : 0.f, ci1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const int step = kb * 16 + xi * 4 + nu;
                // this step's weight pieces have landed; after the barrier everybody's have, and every wave is past its reads of
                // the stage that the DMA issued below overwrites.  Loads complete in order; issued after this step's weights:
                // the next step's WPW pieces and, on the two steps that follow the request of the next block's patch, its PPW
                if (step + 1 >= nsteps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (xi == 3 && (nu == 1 || nu == 2) && kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
                if constexpr (!(DBG & 8)) wg_barrier<(DBG & 32) != 0>();
                const int st2 = st >= 1 ? st - 1 : 2;   // (st + 2) % 3
                if (step + 2 < nsteps && !(DBG & 16)) WINO_DMA_W(st2, step + 2)
                if ((!late || step == 0) && (!(DBG & 4) || step == 0)) {   // (step 0: every wave)
                    if (nu == 0) combine_rows(xi);
                    form_v(nu);
                }
                if (nu == 0 && xi == 3 && kb + 1 < nkb && !(DBG & 16)) {   // (uniform) the patch is free: request the next channel block
                    if constexpr (!(DBG & 8)) wg_barrier<(DBG & 32) != 0>();
                    WINO_DMA_PATCH(kb + 1)
                }
                const float cj1 = nu == 1 ? 1.f : -1.f;   // (nu 0 adds nothing to output column 1, nu 3 nothing to column 0)
                const float c01 = ci0 * cj1, c11 = ci1 * cj1;
                const unsigned char* wc = wsm + st * WS_BYTES + w_addr_l;
                // weights = A operand (rows = output channels), tiles = B operand (columns); small terms first.  The fragments
                // of block j+1 are requested before block j's MFMAs, block j-1's result is folded into Y between them.
                bf16x8 wf[2][3];
                f32x4 mm[2];   // product of block j in mm[j & 1]
#pragma unroll
                for (int k = 0; k < 3; ++k) wf[0][k] = *reinterpret_cast<const bf16x8*>(wc + k * W_PLANE);
#pragma unroll
                for (int j = 0; j <= TJ; ++j) {
                    if (j < TJ) {
                        if (j + 1 < TJ) {
#pragma unroll
                            for (int k = 0; k < 3; ++k)
                                wf[(j + 1) & 1][k] = *reinterpret_cast<const bf16x8*>(wc + (j + 1) * 1024 + k * W_PLANE);
                        }
                        const bf16x8 w0 = wf[j & 1][0], w1 = wf[j & 1][1], w2 = wf[j & 1][2];
                        const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
                        {
                            f32x4 m = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, pf[0], zero, 0, 0, 0);
                            m = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[2], m, 0, 0, 0);
                            m = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, pf[1], m, 0, 0, 0);
                            m = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, pf[0], m, 0, 0, 0);
                            m = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[1], m, 0, 0, 0);
                            mm[j & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[0], m, 0, 0, 0);
                        }
                    }
                    if (j > 0) {
                        const f32x4 mp = mm[(j - 1) & 1];
                        if constexpr (DBG & 2) {
                            asm volatile("" ::"v"(mp));
                        } else if constexpr (XIU) {   // (xi is a constant here: additions / subtractions, nothing for a zero coefficient)
                            if (nu < 3) {
                                if (xi < 3) Y[0][j - 1] += mp;
                                if (xi == 1) Y[2][j - 1] += mp;
                                if (xi >= 2) Y[2][j - 1] -= mp;
                            }
                            if (nu > 0) {
                                const bool neg = nu != 1;   // cj1 = -1
                                if (xi < 3) Y[1][j - 1] = neg ? Y[1][j - 1] - mp : Y[1][j - 1] + mp;
                                if (xi == 1) Y[3][j - 1] = neg ? Y[3][j - 1] - mp : Y[3][j - 1] + mp;
                                if (xi >= 2) Y[3][j - 1] = neg ? Y[3][j - 1] + mp : Y[3][j - 1] - mp;
                            }
                        } else {
                            if (nu < 3) {
                                Y[0][j - 1] = __builtin_elementwise_fma(mp, f32x4{ci0, ci0, ci0, ci0}, Y[0][j - 1]);
                                Y[2][j - 1] = __builtin_elementwise_fma(mp, f32x4{ci1, ci1, ci1, ci1}, Y[2][j - 1]);
                            }
                            if (nu > 0) {
                                Y[1][j - 1] = __builtin_elementwise_fma(mp, f32x4{c01, c01, c01, c01}, Y[1][j - 1]);
                                Y[3][j - 1] = __builtin_elementwise_fma(mp, f32x4{c11, c11, c11, c11}, Y[3][j - 1]);
                            }
                        }
                        // (pins the fold here: without a use inside this scheduling region the optimiser sinks it to the loop's end)
                        // ... in fixed registers (Y[i][j] = v[32 i + 4 j ..]): in-place updates, no copies around the loop
                        switch (j - 1) {
                            case 0: asm volatile("" : "+{v[0:3]}"(Y[0][0]), "+{v[32:35]}"(Y[1][0]), "+{v[64:67]}"(Y[2][0]), "+{v[96:99]}"(Y[3][0])); break;
                            case 1: asm volatile("" : "+{v[4:7]}"(Y[0][1]), "+{v[36:39]}"(Y[1][1]), "+{v[68:71]}"(Y[2][1]), "+{v[100:103]}"(Y[3][1])); break;
                            case 2: asm volatile("" : "+{v[8:11]}"(Y[0][2]), "+{v[40:43]}"(Y[1][2]), "+{v[72:75]}"(Y[2][2]), "+{v[104:107]}"(Y[3][2])); break;
                            case 3: asm volatile("" : "+{v[12:15]}"(Y[0][3]), "+{v[44:47]}"(Y[1][3]), "+{v[76:79]}"(Y[2][3]), "+{v[108:111]}"(Y[3][3])); break;
                            case 4: asm volatile("" : "+{v[16:19]}"(Y[0][4]), "+{v[48:51]}"(Y[1][4]), "+{v[80:83]}"(Y[2][4]), "+{v[112:115]}"(Y[3][4])); break;
                            case 5: asm volatile("" : "+{v[20:23]}"(Y[0][5]), "+{v[52:55]}"(Y[1][5]), "+{v[84:87]}"(Y[2][5]), "+{v[116:119]}"(Y[3][5])); break;
                            case 6: asm volatile("" : "+{v[24:27]}"(Y[0][6]), "+{v[56:59]}"(Y[1][6]), "+{v[88:91]}"(Y[2][6]), "+{v[120:123]}"(Y[3][6])); break;
                            case 7: asm volatile("" : "+{v[28:31]}"(Y[0][7]), "+{v[60:63]}"(Y[1][7]), "+{v[92:95]}"(Y[2][7]), "+{v[124:127]}"(Y[3][7])); break;
                        }
                    }
                    if (j < TJ) {
                        // order: the fragment reads first, then the MFMAs with the fold's VALU work in their shadows
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
                        for (int g = 0; g < 6; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (late && step + 1 < nsteps && !(DBG & 4)) {   // V of the next step (its patch has landed: see the wait of step 15)
                    if (nu == 3) combine_rows((xi + 1) & 3);
                    form_v((nu + 1) & 3);
                }
                st = st == 2 ? 0 : st + 1;
            }
        }
    };
    for (int kb = 0; kb < nkb; ++kb) {
        if constexpr (XIU) {
            xi_group(kb, 0);
            xi_group(kb, 1);
            xi_group(kb, 2);
            xi_group(kb, 3);
        } else {
#pragma unroll 1
            for (int xi = 0; xi < 4; ++xi) xi_group(kb, xi);
        }
    }
#undef WINO_DMA_PATCH
#undef WINO_DMA_W

    // ---- epilogue.  D layout of a 16x16 block: column (lane & 15) = tile, rows 4 (lane >> 4) + e = 4 consecutive channels:
    // stored straight from the accumulators a wave-instruction would write 16 pixels x 64 B, 2 KiB apart.  Through LDS
    // (free now) it writes 2 pixels x 512 B: tile row 0 of every tile, then tile row 1 (256 pixels x 128 channels each).
    //   LDS image: pixel pl = 32 wave + 2 tx + (i & 1), 16-byte chunk c of its 128 channels at chunk c ^ (tx & 15)
    // (CV, K range of a split layer: raw partial sums into the workspace slice [ks][pixel][Npad] through the same code -- a
    // descriptor without bias / residual / gates / masks; the second pass applies the layer's epilogue)
    spaa_tapconv_t pq = p;
    if constexpr (CV) {
        if (geo.ksplit > 1) {
            pq.out = p.splitk_ws + (geo.fix ? SPAA_SPLITK_HDR_FLOATS : 0) + (size_t)ks * ((size_t)p.B * H * W) * npad;
            pq.out_cstride = npad;
            pq.out_coff = 0;
            pq.bias = nullptr;
            pq.add = nullptr;
            pq.gate = nullptr;
            pq.gate2 = nullptr;
            pq.gate_bits = nullptr;
            pq.gate2_bits = nullptr;
            pq.mask_out = nullptr;
            pq.aux_out = nullptr;
            pq.act = SPAA_ACT_NONE;
            pq.io_dtype = 0;
        }
    }
    const spaa_tapconv_t& e = CV ? pq : p;
    const bool vec = !((e.Cout | e.out_cstride | e.out_coff) & 3) &&
                     (e.add == nullptr || !((e.add_cstride | e.add_coff) & 3)) &&
                     (e.gate == nullptr || !((e.gate_cstride | e.gate_coff) & 3)) &&
                     (e.gate2 == nullptr || !((e.gate2_cstride | e.gate2_coff) & 3));
    if constexpr (DBG & 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) asm volatile("" ::"v"(Y[i][j]));
        return;
    }
    constexpr int ROWB = BN * 4;          // bytes of a pixel's BN channels in the LDS image
    constexpr int LPP = BN / 4;           // lanes (16-byte chunks) per pixel: 32 or 16
    constexpr int PPI = 64 / LPP;         // pixels per wave store instruction
    // K ranges, round 6: the second pass inside this kernel.  Every workgroup of a (region, N tile) bumps that tile's arrival counter
    // after its partial sums are visible device-wide; the LAST to arrive -- whichever it is -- adds the K ranges in the fixed order
    // 0, 1, 2, ... (its own included, read back: the summation order, and with it every bit of the result, is the two-pass form's)
    // and applies the layer's epilogue through the same store4_t.  Nobody waits for anybody: no spinning, no dependence on the order
    // in which workgroups are scheduled.  The counters (int32, head of the workspace) are zero before and after every launch.
    auto splitk_fixup = [&]() {
        if constexpr (CV) {
            if (geo.ksplit <= 1 || !geo.fix) return;
            // (this thread's partial sums were agent-scope stores: once they are acknowledged they are where every XCD reads them)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* const flag = reinterpret_cast<int*>(smem);      // (the epilogue's LDS is free now)
            if (tid == 0) {
                int* const cnt = reinterpret_cast<int*>(p.splitk_ws) + region * n_tiles + n_blk / BN;
                const int old = atomicAdd(cnt, 1);               // (agent scope)
                const int last = old == geo.ksplit - 1;
                if (last) atomicExch(cnt, 0);      // (nobody else touches this counter any more in this launch)
                *flag = last;
            }
            __syncthreads();
            if (*flag == 0) return;
            const int M = p.B * H * W;
            const auto rws = rsrc_or_empty(p.splitk_ws + SPAA_SPLITK_HDR_FLOATS, (int64_t)geo.ksplit * M * npad * 4);
            const bool pvec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                              (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                              (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                              (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
            for (int i = tid; i < 2 * TY * 32 * LPP; i += 64 * NW) {
                const int qd = i & (LPP - 1), pxl = i / LPP;
                const int n0 = n_blk + 4 * qd;
                int o;
                if (!canvas_pixel(oy0 + (pxl >> 5), ox0 + (pxl & 31), o) || n0 >= p.Cout) continue;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                for (int s_ = 0; s_ < geo.ksplit; ++s_) {   // agent-scope loads: served from where the other XCDs' stores went
                    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rws, ((s_ * M + o) * npad + n0) * 4, 0, SPAA_AUX_SC1);
                    sum += f32x4{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
                }
                float v[4] = {sum[0], sum[1], sum[2], sum[3]};
                store4_t<float>(p, (size_t)o, n0, v, pvec);
            }
        }
    };
    if (fast_epi_ok(e, vec)) {
        // The operand combinations of the attack loops, branch-free (epilogue.hpp: fast_epi_*): a wave's 32 x 2 pixels go
        // through a PRIVATE LDS region (row = 16 (pixel column & 1) + tile column, padded by 16 bytes: conflict-free writes
        // from the MFMA layout, a lane keeps ONE channel quad), so that one barrier after the main loop is all the
        // synchronisation there is, and the residual / gate operands of eight pixels per lane are in flight together.
        constexpr int ROWP = BN * 4 + 16;
        const int n = n_blk + 4 * (lane & (LPP - 1));
        const bool n_ok = n < e.Cout;
        fast_epi_t fe = make_fast_epi(e, n_ok ? n : 0);
        if constexpr (CV) fe.out_sc1 = geo.ksplit > 1 && geo.fix;
        wg_barrier<false>();
        unsigned char* const eb = smem + wave * (32 * ROWP);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    *reinterpret_cast<f32x4*>(eb + (16 * c + tx) * ROWP + ((4 * j + q8) << 4)) = Y[2 * half + c][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int oy = oy0 + 2 * wave + half;
            const int orow = (img * e.Hout + oy) * e.Wout + ox0;
            constexpr int EB = 8;
#pragma unroll 1
            for (int it0 = 0; it0 < 32 / PPI; it0 += EB) {
                fast_pre_t<float> pre[EB];
                int oo[EB];
                bool ok[EB];
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    const int r = (it0 + u) * PPI + lane / LPP, px = 2 * (r & 15) + (r >> 4);
                    oo[u] = orow + px;
                    ok[u] = n_ok && oy < e.Hout && ox0 + px < e.Wout;
                    if constexpr (CV) ok[u] = canvas_pixel(oy, ox0 + px, oo[u]) && n_ok;
                    pre[u] = fast_epi_load<float>(fe, e, oo[u], n, ok[u]);
                }
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    const int r = (it0 + u) * PPI + lane / LPP;
                    const f32x4 y = *reinterpret_cast<const f32x4*>(eb + r * ROWP + ((lane & (LPP - 1)) << 4));
                    fast_epi_store<float, f32x4, false, CV>(fe, e, oo[u], n, ok[u], y, pre[u]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        splitk_fixup();
        return;
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        wg_barrier<(DBG & 64) != 0>();   // the main loop's (resp. the previous half's) LDS reads are done (raw: the stores stay in flight)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int pl = 32 * wave + 2 * tx + c;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                *reinterpret_cast<f32x4*>(smem + pl * ROWB + (((4 * j + q8) ^ tx) << 4)) = Y[2 * half + c][j];
        }
        wg_barrier<(DBG & 64) != 0>();
        // the residual / gate operands of EB pixels are requested before any of them is finished: EB loads in flight per lane
        constexpr int EB = 8;
#pragma unroll 1
        for (int it0 = 0; it0 < 32 / PPI; it0 += EB) {   // 256 pixels / 8 waves / PPI per wave
            epi_pre_t pre[EB];
            size_t oo[EB];
            bool ok[EB];
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int pl = PPI * ((32 / PPI) * wave + it0 + u) + lane / LPP;
                const int lc = (lane & (LPP - 1)) ^ ((pl & 31) >> 1);
                const int oy = oy0 + 2 * (pl >> 5) + half, ox = ox0 + (pl & 31);
                ok[u] = oy < e.Hout && ox < e.Wout;
                oo[u] = ((size_t)img * e.Hout + (ok[u] ? oy : 0)) * e.Wout + (ok[u] ? ox : 0);
                if constexpr (CV) {
                    int o_;
                    ok[u] = canvas_pixel(oy, ox, o_);
                    oo[u] = ok[u] ? (size_t)o_ : 0;
                }
                if (vec && ok[u]) pre[u] = epi_load<float>(e, oo[u], n_blk + 4 * lc);
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int pl = PPI * ((32 / PPI) * wave + it0 + u) + lane / LPP;
                const int pc = lane & (LPP - 1);                // physical chunk
                const int lc = pc ^ ((pl & 31) >> 1);           // logical chunk: channels 4 lc .. 4 lc + 3 of this n tile
                const f32x4 y = *reinterpret_cast<const f32x4*>(smem + pl * ROWB + (pc << 4));
                if (ok[u]) {
                    float v[4] = {y[0], y[1], y[2], y[3]};
                    if (vec) store4_pre<float>(e, oo[u], n_blk + 4 * lc, v, pre[u]);
                    else store4_t<float>(e, oo[u], n_blk + 4 * lc, v, vec);
                }
            }
        }
    }
    splitk_fixup();
}

// second pass of a K-split layer: out = epilogue( sum over the splits, in fixed order ), 4 channels per thread
__global__ __launch_bounds__(256) void wino_splitk_reduce_kernel(const spaa_tapconv_t p, const int M, const int npad) {
    const int nq = (p.Cout + 3) >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * nq) return;
    const int m = (int)(idx / nq), n0 = (int)(idx - (int64_t)m * nq) * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.ksplit; ++s) sum += *reinterpret_cast<const f32x4*>(p.splitk_ws + ((size_t)s * M + m) * npad + n0);
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    float v[4] = {sum[0], sum[1], sum[2], sum[3]};
    store4_t<float>(p, (size_t)m, n0, v, vec);
}

// ---- launch plan: N tile, canvas layout, K split (one decision for the launcher and for spaa_tapconv_wino_plan)
struct wino_plan_t {
    int bn, ksplit, kb_per, canvas, gy, gx, py, px, ncanvas, wg_y, wg_x, n_tiles;
    int64_t nwg;
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// `force_bn`: 0 = choose, 64 / 128.  `force_ks`: 0 = choose (only when `allow_split`), else that many K ranges (clamped so that
// every range holds at least one 32-channel block).  The choice minimises a cost model fitted to the round-3 per-layer tables
// (profiles/r03_v6_tapconv_layers.json): a workgroup needs 12.8 us (64-wide) / 22.4 us (128-wide) per 32-channel block and 12 / 20
// us for its prologue + epilogue (64-wide canvas / K-range form: + 6), the launch takes about ceil(workgroups / CUs) such rounds, a K
// split adds its second pass (5 us + the partial sums at 12 TB/s: L2 / MALL).  Predicted / measured: conv4 438 / 420-457, conv3_s 130 /
// 128, ResNet layer2 63 / 65, layer1 75 / 75, layer3 (canvas, 4 ranges) 74 / 77, layer4 (8 ranges) 74 / 76, VGG-16 14 x 14 x 512
// (2 ranges) 210 / 205 (tools/lab/wino_small.py).
// `nw` = waves per workgroup: 8 (16 x 32-pixel regions, one workgroup per CU) or 4 (tile 73: 8 x 32-pixel regions, two per CU: `ncu`
// counts slots, the prologue + epilogue of one workgroup mostly hides under the other's main loop).
inline wino_plan_t wino_make_plan(const spaa_tapconv_t& d, const int ncu_, const int force_bn, const int force_ks, const bool allow_split,
                                  const int nw = 8) {
    wino_plan_t pl = {};
    const int H = d.Hout, W = d.Wout, B = d.B;
    const int TY = nw;                       // (shadows the 8-wave constant: tile rows of a region)
    const int ncu = nw == 4 ? 2 * ncu_ : ncu_;
    // plain regions: per image
    const int pwy = cdiv(H, 2 * TY), pwx = cdiv(W, 2 * TX);
    const int64_t plain = (int64_t)B * pwy * pwx;
    // best canvas: gy x gx images with periods (H + 1, W + 1); sides <= 4095, periods <= 255 (the kernel's division by multiplication)
    int64_t best = plain;
    int cgy = 1, cgx = 1, cnc = B, cwy = pwy, cwx = pwx;
    const int py = H + 1, px = W + 1;
    const bool small = (int64_t)B * H * W < ((int64_t)1 << 24) && H <= 4095 && W <= 4095;   // (the canvas / K-range kernel's 24-bit index arithmetic)
    if (py <= 255 && px <= 255 && small && !((d.reserved0 >> 30) & 1)) {
        for (int gx = 1; gx <= B && gx * px - 1 <= 4095; ++gx) {
            const int wx = cdiv(gx * px - 1, 2 * TX);
            for (int gy = 1; gy * gx <= B + gx - 1 && gy * py - 1 <= 4095; ++gy) {   // (gy up to ceil(B / gx))
                const int wy = cdiv(gy * py - 1, 2 * TY);
                const int nc = cdiv(B, gy * gx);
                const int64_t n = (int64_t)nc * wy * wx;
                if (n < best) best = n, cgy = gy, cgx = gx, cnc = nc, cwy = wy, cwx = wx;
            }
        }
    }
    const int nkb = d.Cin / 32;
    const int64_t M = (int64_t)B * H * W;
    const int npad = (d.Cout + 127) & ~127;
    double best_cost = 1e30;
    for (int cv = 0; cv < 2; ++cv) {
        if (cv && best >= plain) break;
        const int64_t regions = cv ? best : plain;
        for (int bn = 64; bn <= 128; bn += 64) {
            if (force_bn ? bn != force_bn : (bn == 128 && d.Cout <= 64)) continue;
            const int nt = cdiv(d.Cout, bn);
            for (int ks = 1; ks <= nkb; ++ks) {
                if (force_ks > 0 ? ks != (force_ks < nkb ? force_ks : nkb) : (ks > 1 && (!allow_split || nkb < 2 * ks))) continue;
                if (ks > 1 && ((d.Cout & 3) || !small)) continue;
                const int kb_per = cdiv(nkb, ks), ksr = cdiv(nkb, kb_per);
                if (ksr != ks && force_ks <= 0) continue;   // (the same plan as a smaller ks)
                const int64_t nwg = regions * nt * ksr;
                // (a partly filled last round costs less than a full one -- idle compute units leave the others a higher clock:
                // VGG-16's 28 x 28 x 512 layers, 432 against 512 workgroups, 720 against 780 us)
                const double rounds = 0.5 * (double)((nwg + ncu - 1) / ncu) + 0.5 * (double)nwg / ncu;
                double cost = rounds * (kb_per * (bn == 128 ? 22.4 : 12.8) + (bn == 128 ? 20.0 : 12.0) + ((cv || ksr > 1) && bn == 64 ? 6.0 : 0.0));
                if (nw == 4) cost = rounds * (kb_per * 12.8 + 6.0);   // (two half-size workgroups per CU share its rate: per slot the same time per block)
                if (ksr > 1) cost += 5.0 + (double)ksr * (double)M * npad * 4.0 / 12e6;   // (second pass: the partial sums come from L2 / MALL)
                if (cv && !((d.reserved0 >> 29) & 1)) cost *= nw == 4 ? 1.15 : 1.05;   // (the canvas form must win by 5 % -- 15 % with four-wave workgroups: 56 x 56, 798 against 896 workgroups, 73 against 69 us -- else the image-aligned regions)
                if (!cv && best < plain && ((d.reserved0 >> 29) & 1)) cost *= 1e6;   // (tests: canvas wherever it has fewer regions)
                if (ksr > 1) cost *= 1.02;  // (ties: no split)
                if (cost < best_cost) {
                    best_cost = cost;
                    pl.bn = bn, pl.ksplit = ksr, pl.kb_per = kb_per, pl.canvas = cv, pl.n_tiles = nt, pl.nwg = nwg;
                }
            }
        }
    }
    if (!pl.canvas && pl.ksplit == 1 && force_bn == 0 && nw == 8) {
        // image-aligned regions without a split: the N tile by the rule the round-2 / round-3 tune tables were measured with (64
        // wide for at most 64 output channels and where the 128-wide grid would leave compute units without a workgroup)
        pl.bn = (d.Cout <= 64 || plain * cdiv(d.Cout, 128) < ncu) ? 64 : 128;
        pl.n_tiles = cdiv(d.Cout, pl.bn);
        pl.nwg = plain * pl.n_tiles;
    }
    if (pl.canvas) pl.gy = cgy, pl.gx = cgx, pl.ncanvas = cnc, pl.wg_y = cwy, pl.wg_x = cwx;
    else pl.gy = pl.gx = 1, pl.ncanvas = B, pl.wg_y = pwy, pl.wg_x = pwx;
    pl.py = py, pl.px = px;
    return pl;
}

inline int wino_ncu() {
    static int ncu[SPAA_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SPAA_MAX_DEVICES) return 256;
    if (ncu[dev] == 0 && hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu[dev] = 256;
    return ncu[dev] > 0 ? ncu[dev] : 256;
}

inline bool wino_shape_ok(const spaa_tapconv_t& d) {
    if (d.w_split == nullptr || (d.Cin % 32) != 0 || d.Cin < 32 || d.nclass != 1 || d.cls[0].ntaps != 16 || d.cls[0].K != 16 * d.Cin || d.cls[0].Kpad < d.cls[0].K || (d.cls[0].Kpad & 7) ||
        d.s_in != 1 || d.s_out != 1 || d.Hout != d.Hin + 2 * wino_pad(d.reserved0) - 2 || d.Wout != d.Win + 2 * wino_pad(d.reserved0) - 2 || d.Hm != d.Hout || d.Wm != d.Wout || d.nfold > 1 ||
        d.ksplit < 0 || d.io_dtype != 0 || d.B < 1 || d.Hout < 1 || d.Wout < 1 || d.Cout < 1)
        return false;
    if ((int64_t)((d.Cout + 127) & ~127) * d.cls[0].Kpad * 6 >= (int64_t)1 << 31) return false;
    if ((int64_t)d.B * d.Hin * d.Win * d.in_cstride * 4 >= (int64_t)1 << 31) return false;   // (32-bit DMA offsets)
    return true;
}

}  // namespace

// The launcher's plan for the Winograd form of `desc` (tile 70: N tile chosen, 71: 64-wide N tile; desc->ksplit > 1: that many K
// ranges, 1: none, 0: chosen): plan[0..7] = {N tile, K ranges, canvas (0 / 1), images per canvas (rows), (columns), workgroups,
// channel blocks per K range, canvases}.  A caller sizes `splitk_ws` (K ranges x B x H x W x Npad floats) from plan[1] and passes
// plan[1] back as `ksplit`.
extern "C" int spaa_tapconv_wino_plan(const spaa_tapconv_t* desc, int32_t* plan) {
    if (desc == nullptr || plan == nullptr) return hipErrorInvalidValue;
    spaa_tapconv_t d = *desc;
    if (d.w_split == nullptr) d.w_split = reinterpret_cast<const uint16_t*>(desc);   // (the plan does not depend on the pointers)
    if (!wino_shape_ok(d) || (d.tile != 70 && d.tile != 71 && d.tile != 73)) return hipErrorInvalidValue;
    const bool nosplit = d.in2 != nullptr || wino_pad(d.reserved0) != 1;   // (two sources / unpadded layers: image-aligned regions, no K ranges)
    if (nosplit) d.reserved0 |= 1 << 30;
    const wino_plan_t pl = wino_make_plan(d, wino_ncu(), d.tile == 70 ? 0 : 64, nosplit ? 1 : d.ksplit, !nosplit, d.tile == 73 ? 4 : 8);
    if (pl.bn == 0 || pl.ksplit < 1 || pl.n_tiles < 1) return hipErrorInvalidValue;   // (a forced K-range count with no admissible candidate)
    plan[0] = pl.bn, plan[1] = pl.ksplit, plan[2] = pl.canvas, plan[3] = pl.gy, plan[4] = pl.gx;
    plan[5] = (int32_t)(pl.nwg > 0x7fffffff ? 0x7fffffff : pl.nwg), plan[6] = pl.kb_per, plan[7] = pl.ncanvas;
    return 0;
}

// called by spaa_tapconv_f32 (tapconv.hip) for tiles 70 / 71 after the common shape checks.  The descriptor describes the layer in
// Winograd form: ONE class with 16 "taps" = the positions of U = G g G^T (tap entries unused), s_in = s_out = 1, same input and
// output size, Cin % 32 == 0.
int spaa_launch_tapconv_wino(const spaa_tapconv_t& d, hipStream_t stream) {
    if (!wino_shape_ok(d)) return hipErrorInvalidValue;
    // K ranges: the caller's (with its workspace), or chosen here when a workspace is there to take them
    const bool has_ws = d.splitk_ws != nullptr;
    if (d.ksplit > 1 && !has_ws) return hipErrorInvalidValue;
    spaa_tapconv_t dp = d;
    const bool nosplit = d.in2 != nullptr || wino_pad(d.reserved0) != 1;
    if (nosplit) dp.reserved0 |= 1 << 30;   // (two sources, unpadded layers: no canvas ...)
    if (nosplit && d.ksplit > 1) return hipErrorInvalidValue;
    const wino_plan_t pl = wino_make_plan(dp, wino_ncu(), d.tile == 70 ? 0 : 64, (has_ws && !nosplit) ? d.ksplit : 1, has_ws && !nosplit,
                                          d.tile == 73 ? 4 : 8);   // (... and no K ranges)
    if (pl.nwg > 0x7fffffff || pl.bn == 0 || pl.ksplit < 1 || pl.n_tiles < 1) return hipErrorInvalidValue;   // (no admissible plan: Cout % 4 with forced K ranges, > 2^24 pixels, sides > 4095)
    const int BN = pl.bn, n_tiles = pl.n_tiles, wg_y = pl.wg_y, wg_x = pl.wg_x;
    const int64_t nwg = pl.nwg;
    const bool two = d.in2 != nullptr;
    if (two && ((d.Cin2 % 32) != 0 || d.Cin2 <= 0 || d.Cin2 >= d.Cin || (d.in2_cstride & 3) || (d.in2_coff & 3) || d.in2_coff + d.Cin2 > d.in2_cstride ||
                d.in_coff + d.Cin - d.Cin2 > d.in_cstride || (int64_t)d.B * d.Hin * d.Win * d.in2_cstride * 4 >= (int64_t)1 << 31))
        return hipErrorInvalidValue;
    const bool cv = pl.canvas || pl.ksplit > 1;
    if (two && cv) return hipErrorInvalidValue;   // (the plan below never asks for it: two sources keep the image-aligned form)
    wino_geo_t geo = {};
    geo.ksplit = pl.ksplit, geo.kb_per = pl.kb_per;
    geo.gy = pl.gy, geo.gx = pl.gx, geo.py = pl.canvas ? pl.py : (1 << 14), geo.px = pl.canvas ? pl.px : (1 << 14);
    // v / p == (v * m) >> 20 with m = ceil(2^20 / p) for v * p < 2^20 (v <= 4095, p <= 255); image-aligned regions
    // (period 2^14 > every coordinate): m = 64 gives 0
    geo.my = ((1u << 20) + geo.py - 1) / geo.py, geo.mx = ((1u << 20) + geo.px - 1) / geo.px;
    geo.nsp = (int)(nwg / ((int64_t)n_tiles * pl.ksplit));
    // workgroup order by what an XCD's L2 should keep: the weight planes (16 positions x three bf16) or the activations
    geo.order = (int64_t)((d.Cout + 127) & ~127) * d.cls[0].Kpad * 6 > (int64_t)d.B * d.Hin * d.Win * d.Cin * 4;
    // `reserved1` bit 8: the workspace begins with SPAA_SPLITK_HDR_FLOATS zeroed counter slots -> the K ranges meet inside the kernel
    // (and the partial sums leave through the branch-free epilogue's agent-scope stores: 4-channel quads, 32-bit offsets)
    geo.fix = pl.ksplit > 1 && (d.reserved1 & 256) && (int64_t)geo.nsp * n_tiles <= SPAA_SPLITK_HDR_FLOATS && !(d.Cout & 3) &&
              (int64_t)pl.ksplit * d.B * d.Hout * d.Wout * ((d.Cout + 127) & ~127) * 4 < ((int64_t)1 << 31);
    if (cv && !pl.canvas && (d.Hout > 4095 || d.Wout > 4095)) return hipErrorInvalidValue;
    if (cv && (int64_t)d.B * d.Hout * d.Wout >= ((int64_t)1 << 24)) return hipErrorInvalidValue;   // (the plan never asks for it: wino_make_plan)
    spaa_tapconv_t dd = d;
    dd.ksplit = pl.ksplit;
    static bool attr_set[15][SPAA_MAX_DEVICES] = {};
    // kernel variant: bit 0 = late V (waves 4-7 transform one step ahead), bit 1 = xi groups expanded.  Default 3 / 2 (measured:
    // conv4 500 -> 462 us, conv5 461 -> 415 us against variant 0); `reserved0` bits 16-17 flip bits for A/B measurements
    const int var = (BN == 64 ? 2 : 3) ^ ((d.reserved0 >> 16) & 3);   // (64-wide tile: late V does not pay: 168 / 167 / 159 us for 0 / 3 / 2)
#define WINO_LAUNCH_T(N, V, C, SLOT)                                                                                      \
    {                                                                                                                     \
        const size_t smem = (size_t)PATCH_BYTES + 3 * (size_t)(((3 * N / 16 + 7) / 8) * 8 * 1024);                        \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<N, V, 0, C>), (int)smem, attr_set[SLOT]); \
        if (e != hipSuccess) return (int)e;                                                                               \
        hipLaunchKernelGGL((wino_x6_kernel<N, V, 0, C>), dim3((unsigned)nwg), dim3(512), smem, stream, dd, wg_y, wg_x, n_tiles, geo); \
    }
#define WINO_LAUNCH(N, V, SLOT) WINO_LAUNCH_T(N, V, false, SLOT)
#define WINO_LAUNCH_T2(N, V, SLOT)                                                                                        \
    {                                                                                                                     \
        const size_t smem = (size_t)PATCH_BYTES + 3 * (size_t)(((3 * N / 16 + 7) / 8) * 8 * 1024);                        \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<N, V, 0, false, true>), (int)smem, attr_set[SLOT]); \
        if (e != hipSuccess) return (int)e;                                                                               \
        hipLaunchKernelGGL((wino_x6_kernel<N, V, 0, false, true>), dim3((unsigned)nwg), dim3(512), smem, stream, dd, wg_y, wg_x, n_tiles, geo); \
    }
#ifdef SPAA_WINO_ABLATE
    const int dbg = (d.reserved0 >> 18) & 127;
#define WINO_LAUNCH_DBG(D)                                                                                                \
    if (dbg == D) {                                                                                                       \
        static bool as_[SPAA_MAX_DEVICES] = {};                                                                           \
        const size_t smem = (size_t)PATCH_BYTES + 3 * (size_t)(((3 * 128 / 16 + 7) / 8) * 8 * 1024);                      \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<128, 3, D>), (int)smem, as_);     \
        if (e != hipSuccess) return (int)e;                                                                               \
        hipLaunchKernelGGL((wino_x6_kernel<128, 3, D>), dim3((unsigned)nwg), dim3(512), smem, stream, dd, wg_y, wg_x, n_tiles, geo); \
        return (int)hipGetLastError();                                                                                    \
    }
    if (BN == 128 && !cv) {
        WINO_LAUNCH_DBG(1) WINO_LAUNCH_DBG(3) WINO_LAUNCH_DBG(5) WINO_LAUNCH_DBG(7) WINO_LAUNCH_DBG(15) WINO_LAUNCH_DBG(31) WINO_LAUNCH_DBG(2) WINO_LAUNCH_DBG(4) WINO_LAUNCH_DBG(8) WINO_LAUNCH_DBG(32) WINO_LAUNCH_DBG(33) WINO_LAUNCH_DBG(64) WINO_LAUNCH_DBG(96)
    }
#undef WINO_LAUNCH_DBG
#endif
    if (d.tile == 73) {   // four-wave workgroups, two per compute unit
        const size_t smem = (size_t)(((2 * 4 + 2) * PW + 7) / 8 + 3) / 4 * 4 * 1024 + 3 * (size_t)(12 * 1024);
        if (two) {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<64, 2, 0, false, true, 4>), (int)smem, attr_set[13]);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((wino_x6_kernel<64, 2, 0, false, true, 4>), dim3((unsigned)nwg), dim3(256), smem, stream, dd, wg_y, wg_x, n_tiles, geo);
        } else if (cv) {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<64, 2, 0, true, false, 4>), (int)smem, attr_set[14]);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((wino_x6_kernel<64, 2, 0, true, false, 4>), dim3((unsigned)nwg), dim3(256), smem, stream, dd, wg_y, wg_x, n_tiles, geo);
            if (pl.ksplit > 1 && !geo.fix) {
                const int npad = (d.Cout + 127) & ~127;
                const int64_t M = (int64_t)d.B * d.Hout * d.Wout, nthr = M * ((d.Cout + 3) >> 2);
                hipLaunchKernelGGL(wino_splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, dd, (int)M, npad);
            }
        } else {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<64, 2, 0, false, false, 4>), (int)smem, attr_set[12]);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((wino_x6_kernel<64, 2, 0, false, false, 4>), dim3((unsigned)nwg), dim3(256), smem, stream, dd, wg_y, wg_x, n_tiles, geo);
        }
    } else if (two) {   // two sources: the default variants only
        if (BN == 64) WINO_LAUNCH_T2(64, 2, 10) else WINO_LAUNCH_T2(128, 3, 11)
    } else if (cv) {   // canvas / K-split form: the default variants only
        if (BN == 64) WINO_LAUNCH_T(64, 2, true, 8) else WINO_LAUNCH_T(128, 3, true, 9)
        if (pl.ksplit > 1 && !geo.fix) {
            const int npad = (d.Cout + 127) & ~127;
            const int64_t M = (int64_t)d.B * d.Hout * d.Wout, nthr = M * ((d.Cout + 3) >> 2);
            hipLaunchKernelGGL(wino_splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, dd, (int)M, npad);
        }
    } else if (BN == 64) {
        if (var == 3) WINO_LAUNCH(64, 3, 4) else if (var == 2) WINO_LAUNCH(64, 2, 5) else if (var == 1) WINO_LAUNCH(64, 1, 6) else WINO_LAUNCH(64, 0, 7)
    } else if (var == 3) WINO_LAUNCH(128, 3, 3) else if (var == 2) WINO_LAUNCH(128, 2, 2) else if (var == 1) WINO_LAUNCH(128, 1, 1) else WINO_LAUNCH(128, 0, 0)
#undef WINO_LAUNCH
#undef WINO_LAUNCH_T
#undef WINO_LAUNCH_T2
    return (int)hipGetLastError();
}
