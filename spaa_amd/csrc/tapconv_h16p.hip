// tapconv_h16p.hip — fp16-STORAGE mode, 3x3 / stride-1 / pad-1 convolutions and their input gradients with the input PATCH
// staged once in LDS.
//
// The implicit-GEMM fp16 kernel (tapconv_h16.hip) gathers its activation operand once per TAP: a 32-pixel row block of 64 bytes
// per 32-deep sub-step and wave, i.e. 128 bytes of L2 -> LDS traffic per MFMA instruction at BN = 128 -- 32 B/clk/CU at the
// full matrix-core rate, more than twice what a CU's LDS-DMA path delivers.  Measured (profiles/r03_*): the six 64 x 64 x
// (128 <-> 256) layers sit at 460-590 TF whatever the pipeline depth (a three-stage counted-vmcnt pipeline: +-0) or the pixel
// block per wave.  A 3x3 convolution reads every input pixel nine times; here a workgroup stages the 18 x 34 pixel patch of
// its 16 x 32 output pixels ONCE per 32-channel block (LDS-DMA, out-of-image pixels = the out-of-range offset = the zero
// padding) and the nine taps read their B fragments from it at shifted positions: 39 KB of patch + 72 KB of weights per 9 x
// 32 MFMAs per wave = 12 B/clk/CU.
//   * workgroup = 8 waves = 16 x 32 output pixels x BN channels; wave w owns output rows 2w, 2w + 1 = four 16-pixel blocks:
//     a weight fragment read from LDS feeds four MFMAs;
//   * K order: 32-channel block, then tap; a step = three taps: 24 KB of weights (three LDS stages, DMA two steps ahead behind
//     a counted vmcnt and a raw s_barrier), 96 MFMAs per wave between barriers; the patch is double-buffered (the next block's
//     is requested at the block's first step);
//   * fp16 operands, fp32 accumulation (v_mfma_f32_16x16x32_f16), the shared fp32 epilogue (epilogue.hpp).
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
}
// chunk swizzle of a 64-byte row (four 16-byte chunks): rows 8 apart swap chunk pairs (tapconv_h16.hip swz64)
__device__ __forceinline__ int swz64(int r) { return ((r >> 3) & 1) << 1; }

constexpr int OW = 32;                         // output pixel columns of a workgroup
// S = 1: 16 x 32 output pixels (a wave: two rows = four 16-pixel blocks), patch 18 x 34 pixels = 39 KiB, double-buffered.
// S = 2 (round 5): the FORWARD form of a 3 x 3 / stride-2 convolution (ShadingNetSPAA.conv2 / conv2_s, models.py:224,230, the input
// gradient of transConv1, the classifiers' stride-2 3 x 3 layers): 8 x 32 output pixels (a wave: one row = two blocks), patch
// 17 x 65 pixels = 70 KiB in ONE buffer (reloaded per 32-channel block behind a barrier); the implicit-GEMM kernel gathered every
// input pixel once per tap that reads it (2.25 times) and ran these byte-bound layers at 140-320 TFLOP/s.
template <int S> struct h16p_geo {
    static constexpr int OH = S == 1 ? 16 : 8;
    static constexpr int NB = S == 1 ? 4 : 2;                        // 16-pixel blocks per wave
    static constexpr int PH = (OH - 1) * S + 3, PW = (OW - 1) * S + 3;
    static constexpr int NPX = PH * PW;
    static constexpr int P_PIECES = (NPX + 15) / 16;                 // 1-KiB pieces of 16 pixels x 32 channels (fp16)
    static constexpr int PPW = (P_PIECES + 7) / 8;                   // per wave
    static constexpr int PATCH_BYTES = PPW * 8 * 1024;
    static constexpr int NBUF = S == 1 ? 2 : 1;
};

// CANVAS / K-RANGE form (CV, round 5; S = 1, unfolded, one source): the mechanism of csrc/tapconv_wino.hip for small images with long K
// -- ResNet-18 layer3 / layer4 (14 x 14, 7 x 7 behind classifier.py:26-28), VGG-16's 14 x 14 x 512 block -- which leave most of a 16 x 32-pixel
// region empty and too few workgroups for 256 compute units, and ran on the implicit-GEMM fp16 kernel at 270-580 TFLOP/s.  The images of
// the batch lie on virtual canvases (gy x gx images with periods (H + 1, W + 1): the gap row / column is the zero padding of both
// neighbours), the regions tile the CANVAS; `ks` of `ksplit` K ranges computes channel blocks [ks * kb_per, ...) into the fp32 workspace
// [range][pixel][Npad], h16p_splitk_reduce_kernel adds the ranges in fixed order and applies the layer's epilogue.
struct h16p_cv_t {
    int ksplit, kb_per;        // K ranges (1 = off) and 32-channel blocks per range
    int gy, gx, py, px;        // canvas: images per canvas (rows x columns) and their periods in pixels
    unsigned int my, mx;       // v / py == (v * my) >> 20 for every canvas coordinate v (launcher: canvas sides <= 4095, periods <= 255)
    int nsp;                   // workgroup regions of all canvases together
    int order;                 // 1: regions fastest in the workgroup order (an XCD shares one weight slice), 0: N tiles / K ranges fastest
    int fix;                   // K ranges: 1 = the last-arriving workgroup of a (region, N tile) adds the partial sums and applies the epilogue
                               // itself (arrival counters at the head of the workspace), 0 = h16p_splitk_reduce_kernel does
};

// LEAN (round 5; S = 1, 64-wide N tile): TWO workgroups per compute unit -- one patch buffer (reloaded per 32-channel block behind a barrier,
// as S = 2) and the three weight stages packed to their 12 KiB (the four pad DMA slots of a step land in the patch buffer's pad piece): 76
// KiB of LDS, at most 128 VGPRs.  The 64 -> 64-channel layers (VGG-16 features.2 at 224 x 224, ResNet-18 layer1) have two channel blocks:
// with one workgroup per CU the patch load, six steps and a 64 KiB epilogue ran strictly one after the other (490 TFLOP/s); now one
// workgroup's memory phases lie under the other's products.
// UNP (round 5; LEAN only; `reserved1` bit 7): the layer is the input gradient of a convolution whose ReLU output fed a 2 x 2 / stride-2
// max-pool (torchvision VGG-16): `in` is the gradient w.r.t. the POOL's output [B, Hin / 2, Win / 2, in_cstride] and `in2` that pool's arg-max
// bytes [B, Hin / 2, Win / 2, in2_cstride] (spaa_maxpool_fwd's format); the patch of the pool's INPUT gradient -- g if the pixel is the
// window's first maximum and that maximum is positive, else 0: spaa_maxpool_bwd with its ReLU gate -- is formed in registers on its way
// to LDS (two loads, eight byte compares and one 16-byte LDS write per piece) instead of being read from a tensor that a separate
// launch wrote: 154 MB read instead of 411 MB written and read at VGG-16's first pool.
template <int BN, int S = 1, bool CV = false, bool LEAN = false, bool UNP = false>
__global__ __launch_bounds__(512, LEAN ? 4 : 1) void h16p_kernel(const spaa_tapconv_t p, const int wg_y, const int wg_x, const int n_tiles, const h16p_cv_t geo) {
    static_assert(!CV || S == 1, "canvas / K-range form: stride-1 layers");
    typedef h16p_geo<S> G;
    constexpr int OH = G::OH, NB = G::NB, PW = G::PW, NPX = G::NPX, PPW = G::PPW, PATCH_BYTES = G::PATCH_BYTES;
    constexpr int TJ = BN / 16;
    constexpr int NW = 8;
    constexpr int W_TAP = BN * 64;                     // one tap's weight rows (32 fp16 each)
    constexpr int W_PIECES = 3 * BN / 16;              // per step (three taps)
    constexpr int WPW = (W_PIECES + NW - 1) / NW;      // 3 (BN = 128) or 2 (BN = 64: 12 pieces, the last four DMA slots are pad)
    static_assert(!LEAN || (S == 1 && BN == 64 && !CV), "lean form: stride 1, 64-wide N tile");
    static_assert(!UNP || LEAN, "pool-adjoint prologue: the two-workgroup form (one patch buffer)");
    constexpr int NBUF = LEAN ? 1 : G::NBUF;
    constexpr int WS_BYTES = LEAN ? W_PIECES * 1024 : WPW * NW * 1024;
    static_assert(!LEAN || G::P_PIECES < PPW * 8, "lean form: the patch buffer ends in a pad piece");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const wsm = smem + NBUF * PATCH_BYTES;
    unsigned char* const wdump = smem + (PPW * 8 - 1) * 1024;    // (LEAN: where the pad DMA slots of a weight step write their zeros)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const spaa_tapclass_t cl = p.cls[0];
    const int Cin = p.Cin, H = p.Hin, W = p.Win;

    int n_blk, img, oy0, ox0, ks = 0;   // (CV: img = the canvas, (oy0, ox0) = the region's origin on it)
    int region = 0;                     // (CV: index of the region among all canvases' regions)
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        if constexpr (CV) {
            int combo;
            if (geo.order) {
                combo = t / geo.nsp;
                t -= combo * geo.nsp;
            } else {
                const int nc = n_tiles * geo.ksplit;
                combo = t % nc;
                t /= nc;
            }
            n_blk = (combo % n_tiles) * BN;
            ks = combo / n_tiles;
        } else {
            n_blk = (t % n_tiles) * BN;
            t /= n_tiles;
        }
        region = t;
        ox0 = (t % wg_x) * OW;
        t /= wg_x;
        oy0 = (t % wg_y) * OH;
        img = t / wg_y;
    }
    const int row_bytes = p.in_cstride * 2;
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(UNP ? (H >> 1) * (W >> 1) : H * W) * (uint32_t)row_bytes;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int K64 = (cl.K + 63) & ~63;
    const int nfold = p.nfold > 1 ? p.nfold : 1;
    const int ntaps = cl.ntaps;
    const int spk = (ntaps + 2) / 3;            // steps of three taps per channel block (2 or 3)
    const int npad = (p.Cout * nfold + 127) & ~127;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_half);
    const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane((uint32_t)npad * (uint32_t)K64 * 2u), 0x00020000);
    typedef const __attribute__((address_space(4))) int* cint_ptr;
    cint_ptr ctaps = (cint_ptr)(uintptr_t)(p.taps + 2 * cl.tap_off);

    // TWO SOURCES (unfolded layers, `in2` set): the layer's LAST Cin2 input channels come from a second tensor of the same B x H x W
    // -- conv(a, Wa) + conv(b, Wb) as one convolution over the concatenated channels, as the Winograd kernel's two-source form:
    // `conv5(x4) + skipConv3(x2)` (/root/reference/src/python/models.py:294,298) and `conv3^T(g3) + skipConv3^T(g5)` in fp16 storage
    // patch origin relative to the first sampled pixel of the tile: the layer's first tap (a same-size 3 x 3 layer: -1; an UNPADDED one: 0,
    // its input gradient: -2 -- Inception-v3's Conv2d_2a / Conv2d_4a, classifier.py:29-33; the taps span at most 3 x 3 pixels)
    const int py0 = p.tap_range[0], px0 = p.tap_range[2];
    // chunk swizzle of a patch pixel's 64-byte row: S = 1 as the weights' (16 consecutive pixels: conflict-free); S = 2: a fragment
    // reads every second pixel (128 bytes apart): lane pairs rotate through the four chunks (two-way conflicts at worst)
    auto pswz = [](const int q) { return S == 1 ? swz64(q) : ((q >> 2) & 3); };
    const bool two = !UNP && p.in2 != nullptr && nfold == 1;
    const int kb1 = two ? (Cin - p.Cin2) >> 5 : 0x7fffffff;      // first channel block of the second source
    const int row_bytes2 = two ? p.in2_cstride * 2 : 0;
    const auto rsrc_in2 = two ? rsrc_or_empty(p.in2, (int64_t)p.B * (H * W) * row_bytes2) : rsrc_in;
    // ---- patch staging: piece i (16 consecutive patch pixels) -> wave i % 8; lane -> (pixel lane >> 2, physical chunk lane & 3)
    // (round 5: the pixel index and chunk of a lane's PPW pieces are computed ONCE -- the per-block recomputation, a division by the
    // patch width per piece, and the per-tap fragment addresses below were 168 VALU instructions per 96-MFMA step: as many issue
    // cycles as the MFMAs themselves)
    // canvas pixel (vy, vx) of canvas `img` -> image pixel index; false: a gap, past the last image, off the canvas
    // (24-bit multiplications: coordinates <= 4095, multipliers < 2^19, B * H * W < 2^24 -- the launcher checks; csrc/tapconv_wino.hip)
    const int ipc = CV ? geo.gy * geo.gx : 1;
    auto canvas_pixel = [&](const int vy, const int vx, int& o) -> bool {
        const int sy = (int)(__umul24((unsigned int)vy, geo.my) >> 20), sx = (int)(__umul24((unsigned int)vx, geo.mx) >> 20);
        const int iy = vy - (int)__umul24(sy, geo.py), ix = vx - (int)__umul24(sx, geo.px);
        const int im = img * ipc + (int)__umul24(sy, geo.gx) + sx;
        o = (int)__umul24(__umul24(im, H) + iy, W) + ix;
        return vy >= 0 && vx >= 0 && iy < H && ix < W && sy < geo.gy && sx < geo.gx && im < p.B;
    };
    // (the chunk swizzle looks at bits 2-3 of the patch pixel index = bits of lane >> 2: the same for all of a lane's pieces)
    int ppix[PPW];
    const int pch16 = ((lane & 3) ^ pswz(lane >> 2)) << 4;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = (wave + NW * i) * 16 + (lane >> 2);          // patch pixel
        const int pr = q / PW, pc = q - pr * PW;
        const int iy = oy0 * S + py0 + pr, ix = ox0 * S + px0 + pc;
        bool ok = q < NPX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        int pxi = (img * H + iy) * W + ix;
        if constexpr (CV) ok = canvas_pixel(iy, ix, pxi) && q < NPX;
        // (UNP: the pooled pixel and the position 2 ky + kx of this pixel in its window)
        if constexpr (UNP) pxi = (((img * (H >> 1) + (iy >> 1)) * (W >> 1) + (ix >> 1)) << 2) | ((iy & 1) << 1) | (ix & 1);
        ppix[i] = ok ? pxi : -1;
    }
    const auto rsrc_arg = UNP ? rsrc_or_empty(p.in2, (int64_t)p.B * (H >> 1) * (W >> 1) * p.in2_cstride) : rsrc_in;
    // 32-channel blocks of this workgroup: all of them, or (CV) the K range [kb0, kb0 + nkb)
    const int kb0 = CV ? ks * geo.kb_per : 0;
    auto dma_patch = [&](const int buf, const int kbl) {
        const int kb = kb0 + kbl;
        if constexpr (UNP) {
            // two batches (three and two pieces): at most 18 registers of loaded data live at a time
            constexpr int HB = (PPW + 1) / 2;
#pragma unroll
            for (int i0 = 0; i0 < PPW; i0 += HB) {
                u32x4 gv[HB];
                u32x2 av[HB];
#pragma unroll
                for (int i = i0; i < i0 + HB && i < PPW; ++i) {
                    int pp = ppix[i];
                    asm volatile("" : "+v"(pp));     // (the two offsets are recomputed per block: not ten more registers across the K loop)
                    const bool ok = pp >= 0;
                    gv[i - i0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, ok ? (pp >> 2) * row_bytes + pch16 : (int)0x80000000,
                                                                       (p.in_coff + kb * 32) * 2, 0);
                    av[i - i0] = __builtin_amdgcn_raw_buffer_load_b64(rsrc_arg, ok ? (pp >> 2) * p.in2_cstride + (pch16 >> 1) : (int)0x80000000,
                                                                      p.in2_coff + kb * 32, 0);
                }
#pragma unroll
                for (int i = i0; i < i0 + HB && i < PPW; ++i) {
                    const unsigned int code = 0x80u | (unsigned int)(ppix[i] & 3);
                    u32x4 o;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {     // two fp16 values (channels 2 e2, 2 e2 + 1 of the chunk) per register
                        const unsigned int a0 = (av[i - i0][e2 >> 1] >> (16 * (e2 & 1))) & 0xffu, a1 = (av[i - i0][e2 >> 1] >> (16 * (e2 & 1) + 8)) & 0xffu;
                        const unsigned int g = gv[i - i0][e2];
                        o[e2] = (a0 == code ? (g & 0xffffu) : 0u) | (a1 == code ? (g & 0xffff0000u) : 0u);
                    }
                    *reinterpret_cast<u32x4*>(smem + buf * PATCH_BYTES + (wave + NW * i) * 1024 + lane * 16) = o;
                }
            }
            return;
        }
        const bool s2 = kb >= kb1;     // (uniform)
        const int rb = s2 ? row_bytes2 : row_bytes, cb = s2 ? (p.in2_coff + (kb - kb1) * 32) * 2 : (p.in_coff + kb * 32) * 2;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int off = ppix[i] >= 0 ? ppix[i] * rb + pch16 : (int)0x80000000;
            if (s2) dma16(rsrc_in2, smem + buf * PATCH_BYTES + (wave + NW * i) * 1024, off, cb);
            else dma16(rsrc_in, smem + buf * PATCH_BYTES + (wave + NW * i) * 1024, off, cb);
        }
    };
    // ---- weights of step (kb, s): taps 3 s .. 3 s + 2; piece q = wave + 8 i -> (tap q / (BN / 16), 16-row block q % (BN / 16))
    const int w_voff = (lane >> 2) * K64 * 2 + (((lane & 3) ^ swz64(lane >> 2)) << 4);
    auto dma_w = [&](const int stage, const int kb, const int s) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int q = wave + NW * i;
            const int tl = q / (BN / 16), rb = q - tl * (BN / 16);
            const int soff = (n_blk + 16 * rb) * K64 * 2 + ((3 * s + tl) * Cin + (kb0 + kb) * 32) * 2;
            dma16(rsrc_w, (LEAN && q >= W_PIECES) ? wdump : wsm + stage * WS_BYTES + q * 1024, (q < W_PIECES && 3 * s + tl < ntaps) ? w_voff : (int)0x80000000, soff);
        }
    };
    const int w_addr_l = (lane & 15) * 64 + (((lane >> 4) ^ swz64(lane & 15)) * 16);
    // fragment address (inside a patch buffer) of every (tap, pixel block) of this lane: the taps are read once here
    // (pixel block b = (row (NB / 2) wave + (b >> 1), columns 16 (b & 1) ..).  The second column block is 16 S patch pixels further:
    // the same chunk swizzle (bits 3 resp. 2-3 of q unchanged by + 16 / + 32), so its address is the first's + 1024 S bytes -- an
    // immediate offset; one register per (tap, row))
    constexpr int NR = NB / 2;
    // (UNP: the two rows' addresses -- below 64 KiB -- share a register: nine registers less where 128 are all there is; one shift / mask
    // per fragment read)
    constexpr bool FPK = UNP && NR == 2;
    int faddr_[9][FPK ? 1 : NR];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const bool has = t < ntaps;
        const int dy = has ? ctaps[2 * t] : py0, dx = has ? ctaps[2 * t + 1] : px0;
        int fa[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int orow = S == 1 ? 2 * wave + r : wave, ocol = lane & 15;
            const int q = (orow * S + dy - py0) * PW + ocol * S + dx - px0;
            fa[r] = q * 64 + (((lane >> 4) ^ pswz(q)) << 4);
        }
        if constexpr (FPK) faddr_[t][0] = fa[0] | (fa[NR - 1] << 16);
        else {
#pragma unroll
            for (int r = 0; r < NR; ++r) faddr_[t][r] = fa[r];
        }
    }
    auto faddr = [&](const int t, const int r) -> int {
        if constexpr (FPK) {
            int v = faddr_[t][0];
            asm volatile("" : "+v"(v));     // (unpacked where it is used: the optimiser would hoist both halves out of the K loop again)
            return r ? (int)((unsigned int)v >> 16) : (v & 0xffff);
        } else return faddr_[t][r];
    };

    f32x4 acc[NB][TJ];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkb = CV ? ((Cin >> 5) - kb0 < geo.kb_per ? (Cin >> 5) - kb0 : geo.kb_per) : (Cin >> 5);
    const int nsteps = spk * nkb;
    dma_patch(0, 0);
    dma_w(0, 0, 0);
    if (nsteps > 1) dma_w(1, 0, 1);
    int st = 0;
    for (int kb = 0; kb < nkb; ++kb) {
        const unsigned char* pb = smem + (NBUF == 2 ? (kb & 1) : 0) * PATCH_BYTES;
#pragma unroll
        for (int s = 0; s < 3; ++s) {   // (expanded: the tap index 3 s + tl below is a constant, `faddr` stays in registers)
            if (s >= spk) break;
            const int step = spk * kb + s;
            if (NBUF == 1 && s == 0 && kb > 0) {   // one patch buffer: everybody is done with the previous block's patch
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                dma_patch(0, kb);
            }
            // this wave's pieces of the step's weights (and, at s == 0, of the block's patch) have landed
            // (one patch buffer: the block's patch was requested just above, AFTER the next step's weights: everything must have landed)
            if (step + 1 >= nsteps || (NBUF == 1 && s == 0 && kb > 0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (NBUF == 2 && s != 0 && kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + PPW) : "memory");   // (the next block's patch, requested at s == 0)
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (step + 2 < nsteps) {
                const int st2 = st >= 1 ? st - 1 : 2;
                const int kb2 = (step + 2) / spk, s2 = step + 2 - kb2 * spk;
                dma_w(st2, kb2, s2);
            }
            if (NBUF == 2 && s == 0 && kb + 1 < nkb) dma_patch((kb + 1) & 1, kb + 1);
            // software pipeline over the step's (tap, 16-channel block) units: the weight fragment of unit u + 1 and, at a tap's first
            // unit, the pixel fragments of the NEXT tap are requested before unit u's MFMAs are issued (the compiler's own order was
            // read -> wait for all of LDS -> four MFMAs -> read ...: a full LDS round trip per four MFMAs, the matrix cores at a third)
            // (LEAN: four waves per SIMD hide the round trip; one set of pixel fragments, read at the tap's start -- 16 registers less)
            constexpr int NBF = LEAN ? 1 : 2;
            h8 bfs[NBF][NB], wfs[2];
            if constexpr (NBF == 2) {
#pragma unroll
                for (int b = 0; b < NB; ++b) bfs[0][b] = *reinterpret_cast<const h8*>(pb + faddr(3 * s, b >> 1) + (b & 1) * (1024 * S));
            }
            wfs[0] = *reinterpret_cast<const h8*>(wsm + st * WS_BYTES + w_addr_l);
#pragma unroll
            for (int tl = 0; tl < 3; ++tl) {
                if (3 * s + tl >= ntaps) break;   // (uniform: a tap list that is not a multiple of three ends inside a step)
                const bool more = tl + 1 < 3 && 3 * s + tl + 1 < ntaps;   // (uniform) another tap in this step
                const unsigned char* wc = wsm + st * WS_BYTES + tl * W_TAP + w_addr_l;
                if constexpr (NBF == 1) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) bfs[0][b] = *reinterpret_cast<const h8*>(pb + faddr(3 * s + tl, b >> 1) + (b & 1) * (1024 * S));
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int u = LEAN ? 0 : tl * TJ + j;
                    if constexpr (LEAN) wfs[0] = *reinterpret_cast<const h8*>(wc + j * 1024);
                    else if (j + 1 < TJ) wfs[(u + 1) & 1] = *reinterpret_cast<const h8*>(wc + (j + 1) * 1024);
                    else if (more) wfs[(u + 1) & 1] = *reinterpret_cast<const h8*>(wc + W_TAP);
                    if (NBF == 2 && j == 0 && more) {
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            bfs[(tl + 1) & (NBF - 1)][b] = *reinterpret_cast<const h8*>(pb + faddr(3 * s + tl + 1, b >> 1) + (b & 1) * (1024 * S));
                    }
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfs[u & 1], bfs[tl & (NBF - 1)][b], acc[b][j], 0, 0, 0);
                    // order: this unit's LDS reads (for the units to come) first, then its MFMAs
                    if constexpr (NBF == 2) {
                        __builtin_amdgcn_sched_group_barrier(0x100, NB + 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
                    }
                }
            }
            st = st == 2 ? 0 : st + 1;
        }
    }

    // ---- optional SECOND SOURCE of a folded stride-2 transposed layer (nfold = 4): a 1 x 1 convolution of a tensor at OUTPUT
    // resolution, `in2` [B, Hout, Wout, in2_cstride] fp16 (channels in2_coff .. + Cin2, Cin2 = 32 or 64) through `w2_split` = ONE fp16
    // matrix [Cout][Cin2], added before bias / residual / activation: ShadingNetSPAA's `transConv1(x5) + skipConv2(x1)`
    // (/root/reference/src/python/models.py:293,299) and, backward, `conv2^T(g2) + skipConv2^T(g6)` as one launch without the R2 / t1
    // tensors.  Every element of `in2` is needed once: its fragments go from global memory to registers (as csrc/tapconv_x6p.hip);
    // GEMM columns n_blk + 16 j .. + 15 lie in ONE parity class (launcher: Cout % 16 == 0), whose output pixel of class-grid pixel
    // (y, x) is (2 y + cy, 2 x + cx).
    if constexpr (S == 1 && !CV) if (p.in2 != nullptr && nfold > 1) {
        const int row2 = p.in2_cstride * 2;
        const auto rsrc_in2 = rsrc_or_empty(p.in2, (int64_t)p.B * p.Hout * p.Wout * row2);
        const auto rsrc_w2 = rsrc_or_empty(p.w2_split, (int64_t)p.Cout * p.Cin2 * 2);
        const int nk2 = p.Cin2 >> 5;
        for (int k2 = 0; k2 < nk2; ++k2) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int ng = n_blk + 16 * j;
                const int fc = ng / p.Cout, n0 = ng - fc * p.Cout;   // (uniform) class and first channel of this 16-column block
                if (fc >= nfold) break;
                const int cy = fc >> 1, cx = fc & 1;
                const h8 wf = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(
                    rsrc_w2, ((n0 + (lane & 15)) * p.Cin2 + k2 * 32 + 8 * (lane >> 4)) * 2, 0, 0));
                h8 xf[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int y = oy0 + 2 * wave + (b >> 1), x = ox0 + 16 * (b & 1) + (lane & 15);
                    const int oy = 2 * y + cy, ox = 2 * x + cx;
                    const bool ok = y < p.Hm && x < p.Wm && oy < p.Hout && ox < p.Wout;
                    const int off = ok ? ((img * p.Hout + oy) * p.Wout + ox) * row2 + (p.in2_coff + k2 * 32 + 8 * (lane >> 4)) * 2 : (int)0x80000000;
                    xf[b] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in2, off, 0, 0));
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[b], acc[b][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D layout of a 16x16 block: column (lane & 15) = pixel, rows 4 (lane >> 4) + e = 4 consecutive channels
    // (CV, K range of a split layer: raw partial sums into the workspace slice [ks][pixel][Npad] through the same code -- a descriptor
    // without bias / residual / gates / masks, fp32; the second pass applies the layer's epilogue)
    spaa_tapconv_t pq = p;
    if constexpr (CV) {
        if (geo.ksplit > 1) {
            pq.out = p.splitk_ws + (geo.fix ? SPAA_SPLITK_HDR_FLOATS : 0) + (size_t)ks * ((size_t)p.B * H * W) * (size_t)((p.Cout + 127) & ~127);
            pq.out_cstride = (p.Cout + 127) & ~127;
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
            pq.io_dtype = p.io_dtype & ~SPAA_IO_OUT_F16;
        }
    }
    const spaa_tapconv_t& e = CV ? pq : p;
    const bool vec = !((e.Cout | e.out_cstride | e.out_coff) & 3) &&
                     (e.add == nullptr || !((e.add_cstride | e.add_coff) & 3)) &&
                     (e.gate == nullptr || !((e.gate_cstride | e.gate_coff) & 3)) &&
                     (e.gate2 == nullptr || !((e.gate2_cstride | e.gate2_coff) & 3));
    // through LDS (a private region per wave, free once every wave has left the K loop): a lane then owns 4 channels of a pixel
    // and BN / 4 consecutive lanes its whole channel row -- 256-byte (fp16) / 512-byte (fp32) contiguous segments per pixel for
    // the output and for every epilogue operand, instead of the MFMA layout's 16 pixels x 32 bytes per instruction
    constexpr int ROWB = BN * 4 + 16;                  // (+16: the 16 pixels of a fragment write to distinct banks)
    constexpr int LPP = BN / 4, PPI = 64 / LPP;        // lanes per pixel, pixels per instruction
    __syncthreads();
    unsigned char* const eb = smem + wave * (32 * ROWB);
    const int ch = 4 * (lane % LPP);
    // the operand combinations of the fp16-storage networks (bias, residual, ReLU, byte-mask gates, byte mask out, second gated
    // output) without a branch: absent tensors are zero-record buffer descriptors (loads give 0, stores are dropped), so that
    // the operand loads of four pixels per lane are in flight together; anything else: the shared store4_t
    const bool fast = fast_epi_ok(e, vec);
    // ---- POOL (round 5; `reserved1` bit 6; S = 1, unfolded, image-aligned regions): the layer's ReLU and the 2 x 2 / stride-2 max-pool that
    // follows it (torchvision VGG-16 `features`: conv -> ReLU -> MaxPool2d(2, 2), /root/reference/src/python/classifier.py:21-24) in this
    // epilogue: `out` is the POOLED tensor [B, Hout / 2, Wout / 2, out_cstride], `mask_out` the pool's arg-max bytes [B, Hout / 2, Wout / 2,
    // Cout] (code 2 ky + kx of the first maximum | 0x80 if it is positive: spaa_maxpool_fwd's format; NULL = not needed) -- the full-size
    // activation (VGG-16 conv1_2 at 224 x 224: 411 MB written and read again per forward pass) never reaches HBM.  A wave owns output rows
    // 2 w, 2 w + 1: the vertical pair of a window is in ONE lane (accumulators b and b + 2), the horizontal pair in neighbouring lanes
    // (a quad-permute DPP move); values are rounded to the storage type BEFORE they are compared, in the window order of
    // maxpool2x2_fwd_kernel: bitwise the pooled values and arg-max bytes of the separate launches (finite values).
    if constexpr (S == 1 && !CV) {
        if (p.reserved1 & 64) {
            const bool o16 = (p.io_dtype & SPAA_IO_OUT_F16) != 0;
            const int Hp = p.Hout >> 1, Wp = p.Wout >> 1;
            const int64_t npool = (int64_t)p.B * Hp * Wp;
            const auto r_out = rsrc_or_empty(p.out, npool * p.out_cstride * (o16 ? 2 : 4));
            const auto r_arg = rsrc_or_empty(p.mask_out, npool * p.Cout);
            const auto r_bias = rsrc_or_empty(p.bias, (int64_t)p.Cout * 4);
            const int c = lane & 15, q = lane >> 4;
            unsigned char* const ab = eb + 16 * ROWB;      // arg-max bytes of the wave's 16 pooled pixels: [16][BN]
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int n = n_blk + 16 * j + 4 * q;
                    const u32x4 bv = __builtin_amdgcn_raw_buffer_load_b128(r_bias, n < p.Cout ? n * 4 : (int)0x80000000, 0, 0);
                    f32x4 best;
                    unsigned int am = 0;
#pragma unroll
                    for (int e_ = 0; e_ < 4; ++e_) {
                        const float b_ = __uint_as_float(bv[e_]);
                        float r0 = fmaxf(acc[bb][j][e_] + b_, 0.f), r1 = fmaxf(acc[2 + bb][j][e_] + b_, 0.f);
                        if (o16) r0 = (float)(_Float16)r0, r1 = (float)(_Float16)r1;
                        // the neighbouring column's pair (lanes 2 i <-> 2 i + 1 swapped: quad_perm [1, 0, 3, 2])
                        const float n0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r0), 0xB1, 0xF, 0xF, true));
                        const float n1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(r1), 0xB1, 0xF, 0xF, true));
                        float m_ = r0;
                        unsigned int k = 0;
                        if (n0 > m_) m_ = n0, k = 1;
                        if (r1 > m_) m_ = r1, k = 2;
                        if (n1 > m_) m_ = n1, k = 3;
                        best[e_] = m_;
                        am |= (k | (m_ > 0.f ? 0x80u : 0u)) << (8 * e_);
                    }
                    if (!(c & 1)) {     // (even columns hold the windows)
                        const int pp = 8 * bb + (c >> 1);
                        *reinterpret_cast<f32x4*>(eb + pp * ROWB + (16 * j + 4 * q) * 4) = best;
                        *reinterpret_cast<uint32_t*>(ab + pp * BN + 16 * j + 4 * q) = am;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int py = (oy0 >> 1) + wave;
            const int n = n_blk + ch;
            if (py < Hp) {
#pragma unroll
                for (int i = 0; i < 16 / PPI; ++i) {
                    const int pr = i * PPI + lane / LPP, px = (ox0 >> 1) + pr;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);
                    const uint32_t am = *reinterpret_cast<const uint32_t*>(ab + pr * BN + ch);
                    const bool ok = n < p.Cout && px < Wp;
                    const int o = (img * Hp + py) * Wp + px;
                    const float v[4] = {a[0], a[1], a[2], a[3]};
                    if (o16) fast_io<_Float16>::st(r_out, ok ? (o * p.out_cstride + p.out_coff + n) * 2 : (int)0x80000000, v);
                    else fast_io<float>::st(r_out, ok ? (o * p.out_cstride + p.out_coff + n) * 4 : (int)0x80000000, v);
                    __builtin_amdgcn_raw_buffer_store_b32(am, r_arg, ok ? o * p.Cout + n : (int)0x80000000, 0, 0);
                }
            }
            return;
        }
    }
#define H16P_TO_LDS(hb)                                                                                            \
    _Pragma("unroll") for (int bb = 0; bb < 2; ++bb)                                                               \
    _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                                                 \
        *reinterpret_cast<f32x4*>(eb + (16 * bb + (lane & 15)) * ROWB + (16 * j + 4 * (lane >> 4)) * 4) = acc[2 * (hb) + bb][j]; \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define H16P_EPI(T, hb)                                                                                            \
    {                                                                                                              \
        H16P_TO_LDS(hb)                                                                                            \
        const int oy = oy0 + (NB / 2) * wave + (hb);                                                               \
        if (CV || oy < e.Hm) {                                                                                     \
            const size_t orow = ((size_t)img * e.Hm + oy) * e.Wm + ox0;                                            \
            for (int i = 0; i < 32 / PPI; ++i) {                                                                   \
                const int pr = i * PPI + lane / LPP;                                                               \
                const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);                          \
                float v[4] = {a[0], a[1], a[2], a[3]};                                                             \
                if constexpr (CV) {                                                                                \
                    int o_;                                                                                        \
                    if (canvas_pixel(oy, ox0 + pr, o_)) store4_t<T>(e, (size_t)o_, n_blk + ch, v, vec);            \
                } else if (ox0 + pr < e.Wm) {                                                                      \
                    if (e.nfold > 1) store4_fold_t<T>(e, (int)(orow + pr), e.B * e.Hm * e.Wm, e.Hm * e.Wm, n_blk + ch, v, vec); \
                    else store4_t<T>(e, orow + pr, n_blk + ch, v, vec);                                            \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    }
#define H16P_EPI_FAST(T, hb)                                                                                       \
    {                                                                                                              \
        H16P_TO_LDS(hb)                                                                                            \
        const int oy = oy0 + (NB / 2) * wave + (hb);                                                               \
        if (CV || oy < e.Hm) {                                                                                     \
            /* output pixel of class-grid pixel (oy, ox0 + pr): itself, or (2 oy + cy, 2 x + cx) of a folded transposed layer; \
               CV: the image pixel of canvas pixel (oy, ox0 + pr), if it is one */                                   \
            const int orow = (img * e.Hout + fs * oy + cy) * e.Wout + fs * ox0 + cx;                               \
            const bool row_ok = n_ok && fs * oy + cy < e.Hout;                                                     \
            _Pragma("unroll 1") for (int i0 = 0; i0 < 32 / PPI; i0 += 4) {                                         \
                fast_pre_t<T> pre[4];                                                                              \
                int oo[4];                                                                                         \
                bool okk[4];                                                                                       \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
                    const int pr = (i0 + i) * PPI + lane / LPP;                                                    \
                    oo[i] = orow + fs * pr;                                                                        \
                    okk[i] = row_ok && ox0 + pr < e.Wm && fs * (ox0 + pr) + cx < e.Wout;                           \
                    if constexpr (CV) okk[i] = canvas_pixel(oy, ox0 + pr, oo[i]) && n_ok;                          \
                    pre[i] = fast_epi_load<T>(fe, e, oo[i], n, okk[i]);                                            \
                }                                                                                                  \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
                    const int pr = (i0 + i) * PPI + lane / LPP;                                                    \
                    const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);                      \
                    fast_epi_store<T, f32x4, false, CV>(fe, e, oo[i], n, okk[i], a, pre[i]);                       \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
    }
    if (fast) {
        // (folded layer: GEMM column n_blk + ch = class * Cout + channel; the lane's class is fixed, so is its output parity)
        const int ng = n_blk + ch;
        const int fc = nfold > 1 ? ng / e.Cout : 0;
        const int n = ng - fc * e.Cout;
        const bool n_ok = fc < nfold && n < e.Cout;
        const int fs = nfold > 1 ? 2 : 1, cy = fc >> 1, cx = fc & 1;
        fast_epi_t fe = make_fast_epi(e, n_ok ? n : 0);
        if constexpr (CV) fe.out_sc1 = geo.ksplit > 1 && geo.fix;
        if (e.io_dtype & SPAA_IO_OUT_F16) {
            H16P_EPI_FAST(_Float16, 0)
            if constexpr (NB == 4) H16P_EPI_FAST(_Float16, 1)
        } else {
            H16P_EPI_FAST(float, 0)
            if constexpr (NB == 4) H16P_EPI_FAST(float, 1)
        }
    } else if (e.io_dtype & SPAA_IO_OUT_F16) {
        H16P_EPI(_Float16, 0)
        if constexpr (NB == 4) H16P_EPI(_Float16, 1)
    } else {
        H16P_EPI(float, 0)
        if constexpr (NB == 4) H16P_EPI(float, 1)
    }
#undef H16P_EPI
#undef H16P_EPI_FAST
#undef H16P_TO_LDS
    // K ranges, round 6: the second pass inside this kernel (as csrc/tapconv_wino.hip).  Every workgroup of a (region, N tile) bumps the
    // tile's arrival counter once its partial sums are visible device-wide; the LAST to arrive adds the K ranges in the fixed order 0, 1, 2,
    // ... (its own read back: every bit of the result is the two-pass form's) and applies the layer's epilogue through the same store4_t.
    // Nobody waits for anybody.  The counters (int32, head of the workspace) are zero before and after every launch.
    if constexpr (CV) {
        if (geo.ksplit > 1 && geo.fix) {
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
            const int M = p.B * H * W, npad = (p.Cout + 127) & ~127;
            const auto rws = rsrc_or_empty(p.splitk_ws + SPAA_SPLITK_HDR_FLOATS, (int64_t)geo.ksplit * M * npad * 4);
            const bool pvec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                              (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                              (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                              (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
            for (int i = tid; i < OH * OW * LPP; i += 512) {
                const int qd = i % LPP, pxl = i / LPP;
                const int n0 = n_blk + 4 * qd;
                int o;
                if (!canvas_pixel(oy0 + pxl / OW, ox0 + pxl % OW, o) || n0 >= p.Cout) continue;
                f32x4 sum = {0.f, 0.f, 0.f, 0.f};
                for (int s_ = 0; s_ < geo.ksplit; ++s_) {   // agent-scope loads: served from where the other XCDs' stores went
                    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rws, ((s_ * M + o) * npad + n0) * 4, 0, SPAA_AUX_SC1);
                    sum += f32x4{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
                }
                float v[4] = {sum[0], sum[1], sum[2], sum[3]};
                if (p.io_dtype & SPAA_IO_OUT_F16) store4_t<_Float16>(p, (size_t)o, n0, v, pvec);
                else store4_t<float>(p, (size_t)o, n0, v, pvec);
            }
        }
    }
}

// second pass of a K-split layer: out = epilogue( sum over the K ranges, in fixed order ), 4 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void h16p_splitk_reduce_kernel(const spaa_tapconv_t p, const int M, const int npad) {
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
    store4_t<T>(p, (size_t)m, n0, v, vec);
}

// ---- launch plan of the canvas / K-range form: N tile, canvas layout, K ranges (one decision for the launcher and for
// spaa_tapconv_h16p_plan).  Cost model in us, fitted to tools/lab/h16p_cv_time.py (profiles/r05_h16p_cv_time.txt; batch 64: ResNet-18
// layer3 / layer4, VGG-16's 14 x 14 block, forced N tiles and K ranges): a 64-wide workgroup needs 16.7 us + 3.4 per 32-channel block, a
// 128-wide one 22.3 + 5.45; a launch takes ceil(workgroups / CUs) such rounds (a partly filled round after the first counts half); a K
// split adds its second pass (5 us + the partial sums at 5 TB/s).
struct h16p_plan_t {
    int bn, ksplit, kb_per, canvas, gy, gx, py, px, ncanvas, wg_y, wg_x, n_tiles;
    int64_t nwg;
};
inline int h16p_cdiv(int a, int b) { return (a + b - 1) / b; }
inline int h16p_ncu() {
    static int ncu[SPAA_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SPAA_MAX_DEVICES) return 256;
    if (ncu[dev] == 0 && hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu[dev] = 256;
    return ncu[dev] > 0 ? ncu[dev] : 256;
}
// the canvas / K-range form serves: stride 1, unfolded, one source, taps within [-1, 1]^2, same-size output, < 2^24 pixels
inline bool h16p_cv_shape_ok(const spaa_tapconv_t& d) {
    return d.s_in == 1 && d.s_out == 1 && d.nfold <= 1 && d.in2 == nullptr && d.Hm == d.Hin && d.Wm == d.Win && d.Hm == d.Hout && d.Wm == d.Wout &&
           (int64_t)d.B * d.Hout * d.Wout < ((int64_t)1 << 24) && d.Hout <= 4095 && d.Wout <= 4095 && d.Cin >= 32 && (d.Cin % 32) == 0;
}
// `force_bn`: 0 = choose, 64 / 128 (64 only for Cout <= 64 ... any); `force_ks`: 0 = choose, else that many K ranges (clamped to one block per range)
inline h16p_plan_t h16p_make_plan(const spaa_tapconv_t& d, const int ncu, const int force_bn, const int force_ks, const bool allow_split) {
    h16p_plan_t pl = {};
    const int H = d.Hout, W = d.Wout, B = d.B;
    constexpr int OH = h16p_geo<1>::OH;
    const int pwy = h16p_cdiv(H, OH), pwx = h16p_cdiv(W, OW);
    const int64_t plain = (int64_t)B * pwy * pwx;
    int64_t best = plain;
    int cgy = 1, cgx = 1, cnc = B, cwy = pwy, cwx = pwx;
    const int py = H + 1, px = W + 1;
    if (py <= 255 && px <= 255) {
        for (int gx = 1; gx <= B && gx * px - 1 <= 4095; ++gx) {
            const int wx = h16p_cdiv(gx * px - 1, OW);
            for (int gy = 1; gy * gx <= B + gx - 1 && gy * py - 1 <= 4095; ++gy) {
                const int wy = h16p_cdiv(gy * py - 1, OH);
                const int nc = h16p_cdiv(B, gy * gx);
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
            const int nt = h16p_cdiv(d.Cout, bn);
            for (int ks = 1; ks <= nkb; ++ks) {
                if (force_ks > 0 ? ks != (force_ks < nkb ? force_ks : nkb) : (ks > 1 && (!allow_split || nkb < 2 * ks))) continue;
                if (ks > 1 && (d.Cout & 3)) continue;
                const int kb_per = h16p_cdiv(nkb, ks), ksr = h16p_cdiv(nkb, kb_per);
                if (ksr != ks && force_ks <= 0) continue;   // (the same plan as a smaller ks)
                const int64_t nwg = regions * nt * ksr;
                double rounds = 0.5 * (double)((nwg + ncu - 1) / ncu) + 0.5 * (double)nwg / ncu;
                rounds = rounds < 1.0 ? 1.0 : rounds;
                double cost = rounds * (kb_per * (bn == 128 ? 5.45 : 3.4) + (bn == 128 ? 22.3 : 16.7));
                if (ksr > 1) cost += 5.0 + (double)ksr * (double)M * npad * 4.0 / 5e6;
                if (cv && !(d.reserved1 & 8)) cost *= 1.05;
                if (!cv && best < plain && (d.reserved1 & 8)) cost *= 1e6;   // (tests: canvas wherever it has fewer regions)
                if (ksr > 1) cost *= 1.02;
                if (cost < best_cost) {
                    best_cost = cost;
                    pl.bn = bn, pl.ksplit = ksr, pl.kb_per = kb_per, pl.canvas = cv, pl.n_tiles = nt, pl.nwg = nwg;
                }
            }
        }
    }
    if (pl.canvas) pl.gy = cgy, pl.gx = cgx, pl.ncanvas = cnc, pl.wg_y = cwy, pl.wg_x = cwx;
    else pl.gy = pl.gx = 1, pl.ncanvas = B, pl.wg_y = pwy, pl.wg_x = pwx;
    pl.py = py, pl.px = px;
    return pl;
}

}  // namespace

// The launcher's plan for the canvas / K-range form of tile 68 (`reserved1` bit 2 set; bits 0-1: N tile 0 = chosen, 1 = 64, 2 = 128;
// desc->ksplit > 1: that many K ranges, 1: none, 0: chosen): plan[0..7] = {N tile, K ranges, canvas (0 / 1), images per canvas (rows),
// (columns), workgroups, channel blocks per K range, canvases}.  A caller sizes `splitk_ws` (K ranges x B x H x W x Npad floats) from
// plan[1] and passes plan[1] back as `ksplit`.
extern "C" int spaa_tapconv_h16p_plan(const spaa_tapconv_t* desc, int32_t* plan) {
    if (desc == nullptr || plan == nullptr) return hipErrorInvalidValue;
    const spaa_tapconv_t& d = *desc;
    if (!h16p_cv_shape_ok(d) || d.ksplit < 0 || d.B < 1 || d.Cout < 1) return hipErrorInvalidValue;
    const int fb = d.reserved1 & 3;
    const h16p_plan_t pl = h16p_make_plan(d, h16p_ncu(), fb == 1 ? 64 : fb == 2 ? 128 : 0, d.ksplit, true);
    if (pl.bn == 0 || pl.ksplit < 1 || pl.n_tiles < 1) return hipErrorInvalidValue;
    plan[0] = pl.bn, plan[1] = pl.ksplit, plan[2] = pl.canvas, plan[3] = pl.gy, plan[4] = pl.gx;
    plan[5] = (int32_t)(pl.nwg > 0x7fffffff ? 0x7fffffff : pl.nwg), plan[6] = pl.kb_per, plan[7] = pl.ncanvas;
    return 0;
}

// called by spaa_tapconv_f32 (tapconv.hip) for tile 68 after the common shape checks: ONE class of four to nine taps inside
// [-1, 1]^2 sampled at stride 1 -- a 3x3 convolution or its input gradient (same input and output size), or the FOLDED form of a
// stride-2 transposed layer (nfold = 4: a 3x3 / s2 transposed convolution, the input gradient of a 3x3 / s2 convolution: four
// taps, GEMM columns = parity class * Cout + channel, output twice the input size); fp16 input, Cin % 32 == 0
int spaa_launch_tapconv_h16p(const spaa_tapconv_t& d, hipStream_t stream) {
    const int nfold = d.nfold > 1 ? d.nfold : 1;
    const int S = d.s_in;
    if (!(d.io_dtype & SPAA_IO_IN_F16) || d.w_half == nullptr || (d.Cin % 32) != 0 || d.nclass != 1 || d.cls[0].ntaps < 4 ||
        d.cls[0].ntaps > 9 || d.cls[0].K != d.cls[0].ntaps * d.Cin || (S != 1 && S != 2) || d.ksplit < 0)
        return hipErrorInvalidValue;
    if (S == 1) {
        if (nfold == 1) {
            // unfolded: any 3 x 3 tap window (padded, unpadded, the unpadded layer's input gradient: the output 2 smaller / larger)
            if (d.s_out != 1 || d.Hm != d.Hout || d.Wm != d.Wout || d.tap_range[1] - d.tap_range[0] > 2 || d.tap_range[3] - d.tap_range[2] > 2)
                return hipErrorInvalidValue;
        } else {
            if (d.Hm != d.Hin || d.Wm != d.Win || nfold != 4 || d.s_out != 2 || (d.Cout & 3) || d.Hm != (d.Hout + 1) / 2 || d.Wm != (d.Wout + 1) / 2)
                return hipErrorInvalidValue;
            if (d.tap_range[0] < -1 || d.tap_range[1] > 1 || d.tap_range[2] < -1 || d.tap_range[3] > 1) return hipErrorInvalidValue;
        }
    } else {
        // the forward form of a 3 x 3 / stride-2 convolution: unfolded, one source, taps inside a 3 x 3 box starting at the first tap
        if (nfold != 1 || d.s_out != 1 || d.Hm != d.Hout || d.Wm != d.Wout || d.in2 != nullptr || d.tap_range[1] - d.tap_range[0] > 2 ||
            d.tap_range[3] - d.tap_range[2] > 2)
            return hipErrorInvalidValue;
    }
    if (d.reserved1 & 128) {
        // (`in2` = arg-max bytes of the pool-adjoint prologue: checked with the launch below)
    } else if (d.in2 != nullptr && nfold == 1) {   // two sources of an unfolded layer: the last Cin2 channels (whole 32-channel blocks) from `in2`
        if ((d.Cin2 & 31) || d.Cin2 <= 0 || d.Cin2 >= d.Cin || (d.in2_cstride & 7) || (d.in2_coff & 7) || d.in2_coff + d.Cin2 > d.in2_cstride ||
            (int64_t)d.B * d.Hin * d.Win * d.in2_cstride * 2 >= (int64_t)1 << 31)
            return hipErrorInvalidValue;
    } else if (d.in2 != nullptr) {   // second source: folded layers, fp16 in and out, whole 16-column blocks per parity class
        if (nfold != 4 || d.w2_split == nullptr || (d.Cin2 != 32 && d.Cin2 != 64) || (d.Cout & 15) || !(d.io_dtype & SPAA_IO_OUT_F16) ||
            (d.in2_cstride & 7) || (d.in2_coff & 7) || d.in2_coff + d.Cin2 > d.in2_cstride ||
            (int64_t)d.B * d.Hout * d.Wout * d.in2_cstride * 2 >= (int64_t)1 << 31)
            return hipErrorInvalidValue;
    }
    if (d.reserved1 & 64) {
        // fused ReLU + 2 x 2 / stride-2 max-pool: `out` / `mask_out` are the POOLED tensor and its arg-max bytes
        if (S != 1 || nfold != 1 || (d.reserved1 & 4) || d.act != SPAA_ACT_RELU || d.add != nullptr || d.gate != nullptr || d.gate2 != nullptr ||
            d.gate_bits != nullptr || d.gate2_bits != nullptr || d.aux_out != nullptr || (d.Hout & 1) || (d.Wout & 1) || (d.Cout & 3) ||
            (d.out_cstride & 3) || (d.out_coff & 3) || d.out_coff + d.Cout > d.out_cstride ||
            (int64_t)d.B * (d.Hout / 2) * (d.Wout / 2) * d.out_cstride * 4 >= (int64_t)1 << 31)
            return hipErrorInvalidValue;
    }
    if ((int64_t)((d.Cout * nfold + 127) & ~127) * ((d.cls[0].K + 63) & ~63) * 2 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    if ((int64_t)d.B * d.Hin * d.Win * d.in_cstride * 2 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    const h16p_cv_t nogeo = {};
    static bool attr_set[6][SPAA_MAX_DEVICES] = {};
    if (d.reserved1 & 4) {
        // canvas / K-range form (small images, few workgroups with long K): the plan of spaa_tapconv_h16p_plan, recomputed here with
        // the caller's K ranges (`ksplit`, with its workspace) and N tile
        if (!h16p_cv_shape_ok(d)) return hipErrorInvalidValue;
        if (d.tap_range[0] < -1 || d.tap_range[1] > 1 || d.tap_range[2] < -1 || d.tap_range[3] > 1) return hipErrorInvalidValue;
        if (d.ksplit > 1 && d.splitk_ws == nullptr) return hipErrorInvalidValue;
        const int fb = d.reserved1 & 3;
        const h16p_plan_t pl = h16p_make_plan(d, h16p_ncu(), fb == 1 ? 64 : fb == 2 ? 128 : 0, d.ksplit > 1 ? d.ksplit : 1, false);
        if (pl.nwg > 0x7fffffff || pl.bn == 0 || pl.ksplit < 1 || pl.n_tiles < 1 || pl.ksplit != (d.ksplit > 1 ? d.ksplit : 1)) return hipErrorInvalidValue;
        h16p_cv_t geo = {};
        geo.ksplit = pl.ksplit, geo.kb_per = pl.kb_per;
        geo.gy = pl.gy, geo.gx = pl.gx, geo.py = pl.canvas ? pl.py : (1 << 14), geo.px = pl.canvas ? pl.px : (1 << 14);
        // v / p == (v * m) >> 20 with m = ceil(2^20 / p) for v * p < 2^20 (v <= 4095, p <= 255); image-aligned regions (period 2^14 >
        // every coordinate): m = 64 gives 0
        geo.my = ((1u << 20) + geo.py - 1) / geo.py, geo.mx = ((1u << 20) + geo.px - 1) / geo.px;
        geo.nsp = (int)(pl.nwg / ((int64_t)pl.n_tiles * pl.ksplit));
        // workgroup order by what an XCD's L2 should keep: the fp16 weights or the activations
        geo.order = (int64_t)((d.Cout + 127) & ~127) * ((d.cls[0].K + 63) & ~63) * 2 > (int64_t)d.B * d.Hin * d.Win * d.Cin * 2;
        // `reserved1` bit 8: the workspace begins with SPAA_SPLITK_HDR_FLOATS zeroed counter slots -> the K ranges meet inside the kernel
        // (and the partial sums leave through the branch-free epilogue's agent-scope stores: 4-channel quads, 32-bit offsets)
        geo.fix = pl.ksplit > 1 && (d.reserved1 & 256) && (int64_t)geo.nsp * pl.n_tiles <= SPAA_SPLITK_HDR_FLOATS && !(d.Cout & 3) &&
                  (int64_t)pl.ksplit * d.B * d.Hout * d.Wout * ((d.Cout + 127) & ~127) * 4 < ((int64_t)1 << 31);
        spaa_tapconv_t dd = d;
        dd.ksplit = pl.ksplit;
#define H16P_LAUNCH_CV(N, SLOT)                                                                                            \
    {                                                                                                                      \
        typedef h16p_geo<1> G;                                                                                             \
        const size_t wbytes = 3 * (size_t)(((3 * N / 16 + 7) / 8) * 8 * 1024);                                             \
        const size_t mainb = G::NBUF * (size_t)G::PATCH_BYTES + wbytes, epib = 8 * 32 * (size_t)(N * 4 + 16);              \
        const size_t smem = mainb > epib ? mainb : epib;                                                                   \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&h16p_kernel<N, 1, true>), (int)smem, attr_set[SLOT]); \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((h16p_kernel<N, 1, true>), dim3((unsigned)pl.nwg), dim3(512), smem, stream, dd, pl.wg_y, pl.wg_x, pl.n_tiles, geo); \
    }
        if (pl.bn == 64) H16P_LAUNCH_CV(64, 4) else H16P_LAUNCH_CV(128, 5)
#undef H16P_LAUNCH_CV
        if (pl.ksplit > 1 && !geo.fix) {
            const int npad = (d.Cout + 127) & ~127;
            const int64_t M = (int64_t)d.B * d.Hout * d.Wout, nthr = M * ((d.Cout + 3) >> 2);
            if (d.io_dtype & SPAA_IO_OUT_F16)
                hipLaunchKernelGGL(h16p_splitk_reduce_kernel<_Float16>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, dd, (int)M, npad);
            else
                hipLaunchKernelGGL(h16p_splitk_reduce_kernel<float>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, dd, (int)M, npad);
        }
        return (int)hipGetLastError();
    }
    if (d.ksplit > 1) return hipErrorInvalidValue;
    const int OH = S == 1 ? h16p_geo<1>::OH : h16p_geo<2>::OH;
    const int wg_y = (d.Hm + OH - 1) / OH, wg_x = (d.Wm + OW - 1) / OW;
    // (`reserved1` bit 5: the 64-wide two-workgroups-per-CU form for wider layers too -- short K, where prologue and epilogue outweigh the
    // products: chosen per layer shape by spaa_amd/convplan.py)
    const int BN = (d.Cout * nfold <= 64 || (S == 1 && (d.reserved1 & 32) && !(d.reserved1 & 16))) ? 64 : 128;
    const int n_tiles = (d.Cout * nfold + BN - 1) / BN;
    const int64_t nwg = (int64_t)d.B * wg_y * wg_x * n_tiles;
    if (nwg > 0x7fffffff) return hipErrorInvalidValue;
#define H16P_LAUNCH(N, SS, SLOT)                                                                                           \
    {                                                                                                                      \
        typedef h16p_geo<SS> G;                                                                                            \
        const size_t wbytes = 3 * (size_t)(((3 * N / 16 + 7) / 8) * 8 * 1024);                                             \
        const size_t mainb = G::NBUF * (size_t)G::PATCH_BYTES + wbytes, epib = 8 * 32 * (size_t)(N * 4 + 16);              \
        const size_t smem = mainb > epib ? mainb : epib;                                                                   \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&h16p_kernel<N, SS>), (int)smem, attr_set[SLOT]);  \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((h16p_kernel<N, SS>), dim3((unsigned)nwg), dim3(512), smem, stream, d, wg_y, wg_x, n_tiles, nogeo); \
    }
    const bool unp = (d.reserved1 & 128) != 0;
    if (unp) {
        // pool-adjoint prologue: `in` = the pooled gradient, `in2` = the pool's arg-max bytes; the two-workgroup form only
        if (S != 1 || BN != 64 || (d.reserved1 & 16) || nfold != 1 || d.in2 == nullptr || (d.Hin & 1) || (d.Win & 1) || (d.in2_cstride & 7) || (d.in2_coff & 7) ||
            d.in2_coff + d.Cin > d.in2_cstride || d.tap_range[0] < -1 || d.tap_range[1] > 1 || d.tap_range[2] < -1 || d.tap_range[3] > 1)
            return hipErrorInvalidValue;
    }
    if (S == 1 && BN == 64 && !(d.reserved1 & 16)) {
        // two workgroups per compute unit (LEAN): one 40 KiB patch buffer + three 12 KiB weight stages; epilogue 8 x 32 rows of 272 bytes
        typedef h16p_geo<1> G;
        const size_t mainb = (size_t)G::PATCH_BYTES + 3 * (size_t)(12 * 1024), epib = 8 * 32 * (size_t)(64 * 4 + 16);
        const size_t smem = mainb > epib ? mainb : epib;
        static bool lean_set[2][SPAA_MAX_DEVICES] = {};
        if (unp) {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&h16p_kernel<64, 1, false, true, true>), (int)smem, lean_set[1]);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((h16p_kernel<64, 1, false, true, true>), dim3((unsigned)nwg), dim3(512), smem, stream, d, wg_y, wg_x, n_tiles, nogeo);
        } else {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&h16p_kernel<64, 1, false, true>), (int)smem, lean_set[0]);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((h16p_kernel<64, 1, false, true>), dim3((unsigned)nwg), dim3(512), smem, stream, d, wg_y, wg_x, n_tiles, nogeo);
        }
    } else if (S == 1) {
        if (BN == 64) H16P_LAUNCH(64, 1, 0) else H16P_LAUNCH(128, 1, 1)
    } else {
        if (BN == 64) H16P_LAUNCH(64, 2, 2) else H16P_LAUNCH(128, 2, 3)
    }
#undef H16P_LAUNCH
    return (int)hipGetLastError();
}
