// tapconv_x6d.hip — bf16x6 tap-list convolution (see tapconv_x6.hip for the arithmetic) with LDS-DMA staging.
//
// The v2/v3 kernels move every operand global -> VGPR -> (split) -> ds_write -> LDS; PMC shows their matrix cores busy
// 49 % of the time with the two co-resident workgroups convoying through load / split / LDS-store phases.  Here NO
// operand passes through registers on its way to LDS:
//   * activations: `buffer_load_dwordx4 ... lds` gathers the im2col rows as fp32 (128-B rows, 16-B chunks XOR-swizzled
//     on the SOURCE side: the LDS image of an LDS-DMA is lane-linear); out-of-image taps use the out-of-range offset
//     0x80000000, for which the DMA writes zeros (checked on gfx950: tools/micro/dma_oob.hip);
//   * weights: the host's pre-split bf16 planes are DMA'd as they are (64-B rows, swizzled the same way);
//   * each wave owns 32 pixels x BN output channels: it reads its pixels' 8 fp32 per lane (2 x ds_read_b128), splits
//     them into the three bf16 fragments in registers (no redundancy between waves) and reads the weight fragments
//     (ds_read_b128, conflict-free); 6 MFMAs per (32 channels x 16 k).
// Two LDS stages; one __syncthreads() per 32-deep K-step; the DMA of step t+1 is issued right after the barrier of
// step t and lands during its MFMAs.  Needs Cin % 32 == 0 (a K-step lies inside one tap: the tap is wave-uniform).
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ unsigned int cvt2(float a, float b) {
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned int p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned int p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// One LDS-DMA piece: 64 lanes x 16 bytes, global (buffer, per-lane byte offset `voff` + uniform `soff`) -> LDS at the
// wave-uniform address `dst` + 16 * lane.  An out-of-range offset writes zeros.  (A __device__ helper: the builtin has no
// host-side meaning and would silently drop the kernel's host stub if it sat in the kernel template itself.)
__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
}

// 8 fp32 -> three bf16x8 with x == h + m + l exactly
__device__ __forceinline__ void split8(const f4 x0, const f4 x1, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
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

// LDS chunk swizzles.  A b128 LDS read is served in groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...); the
// 16 lanes of a group must hit 16 different 16-byte bank groups.  With 32x32x16 fragments (SH = 32) a group holds 16
// consecutive rows at ONE k-chunk; with 16x16x32 fragments (SH = 16) it holds rows {0-3,12-15} at chunk c and rows
// {4-11} at chunk c+1 (weights) / c+2 (pixels), which needs the second form below.
template <int SH>
__device__ __forceinline__ int swz_pix(int r) {  // 128-byte rows, 8 chunks
    if (SH == 32) return (r >> 1) & 7;
    const int v = (r >> 1) & 1, u = (r >> 2) & 3;
    return (((u ^ (u >> 1)) & 1)) | (v << 1) | ((u >> 1) << 2);
}
template <int SH>
__device__ __forceinline__ int swz_w(int n) {  // 64-byte rows, 4 chunks
    return SH == 32 ? (n >> 2) & 3 : ((n >> 3) & 1) << 1;
}

template <int NW, int BN, int SH, bool CO, int NA>
__global__ __launch_bounds__(64 * NW, 2) void tapconv_x6d_kernel(const spaa_tapconv_t p, const int m_tiles,
                                                                 const int n_tiles) {
    constexpr int BM = 32 * NW;
    constexpr int TN = BN / 32;
    constexpr int A_BYTES = BM * 128;               // fp32 pixels: [BM][32] floats
    constexpr int W_PLANE = BN * 64;                // one bf16 plane: [BN][32] bf16
    constexpr int WS_BYTES = 3 * W_PLANE;           // one weight stage (h / m / l planes)
    constexpr int W_BASE = NA * A_BYTES;            // LDS: NA pixel stages, then 2 weight stages
    constexpr int LDS_BYTES = NA * A_BYTES + 2 * WS_BYTES;
    constexpr int W_PIECES = 3 * BN / 16;           // 1-KiB pieces (16 rows of one plane)
    constexpr int WPW = (W_PIECES + NW - 1) / NW;   // weight pieces per wave per K-step (piece q -> wave q % NW)

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const spaa_tapclass_t cl = p.cls[blockIdx.y];

    // Work distribution.  The grid's x dimension may be SMALLER than the number of tiles (persistent launch): workgroup g
    // then walks the tiles orig = (g & 7) + 8 j, j = g >> 3, (g >> 3) + G8, ... — the share of its own XCD (blockIdx.x % 8),
    // in the XCD-aware order below — and overlaps the epilogue of a tile with the first gathers of the next one.  With
    // gridDim.x == number of tiles every workgroup has exactly one tile (the plain launch).
    const int nwg = m_tiles * n_tiles;
    const int xcd_ = blockIdx.x & 7;
    const int q_x = (nwg - xcd_ + 7) >> 3;              // tiles whose index is congruent to this XCD
    const int G8 = ((int)gridDim.x - xcd_ + 7) >> 3;    // workgroups on this XCD
    int jt = blockIdx.x >> 3;
    int n_blk = 0, m_blk = 0;
#define X6D_TILE_COORDS()                                                                                          \
    {                                                                                                              \
        const int q = nwg >> 3, r = nwg & 7;                                                                       \
        const int tile = (xcd_ < r ? xcd_ * (q + 1) : r * (q + 1) + (xcd_ - r) * q) + jt;                          \
        n_blk = (tile % n_tiles) * BN;                                                                             \
        m_blk = (tile / n_tiles) * BM;                                                                             \
    }

    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;
    const int row_bytes = p.in_cstride * 4;
    const int* __restrict__ taps = p.taps + 2 * cl.tap_off;

    // ---- activation staging: this wave's 32 pixels = 4 pieces of 8 rows; lane -> (row, physical chunk)
    int a_off[4];
    uint32_t a_mlo[4], a_mhi[4];
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)row_bytes;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    // ---- weight staging: piece q = wave * WPW + i -> (plane, 16-row block); lane -> (row, physical chunk)
    const int npad = (p.Cout * (p.nfold > 1 ? p.nfold : 1) + 127) & ~127;
    const int plane_bytes = npad * cl.Kpad * 2;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_split) + (uint64_t)cl.w_off * 6u;
    const uint32_t w_lo = __builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = __builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane(3u * (uint32_t)plane_bytes),
                                                           0x00020000);
    int w_goff[WPW];

    const int nk_all = cl.Kpad / BK;
    const int Cin = p.Cin;
    // split-K: this workgroup multiplies K-steps [ks_begin, ks_end) only
    const int nsplit = p.ksplit > 1 ? p.ksplit : 1;
    int ks_begin = (int)((int64_t)blockIdx.z * nk_all / nsplit);
    int ks_end = (int)((int64_t)(blockIdx.z + 1) * nk_all / nsplit);
    int nk = ks_end - ks_begin;
    // stream-K (p.ksplit == -1, persistent launch): the K-steps of ALL tiles of this XCD form one sequence of
    // q_x * nk_all iterations that is cut into G8 equal ranges, one per workgroup: whole tiles are finished as usual; a
    // tile cut by a range boundary leaves raw partial sums in the workspace (two slots per workgroup: slot 0 for a
    // segment that starts inside a tile, slot 1 for one that starts a tile but does not finish it) and
    // streamk_reduce_kernel adds the segments in order and applies the epilogue.  Removes the last-round quantisation
    // of layers whose tile count is not a multiple of the resident workgroups.
    const bool streamk = p.ksplit == -1;
    const int64_t total_x = (int64_t)q_x * nk_all;
    int64_t it = 0, it1 = 0;
    if (streamk) {
        it = (int64_t)(blockIdx.x >> 3) * total_x / G8;
        it1 = (int64_t)((blockIdx.x >> 3) + 1) * total_x / G8;
        if (it >= it1) return;  // (more workgroups than iterations: nothing to do; uniform for the workgroup)
        jt = (int)(it / nk_all);
        ks_begin = (int)(it - (int64_t)jt * nk_all);
        ks_end = (int)((int64_t)nk_all < ks_begin + (it1 - it) ? (int64_t)nk_all : ks_begin + (it1 - it));
        nk = ks_end - ks_begin;
    }
    // wave-uniform state of the K-step being STAGED (one ahead of the one being multiplied); the tap offsets come through
    // scalar loads issued one K-step before they are used
    typedef const __attribute__((address_space(4))) int* cint_ptr;
    cint_ptr ctaps = (cint_ptr)(uintptr_t)taps;
    // K-step order: with several taps and several 32-channel chunks the steps run CHUNK-major (all taps of a chunk, then
    // the next chunk): the 128-byte slice of a pixel is then re-read by the 9 taps within 9 consecutive steps and stays
    // in the XCD's L2 (tap-major order re-reads every slice 12+ steps later, when 64 workgroups' traffic has evicted it).
    // The sum over K is the same set of products either way.
    const bool chunk_major = (cl.ntaps > 1) && (Cin > BK) && !(p.reserved0 & 1);  // (reserved0 bit 0: tap-major, A/B runs)
#define X6D_ADVANCE(tap, kc)                                                                                       \
    if (chunk_major) {                                                                                             \
        tap += 1;                                                                                                  \
        if (tap >= cl.ntaps) {                                                                                     \
            tap = 0;                                                                                               \
            kc += BK;                                                                                              \
        }                                                                                                          \
    } else {                                                                                                       \
        kc += BK;                                                                                                  \
        if (kc >= Cin) {                                                                                           \
            kc = 0;                                                                                                \
            tap += 1;                                                                                              \
        }                                                                                                          \
    }
    int s_tap = 0, s_kc = 0;  // (tap, channel offset) of the step whose PIXELS are staged next
    int w_tap = 0, w_kc = 0;  // ... and of the step whose WEIGHTS are staged next
    int n_dy = 0, n_dx = 0;
    int voff[4];

#define X6D_PREP(more, ks_next)                                                                                    \
    {                                                                                                              \
        const int tapoff = (n_dy * p.Win + n_dx) * row_bytes + s_kc * 4;                                           \
        const uint32_t bit = (more) ? 1u << (s_tap & 31) : 0u;                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                            \
            const uint32_t mw = s_tap < 32 ? a_mlo[j] : a_mhi[j];                                                  \
            voff[j] = (mw & bit) ? a_off[j] + tapoff : (int)0x80000000;                                            \
        }                                                                                                          \
        X6D_ADVANCE(s_tap, s_kc)                                                                                   \
        const int tn = min(s_tap, cl.ntaps - 1);                                                                   \
        n_dy = ctaps[2 * tn];                                                                                      \
        n_dx = ctaps[2 * tn + 1];                                                                                  \
    }
#define X6D_DMA_A(abase, j) dma16(rsrc_in, (abase) + (4 * wave + (j)) * 1024, voff[j], 0);
#define X6D_DMA_W(wbase, i, soff)                                                                                  \
    if (W_PIECES % NW == 0 || wave + NW * (i) < W_PIECES)                                                          \
        dma16(rsrc_w, (wbase) + (wave + NW * (i)) * 1024, w_goff[i], (soff));
#define X6D_LDW(dst, ptr)                                                                                          \
    {                                                                                                              \
        const unsigned char* wb_ = (ptr);                                                                          \
        dst[0] = *reinterpret_cast<const bf16x8*>(wb_);                                                            \
        dst[1] = *reinterpret_cast<const bf16x8*>(wb_ + W_PLANE);                                                  \
        dst[2] = *reinterpret_cast<const bf16x8*>(wb_ + 2 * W_PLANE);                                              \
    }
// weights = A operand (rows = output channels), pixels = B operand (columns); small terms first
#define X6D_MFMA6(MF, accv, wf, pf)                                                                                \
    accv = MF(wf[2], pf[0], accv, 0, 0, 0);                                                                        \
    accv = MF(wf[0], pf[2], accv, 0, 0, 0);                                                                        \
    accv = MF(wf[1], pf[1], accv, 0, 0, 0);                                                                        \
    accv = MF(wf[1], pf[0], accv, 0, 0, 0);                                                                        \
    accv = MF(wf[0], pf[1], accv, 0, 0, 0);                                                                        \
    accv = MF(wf[0], pf[0], accv, 0, 0, 0);

    // DMA schedule.  A wave reads back only ITS OWN pixel rows, so the pixel fragments of step t+1 are read and split at
    // the end of step t, behind a counted vmcnt that covers the pixel pieces (no barrier needed for own DMA data); the
    // barrier at the top of a step then only orders the weight planes.  The pieces of a step are spread over its MFMA
    // blocks (piece s of NP goes to block s * NBLK / NP):
    //   NA == 2: step t issues pixels(t+1) then weights(t+1); waits: top vmcnt(0), end vmcnt(#weights).
    //   NA == 3: pixels run TWO steps ahead (narrow tiles: a step's MFMAs are shorter than the gather latency):
    //            step t issues weights(t+1) then pixels(t+2); waits: top vmcnt(4) (leaves pixels(t+1) in flight), end
    //            vmcnt(#weights + 4) (leaves weights(t+1), pixels(t+2)).
    constexpr int NP = 4 + WPW;
    constexpr int W_MIN = W_PIECES / NW;  // weight pieces every wave issues
#define X6D_ISSUE(blk, NBLK, an, wn, soff)                                                                         \
    {                                                                                                              \
        _Pragma("unroll") for (int s_ = 0; s_ < NP; ++s_)                                                          \
            if (s_ * (NBLK) / NP == (blk)) {                                                                       \
                if (NA == 2) {                                                                                     \
                    if (s_ < 4) { X6D_DMA_A(an, s_) } else { X6D_DMA_W(wn, s_ - 4, soff) }                         \
                } else {                                                                                           \
                    if (s_ < WPW) { X6D_DMA_W(wn, s_, soff) } else { X6D_DMA_A(an, s_ - WPW) }                     \
                }                                                                                                  \
            }                                                                                                      \
    }

    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));

    // fragment read addresses (bytes inside a stage).  Pixels: a lane holds 2 x 8 fp32 of its pixel(s) per K-step.
    int p_addr[2][2];
    int w_addr_l[2];
    if constexpr (SH == 32) {
        // 32x32x16: pixel (lane & 31); half-step kk covers k-chunks 4 kk + 2 (lane >> 5) + {0, 1}
        const int prow = 32 * wave + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                p_addr[kk][h] = prow * 128 + (((kk * 4 + (lane >> 5) * 2 + h) ^ swz_pix<SH>(prow)) * 16);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int n = lane & 31;
            w_addr_l[kk] = n * 64 + (((2 * kk + (lane >> 5)) ^ swz_w<SH>(n)) * 16);
        }
    } else {
        // 16x16x32: pixel (lane & 15) of each of the wave's two 16-pixel blocks; k-chunks 2 (lane >> 4) + {0, 1}
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int r = 32 * wave + 16 * ib + (lane & 15);
#pragma unroll
            for (int h = 0; h < 2; ++h) p_addr[ib][h] = r * 128 + ((((lane >> 4) * 2 + h) ^ swz_pix<SH>(r)) * 16);
        }
        w_addr_l[0] = (lane & 15) * 64 + (((lane >> 4) ^ swz_w<SH>(lane & 15)) * 16);
        w_addr_l[1] = 0;
    }
#define X6D_LDP(pf, sbase)                                                                                         \
    {                                                                                                              \
        const f4 x0_ = *reinterpret_cast<const f4*>((sbase) + p_addr[0][0]);                                       \
        const f4 x1_ = *reinterpret_cast<const f4*>((sbase) + p_addr[0][1]);                                       \
        const f4 y0_ = *reinterpret_cast<const f4*>((sbase) + p_addr[1][0]);                                       \
        const f4 y1_ = *reinterpret_cast<const f4*>((sbase) + p_addr[1][1]);                                       \
        split8(x0_, x1_, pf[0][0], pf[0][1], pf[0][2]);                                                            \
        split8(y0_, y1_, pf[1][0], pf[1][1], pf[1][2]);                                                            \
    }

    constexpr int TJ = BN / 16;
    constexpr int NACC = SH == 32 ? TN : 1;
    f32x16 acc32[NACC];
    f32x4 acc16[2][SH == 16 ? TJ : 1];

    // The tap offsets are needed by every tile setup (validity masks): fetch the first NTREG of them once, back to back
    // (a dependent scalar load per tap and pixel piece costs ~4 us per tile otherwise).
    constexpr int NTREG = 9;
    int tdy[NTREG], tdx[NTREG];
#pragma unroll
    for (int t = 0; t < NTREG; ++t) {
        const int tt = min(t, cl.ntaps - 1);
        tdy[t] = ctaps[2 * tt];
        tdx[t] = ctaps[2 * tt + 1];
    }
    // Everything that depends on the tile: gather offsets and tap-validity masks of the lane's pixel rows, weight row
    // offsets, the K-step state, and the DMAs of the tile's first K-step (and second pixel step when NA == 3).
#define X6D_TILE_SETUP()                                                                                           \
    {                                                                                                              \
        X6D_TILE_COORDS()                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                            \
            const int r = 32 * wave + 8 * j + (lane >> 3);                                                         \
            const int c = (lane & 7) ^ swz_pix<SH>(r); /* logical 16-byte chunk held at this lane's LDS slot */    \
            const int m = m_blk + r;                                                                               \
            const bool ok = m < M;                                                                                 \
            const int mm = ok ? m : 0;                                                                             \
            const int b = mm / HWm;                                                                                \
            const int rr = mm - b * HWm;                                                                           \
            const int y = rr / p.Wm;                                                                               \
            const int x = rr - y * p.Wm;                                                                           \
            const int iy0 = y * p.s_in, ix0 = x * p.s_in;                                                          \
            a_off[j] = ((b * p.Hin + iy0) * p.Win + ix0) * row_bytes + p.in_coff * 4 + c * 16;                     \
            uint32_t lo = 0, hi = 0;                                                                               \
            _Pragma("unroll") for (int t = 0; t < NTREG; ++t) { /* tap offsets preloaded in registers */           \
                const bool v = t < cl.ntaps && ok && (unsigned)(iy0 + tdy[t]) < (unsigned)p.Hin &&                 \
                               (unsigned)(ix0 + tdx[t]) < (unsigned)p.Win;                                         \
                lo |= (v ? 1u : 0u) << t;                                                                          \
            }                                                                                                      \
            for (int t = NTREG; t < cl.ntaps; ++t) { /* (more taps than registers hold: 5x5 kernels) */            \
                const int dy = ctaps[2 * t], dx = ctaps[2 * t + 1];                                                \
                const bool v = ok && (unsigned)(iy0 + dy) < (unsigned)p.Hin && (unsigned)(ix0 + dx) < (unsigned)p.Win; \
                if (t < 32) lo |= (v ? 1u : 0u) << t;                                                              \
                else hi |= (v ? 1u : 0u) << (t - 32);                                                              \
            }                                                                                                      \
            a_mlo[j] = lo;                                                                                         \
            a_mhi[j] = hi;                                                                                         \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < WPW; ++i) {                                                          \
            const int q = wave + NW * i; /* (q >= W_PIECES: no such piece, never issued) */                        \
            const int pl = q / (BN / 16), rb = q % (BN / 16);                                                      \
            const int n = 16 * rb + (lane >> 2);                                                                   \
            const int c = (lane & 3) ^ swz_w<SH>(n);                                                               \
            w_goff[i] = pl * plane_bytes + (n_blk + n) * cl.Kpad * 2 + c * 16;                                     \
        }                                                                                                          \
        s_tap = chunk_major ? ks_begin % cl.ntaps : (ks_begin * BK) / Cin;                                         \
        s_kc = chunk_major ? (ks_begin / cl.ntaps) * BK : (ks_begin * BK) % Cin;                                   \
        w_tap = s_tap;                                                                                             \
        w_kc = s_kc;                                                                                               \
        n_dy = ctaps[2 * min(s_tap, cl.ntaps - 1)];                                                                \
        n_dx = ctaps[2 * min(s_tap, cl.ntaps - 1) + 1];                                                            \
        if (nk > 0) {                                                                                              \
            X6D_PREP(true, 0)                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) X6D_DMA_A(smem, j)                                       \
            _Pragma("unroll") for (int i = 0; i < WPW; ++i) X6D_DMA_W(smem + W_BASE, i, (w_tap * Cin + w_kc) * 2)  \
            if constexpr (NA == 3) {                                                                               \
                X6D_PREP(nk > 1, 1)                                                                                \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) X6D_DMA_A(smem + A_BYTES, j)                         \
            }                                                                                                      \
        }                                                                                                          \
    }

    bf16x8 pfc[2][3];  // pixel fragments of the current K-step (h / m / l planes of the two halves or pixel blocks)
    bf16x8 pfd[2][3];
    int ia = 0;        // pixel stage of the current step (stages rotate 0 .. NA-1)
    X6D_TILE_SETUP()

  for (;;) {  // one tile per pass (persistent launch: several)
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[j][r] = 0.f;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int j = 0; j < (SH == 16 ? TJ : 1); ++j) acc16[ib][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    ia = 0;
    if (nk > 0) {
        if constexpr (NA == 3) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_MIN + 4) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_MIN) : "memory");
        }
        X6D_LDP(pfc, smem)
    }

    constexpr int NBLK = SH == 32 ? 2 * TN : TJ;  // MFMA blocks per K-step (6 resp. 12 MFMAs each)
    // One K-step: multiplies with the pixel fragments PFC and leaves those of the next step in PFN (the loop below is
    // unrolled by two with the roles swapped, so the fragments never have to be copied).
#define X6D_STEP(PFC, PFN)                                                                                         \
    {                                                                                                              \
        /* own weight DMAs of step ks have landed (vmcnt) and everybody's have (barrier); every wave is also past */  \
        /* its reads of the other weight stage, which is overwritten during this step */                          \
        if constexpr (NA == 3) {                                                                                   \
            /* raw barrier: __syncthreads() would drain vmcnt to 0 and with it the pixel gather that is meant to */ \
            /* stay in flight across this barrier; own LDS reads are retired by lgkmcnt(0) */                      \
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");                                            \
            __builtin_amdgcn_s_barrier();                                                                          \
            asm volatile("" ::: "memory");                                                                         \
        } else {                                                                                                   \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                       \
            __syncthreads();                                                                                       \
        }                                                                                                          \
        const int cur = (ks - ks_begin) & 1;                                                                       \
        const unsigned char* wc = smem + W_BASE + cur * WS_BYTES;                                                  \
        unsigned char* wn = smem + W_BASE + (cur ^ 1) * WS_BYTES;                                                  \
        const int ia1 = ia + 1 == NA ? 0 : ia + 1;         /* pixel stage of step ks+1 (read at the end) */        \
        const int ia2 = NA == 3 ? (ia1 + 1 == NA ? 0 : ia1 + 1) : ia1; /* pixel stage filled during this step */   \
        unsigned char* an = smem + ia2 * A_BYTES;                                                                  \
        const bool more = ks + (NA - 1) < ks_end;          /* is there a step whose pixels are gathered now? */    \
        if (ks + 1 < ks_end) { X6D_ADVANCE(w_tap, w_kc) }  /* weights of step ks+1 (the last step is re-staged) */ \
        const int soff = (w_tap * Cin + w_kc) * 2;                                                                 \
        bf16x8 wf[2][3];                                                                                           \
        X6D_LDW(wf[0], wc + w_addr_l[0])                                                                           \
        X6D_PREP(more, ks + NA - 1)                                                                                \
        _Pragma("unroll") for (int b = 0; b < NBLK; ++b) {                                                         \
            if (b + 1 < NBLK) {                                                                                    \
                if constexpr (SH == 32) {                                                                          \
                    X6D_LDW(wf[(b + 1) & 1], wc + w_addr_l[(b + 1) / TN] + ((b + 1) % TN) * 2048)                  \
                } else {                                                                                           \
                    X6D_LDW(wf[(b + 1) & 1], wc + w_addr_l[0] + (b + 1) * 1024)                                    \
                }                                                                                                  \
            }                                                                                                      \
            X6D_ISSUE(b, NBLK, an, wn, soff)                                                                       \
            if (b == NBLK - 1) {                                                                                   \
                /* the pixel pieces of step ks+1 (this wave's own rows) have landed: fetch and split them now, */  \
                /* under the last block's MFMAs */                                                                 \
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA == 3 ? W_MIN + 4 : W_MIN) : "memory");                 \
                X6D_LDP(PFN, smem + ia1 * A_BYTES)                                                                 \
            }                                                                                                      \
            if constexpr (SH == 32) {                                                                              \
                X6D_MFMA6(__builtin_amdgcn_mfma_f32_32x32x16_bf16, acc32[b % TN], wf[b & 1], PFC[b / TN])          \
            } else {                                                                                               \
                X6D_MFMA6(__builtin_amdgcn_mfma_f32_16x16x32_bf16, acc16[0][b], wf[b & 1], PFC[0])                 \
                X6D_MFMA6(__builtin_amdgcn_mfma_f32_16x16x32_bf16, acc16[1][b], wf[b & 1], PFC[1])                 \
            }                                                                                                      \
        }                                                                                                          \
        ia = ia1;                                                                                                  \
    }
    int ks = ks_begin;
    for (; ks + 1 < ks_end; ks += 2) {
        X6D_STEP(pfc, pfd)
        ++ks;
        X6D_STEP(pfd, pfc)
        --ks;
    }
    if (ks < ks_end) X6D_STEP(pfc, pfd)
#undef X6D_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last step's (all-zero) prefetch must not outlive the wave
    // the tile is finished; in a persistent launch the next tile's first gathers are issued BEFORE this tile's
    // epilogue (the stage buffers are free once every wave is past its last reads), so the stores below overlap them
    const int m_blk_e = m_blk, n_blk_e = n_blk;
    const bool partial_e = streamk && !(ks_begin == 0 && ks_end == nk_all);
    const int slot_e = ks_begin > 0 ? 0 : 1;
    bool has_next;
    if (streamk) {
        it += nk;
        has_next = it < it1;
    } else {
        has_next = jt + G8 < q_x;
    }
#define X6D_NEXT_TILE()                                                                                            \
    {                                                                                                              \
        __syncthreads();                                                                                           \
        if (streamk) {                                                                                             \
            jt = (int)(it / nk_all);                                                                               \
            ks_begin = (int)(it - (int64_t)jt * nk_all);                                                           \
            ks_end = (int)((int64_t)nk_all < ks_begin + (it1 - it) ? (int64_t)nk_all : ks_begin + (it1 - it));     \
            nk = ks_end - ks_begin;                                                                                \
        } else {                                                                                                   \
            jt += G8;                                                                                              \
        }                                                                                                          \
        X6D_TILE_SETUP()                                                                                           \
    }
    if constexpr (!CO) {
        if (has_next) X6D_NEXT_TILE()
    }
    do {
        if (partial_e) {
            // stream-K segment: raw partial sums, tile-local [BM][BN], into this workgroup's slot
            float* ws = p.splitk_ws + ((size_t)blockIdx.x * 2 + slot_e) * (BM * BN);
            if constexpr (SH == 32) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f4*>(ws + (32 * wave + (lane & 31)) * BN + 32 * j + 8 * g + 4 * (lane >> 5)) =
                            f4{acc32[j][4 * g], acc32[j][4 * g + 1], acc32[j][4 * g + 2], acc32[j][4 * g + 3]};
            } else {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        *reinterpret_cast<f32x4*>(ws + (32 * wave + 16 * ib + (lane & 15)) * BN + 16 * j + 4 * (lane >> 4)) =
                            acc16[ib][j];
            }
            break;
        }

    if (p.ksplit > 1) {
        // split-K: raw partial sums to the workspace [split][M][Npad]; splitk_reduce_kernel finishes the layer
        float* ws = p.splitk_ws + (size_t)blockIdx.z * M * npad;
        if constexpr (SH == 32) {
            const int m = m_blk_e + 32 * wave + (lane & 31);
            if (m < M) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f4*>(ws + (size_t)m * npad + n_blk_e + 32 * j + 8 * g + 4 * (lane >> 5)) =
                            f4{acc32[j][4 * g], acc32[j][4 * g + 1], acc32[j][4 * g + 2], acc32[j][4 * g + 3]};
            }
        } else {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const int m = m_blk_e + 32 * wave + 16 * ib + (lane & 15);
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    *reinterpret_cast<f4*>(ws + (size_t)m * npad + n_blk_e + 16 * j + 4 * (lane >> 4)) = acc16[ib][j];
            }
        }
        break;
    }
    if constexpr (CO) {
        // ---- coalesced epilogue (memory-heavy layers): the result tile goes through LDS (the stage buffers are free
        // now) so that a store instruction writes whole channel rows of a few pixels, and the residual / gate loads read
        // whole rows, instead of 16-byte pieces of 16 different rows.
        static_assert(SH == 16, "coalesced epilogue: 16x16x32 variant only");
        constexpr int PITCH = BN + 4;  // floats: a 16-B bank group apart per row -> conflict-free b128 writes
        static_assert(BM * PITCH * 4 <= LDS_BYTES, "result tile must fit in the stage buffers");
        if (vec) {
            __syncthreads();  // every wave is past its last reads of the stages
            float* slab = reinterpret_cast<float*>(smem) + (32 * wave) * PITCH;  // this wave's 32 rows
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    *reinterpret_cast<f32x4*>(slab + (16 * ib + (lane & 15)) * PITCH + 16 * j + 4 * (lane >> 4)) = acc16[ib][j];
            constexpr int LPR = BN / 4;    // lanes per row (4 channels each)
            constexpr int RPI = 64 / LPR;  // rows per store instruction
            const int n0 = n_blk_e + 4 * (lane % LPR);
            const int cfold = p.nfold > 1 ? n0 / p.Cout : 0;
            const bool linear = p.nfold <= 1 && (p.s_out == 1) && (cl.oy0 == 0) && (cl.ox0 == 0) && (p.Hm == p.Hout) &&
                                (p.Wm == p.Wout);
            int m = m_blk_e + 32 * wave + lane / LPR;
            int pb = m / HWm, py = (m - pb * HWm) / p.Wm, px = m - pb * HWm - py * p.Wm;
            if (fast_epi_ok(p, vec)) {   // branch-free operand accesses, several rows in flight per lane (epilogue.hpp: fast_epi_*)
                const int n_e = n0 - cfold * p.Cout;
                const bool n_ok = cfold < (p.nfold > 1 ? p.nfold : 1) && n_e < p.Cout;
                const fast_epi_t fe = make_fast_epi(p, n_ok ? n_e : 0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                constexpr int NP = 32 / RPI, EB = NP < 8 ? NP : 8;
#pragma unroll 1
                for (int i0 = 0; i0 < NP; i0 += EB) {
                    fast_pre_t<float> pre[EB];
                    int oo[EB];
                    bool ok[EB];
#pragma unroll
                    for (int u = 0; u < EB; ++u) {
                        int oy, ox;
                        if (p.nfold > 1) {
                            oy = 2 * py + (cfold >> 1), ox = 2 * px + (cfold & 1);
                        } else {
                            oy = cl.oy0 + py * p.s_out, ox = cl.ox0 + px * p.s_out;
                        }
                        ok[u] = m < M && n_ok && oy < p.Hout && ox < p.Wout;
                        oo[u] = (pb * p.Hout + oy) * p.Wout + ox;
                        pre[u] = fast_epi_load<float>(fe, p, oo[u], n_e, ok[u]);
                        m += RPI;
                        px += RPI;
                        while (px >= p.Wm) {
                            px -= p.Wm;
                            py += 1;
                        }
                        while (py >= p.Hm) {
                            py -= p.Hm;
                            pb += 1;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < EB; ++u) {
                        const f32x4 t = *reinterpret_cast<const f32x4*>(slab + ((i0 + u) * RPI + lane / LPR) * PITCH + 4 * (lane % LPR));
                        fast_epi_store<float>(fe, p, oo[u], n_e, ok[u], t, pre[u]);
                    }
                }
                break;
            }
#pragma unroll
            for (int i = 0; i < 32 / RPI; ++i) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(slab + (i * RPI + lane / LPR) * PITCH + 4 * (lane % LPR));
                if (m < M) {
                    float v[4] = {t[0], t[1], t[2], t[3]};
                    if (p.nfold > 1) {
                        const int oy = 2 * py + (cfold >> 1), ox = 2 * px + (cfold & 1);
                        if (cfold < p.nfold && oy < p.Hout && ox < p.Wout)
                            store4_t<float>(p, ((size_t)pb * p.Hout + oy) * p.Wout + ox, n0 - cfold * p.Cout, v, vec);
                    } else if (linear) {
                        store4_t<float>(p, (size_t)m, n0, v, vec);
                    } else {
                        const int oy = cl.oy0 + py * p.s_out, ox = cl.ox0 + px * p.s_out;
                        if (oy < p.Hout && ox < p.Wout) store4_t<float>(p, ((size_t)pb * p.Hout + oy) * p.Wout + ox, n0, v, vec);
                    }
                }
                m += RPI;
                px += RPI;
                while (px >= p.Wm) {
                    px -= p.Wm;
                    py += 1;
                }
                while (py >= p.Hm) {
                    py -= p.Hm;
                    pb += 1;
                }
            }
            break;
        }
    }
    if (p.nfold > 1) {
        if constexpr (SH == 32) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc32[j][4 * g + e];
                    store4_fold_t<float>(p, m_blk_e + 32 * wave + (lane & 31), M, HWm, n_blk_e + 32 * j + 8 * g + 4 * (lane >> 5), v, vec);
                }
        } else {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    float v[4] = {acc16[ib][j][0], acc16[ib][j][1], acc16[ib][j][2], acc16[ib][j][3]};
                    store4_fold_t<float>(p, m_blk_e + 32 * wave + 16 * ib + (lane & 15), M, HWm, n_blk_e + 16 * j + 4 * (lane >> 4), v, vec);
                }
        }
        break;
    }
    if constexpr (SH == 32) {
        // D layout of a 32x32 tile: column (lane & 31) = pixel, row (r&3) + 8*(r>>2) + 4*(lane>>5) = output channel:
        // registers 4g..4g+3 of a lane are 4 consecutive channels of its pixel.
        size_t o;
        if (!out_pixel(p, cl, m_blk_e + 32 * wave + (lane & 31), M, HWm, o)) break;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc32[j][4 * g + e];
                store4_t<float>(p, o, n_blk_e + 32 * j + 8 * g + 4 * (lane >> 5), v, vec);
            }
    } else {
        // D layout of a 16x16 tile: column (lane & 15) = pixel, rows 4*(lane>>4) + i = 4 consecutive output channels
        if (fast_epi_ok(p, vec)) {   // branch-free operand accesses, a pixel's TJ channel quads in flight (epilogue.hpp: fast_epi_*)
            const fast_epi_t fe = make_fast_epi(p, 0);
            size_t o[2] = {0, 0};
            bool okp[2];
            fast_pre_t<float> pre[2][TJ];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) okp[ib] = out_pixel(p, cl, m_blk_e + 32 * wave + 16 * ib + (lane & 15), M, HWm, o[ib]);
            // narrow tiles: both pixel blocks' operands in flight; wide ones: one block at a time (registers)
#pragma unroll
            for (int ib = 0; ib < (TJ <= 4 ? 2 : 1); ++ib)
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int n = n_blk_e + 16 * j + 4 * (lane >> 4);
                    pre[ib][j] = fast_epi_load<float, true>(fe, p, (int)o[ib], n, okp[ib] && n < p.Cout);
                }
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                if (TJ > 4 && ib == 1) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
                        const int n = n_blk_e + 16 * j + 4 * (lane >> 4);
                        pre[1][j] = fast_epi_load<float, true>(fe, p, (int)o[1], n, okp[1] && n < p.Cout);
                    }
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int n = n_blk_e + 16 * j + 4 * (lane >> 4);
                    fast_epi_store<float, f32x4, true>(fe, p, (int)o[ib], n, okp[ib] && n < p.Cout, acc16[ib][j], pre[ib][j]);
                }
            }
            break;
        }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            size_t o;
            if (!out_pixel(p, cl, m_blk_e + 32 * wave + 16 * ib + (lane & 15), M, HWm, o)) continue;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float v[4] = {acc16[ib][j][0], acc16[ib][j][1], acc16[ib][j][2], acc16[ib][j][3]};
                store4_t<float>(p, o, n_blk_e + 16 * j + 4 * (lane >> 4), v, vec);
            }
        }
    }
    } while (0);
    if constexpr (CO) {  // (its epilogue goes through the stage buffers: set the next tile up afterwards)
        if (has_next) X6D_NEXT_TILE()
    }
#undef X6D_NEXT_TILE
    if (!has_next) break;
  }
#undef X6D_ISSUE
#undef X6D_LDP
#undef X6D_PREP
#undef X6D_TILE_SETUP
#undef X6D_TILE_COORDS
#undef X6D_ADVANCE
#undef X6D_DMA_A
#undef X6D_DMA_W
#undef X6D_LDW
#undef X6D_MFMA6
}

// second pass of split-K: out = epilogue( sum over splits, in fixed order ), 4 channels per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const spaa_tapconv_t p, const int M, const int npad) {
    const int nq = (p.Cout + 3) >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * nq) return;
    const int m = (int)(idx / nq), n0 = (int)(idx - (int64_t)m * nq) * 4;
    const spaa_tapclass_t cl = p.cls[0];
    f4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.ksplit; ++s) sum += *reinterpret_cast<const f4*>(p.splitk_ws + ((size_t)s * M + m) * npad + n0);
    size_t o;
    if (!out_pixel(p, cl, m, M, p.Hm * p.Wm, o)) return;
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    float v[4] = {sum.x, sum.y, sum.z, sum.w};
    store4_t<float>(p, o, n0, v, vec);
}

// second pass of stream-K: tiles that were cut into segments are summed in segment order and finished.  Workgroup s
// of XCD x owns iterations [s*T/G, (s+1)*T/G) of that XCD's q_x*nk iterations; iteration I belongs to workgroup
// ceil((I+1)*G/T) - 1.
template <int BM, int BN>
__global__ __launch_bounds__(256) void streamk_reduce_kernel(const spaa_tapconv_t p, const int m_tiles, const int n_tiles,
                                                             const int gx, const int nk_all) {
    const int nwg = m_tiles * n_tiles;
    const int orig = blockIdx.y;  // tile in launch order: XCD = orig & 7, position in the XCD's list = orig >> 3
    const int xcd = orig & 7, j = orig >> 3;
    const int q_x = (nwg - xcd + 7) >> 3, G8 = (gx - xcd + 7) >> 3;
    const int64_t T = (int64_t)q_x * nk_all;
    const int s_a = (int)((((int64_t)j * nk_all + 1) * G8 + T - 1) / T) - 1;
    const int s_b = (int)((((int64_t)(j + 1) * nk_all) * G8 + T - 1) / T) - 1;
    if (s_a == s_b) return;  // the tile was computed whole by one workgroup
    const int e = blockIdx.x * 256 + threadIdx.x;  // 4 channels of one tile row
    if (e >= BM * BN / 4) return;
    const int row = e / (BN / 4), col = (e - row * (BN / 4)) * 4;
    const size_t slot = (size_t)BM * BN;
    const float* ws = p.splitk_ws;
    f4 sum = *reinterpret_cast<const f4*>(ws + ((size_t)(xcd + 8 * s_a) * 2 + 1) * slot + row * BN + col);
    for (int s = s_a + 1; s <= s_b; ++s)
        sum += *reinterpret_cast<const f4*>(ws + ((size_t)(xcd + 8 * s) * 2 + 0) * slot + row * BN + col);
    const int q = nwg >> 3, r = nwg & 7;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    const int n0 = (tile % n_tiles) * BN + col;
    const int m = (tile / n_tiles) * BM + row;
    const spaa_tapclass_t cl = p.cls[0];
    const int M = p.B * p.Hm * p.Wm;
    size_t o;
    if (!out_pixel(p, cl, m, M, p.Hm * p.Wm, o)) return;
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    float v[4] = {sum.x, sum.y, sum.z, sum.w};
    store4_t<float>(p, o, n0, v, vec);
}

constexpr int STREAMK_MAX_WG = 768;  // workspace contract: 2 * STREAMK_MAX_WG * 128 * 128 floats

template <int NW, int BN, int SH = 32, bool CO = false, int NA = 2>
int launch_x6d(const spaa_tapconv_t& d, hipStream_t stream, bool persistent = false) {
    constexpr int BM = 32 * NW;
    const int64_t M = (int64_t)d.B * d.Hm * d.Wm;
    const int m_tiles = (int)((M + BM - 1) / BM);
    const int nfold = d.nfold > 1 ? d.nfold : 1;
    if (nfold > 1 && (nfold != 4 || d.nclass != 1 || d.s_out != 2 || (d.Cout & 3) || d.ksplit > 1)) return hipErrorInvalidValue;
    const int n_tiles = (d.Cout * nfold + BN - 1) / BN;
    const size_t smem = (size_t)NA * (BM * 128) + 2 * (size_t)(3 * BN * 64);
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&tapconv_x6d_kernel<NW, BN, SH, CO, NA>), (int)smem, attr_set);
        if (e != hipSuccess) return (int)e;
    }
    const bool streamk = d.ksplit == -1;
    if (streamk && (!persistent || CO || d.nclass != 1 || d.nfold > 1 || d.splitk_ws == nullptr ||
                    (int64_t)m_tiles * ((d.Cout + BN - 1) / BN) > 65535))
        return hipErrorInvalidValue;
    const int ksplit = d.ksplit > 1 ? d.ksplit : 1;
    if (ksplit > 1 && (d.nclass != 1 || d.splitk_ws == nullptr || d.cls[0].Kpad / BK < 2 * ksplit)) return hipErrorInvalidValue;
    int gx = m_tiles * n_tiles;
    if (persistent) {
        // as many workgroups as the chip holds at once (LDS-limited), a multiple of 8 so every XCD gets the same number;
        // each walks its XCD's tiles and overlaps a tile's epilogue with the next tile's first gathers
        const int per_cu = (int)(160 * 1024 / smem) < 1 ? 1 : (int)(160 * 1024 / smem);
        const int resident = 256 * (per_cu > 8 * 4 / NW ? 8 * 4 / NW : per_cu);
        int cap = resident / (d.nclass * ksplit);
        if ((d.reserved0 >> 8) > 0) cap = d.reserved0 >> 8;  // tests: force several tiles per workgroup on small layers
        cap = cap < 8 ? 8 : (cap & ~7);
        if (streamk && cap > STREAMK_MAX_WG) cap = STREAMK_MAX_WG;
        if (streamk) {
            // every resident workgroup gets a share of the K-steps, however few tiles there are (>= 4 steps each)
            const int64_t iters = (int64_t)gx * (d.cls[0].Kpad / BK);
            gx = cap;
            while (gx > 8 && iters / gx < 4) gx -= 8;
        } else if (gx > cap) {
            gx = cap;
        }
    }
    dim3 grid(gx, d.nclass, ksplit);
    hipLaunchKernelGGL((tapconv_x6d_kernel<NW, BN, SH, CO, NA>), grid, dim3(64 * NW), smem, stream, d, m_tiles, n_tiles);
    if (streamk) {
        dim3 rgrid((BM * BN / 4 + 255) / 256, m_tiles * n_tiles, 1);
        hipLaunchKernelGGL((streamk_reduce_kernel<BM, BN>), rgrid, dim3(256), 0, stream, d, m_tiles, n_tiles, gx,
                           d.cls[0].Kpad / BK);
    }
    if (ksplit > 1) {
        const int npad = (d.Cout + 127) & ~127;
        const int64_t nthr = M * ((d.Cout + 3) >> 2);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, d, (int)M, npad);
    }
    return (int)hipGetLastError();
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tiles 25.. after the common shape checks
int spaa_launch_tapconv_x6d(const spaa_tapconv_t& d, int tile, hipStream_t stream) {
    if (d.w_split == nullptr || (d.Cin % BK) != 0) return hipErrorInvalidValue;
    for (int c = 0; c < d.nclass; ++c) {
        if ((int64_t)((d.Cout * (d.nfold > 1 ? d.nfold : 1) + 127) & ~127) * d.cls[c].Kpad * 6 >= (int64_t)1 << 31)
            return hipErrorInvalidValue;
        if (d.cls[c].Kpad != d.cls[c].K) return hipErrorInvalidValue;
    }
    switch (tile) {
        case 25: return launch_x6d<4, 128>(d, stream);
        case 26: return launch_x6d<8, 128>(d, stream);
        case 27: return launch_x6d<4, 64>(d, stream);
        case 30: return launch_x6d<4, 32>(d, stream);
        case 31: return launch_x6d<2, 64>(d, stream);
        case 32: return launch_x6d<2, 128>(d, stream);
        case 33: return launch_x6d<8, 64>(d, stream);
        case 34: return launch_x6d<4, 128, 16>(d, stream);
        case 35: return launch_x6d<8, 128, 16>(d, stream);
        case 36: return launch_x6d<4, 64, 16>(d, stream);
        case 37: return launch_x6d<4, 32, 16>(d, stream);
        case 39: return launch_x6d<4, 128, 16, true>(d, stream);
        case 40: return launch_x6d<4, 64, 16, true>(d, stream);
        case 41: return launch_x6d<4, 32, 16, true>(d, stream);
        case 42: return launch_x6d<4, 64, 16, false, 3>(d, stream);
        case 43: return launch_x6d<4, 32, 16, false, 3>(d, stream);
        case 44: return launch_x6d<4, 64, 32, false, 3>(d, stream);
        case 45: return launch_x6d<4, 64, 16, true, 3>(d, stream);
        case 46: return launch_x6d<4, 32, 16, true, 3>(d, stream);
        case 48: return launch_x6d<4, 128, 16>(d, stream, true);
        case 49: return launch_x6d<4, 64, 16>(d, stream, true);
        case 50: return launch_x6d<4, 64, 16, false, 3>(d, stream, true);
        case 51: return launch_x6d<4, 64, 32, false, 3>(d, stream, true);
        case 52: return launch_x6d<8, 128, 16>(d, stream, true);
        case 53: return launch_x6d<4, 32, 16>(d, stream, true);
        case 54: return launch_x6d<4, 128, 32>(d, stream, true);
        default: return hipErrorInvalidValue;
    }
}
