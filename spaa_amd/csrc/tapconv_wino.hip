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

template <int BN, int VAR, int DBG = 0>   // DBG: timing-only ablations (wrong results): 1 no epilogue, 2 no fold, 4 no V, 8 no barrier, 16 no DMA; 32: raw barrier
__global__ __launch_bounds__(512, 1) void wino_x6_kernel(const spaa_tapconv_t p, const int wg_y, const int wg_x, const int n_tiles) {
    constexpr int TJ = BN / 16;
    constexpr int W_PLANE = BN * 64;
    constexpr int NW = 8;
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

    // XCD-aware order over (image, patch row, patch column, n tile): an XCD takes a contiguous range
    int n_blk, img, oy0, ox0;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        n_blk = (t % n_tiles) * BN;
        t /= n_tiles;
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

    const int nkb = Cin >> 5;           // 32-channel blocks
    const int nsteps = nkb * 16;        // (block, position) steps
#define WINO_DMA_W(st, step)                                                                                      \
    {                                                                                                              \
        const int kb_ = (step) >> 4, pos_ = (step) & 15;                                                           \
        const int soff_ = (pos_ * Cin + kb_ * 32) * 2;                                                             \
        _Pragma("unroll") for (int i = 0; i < WPW; ++i)                                                             \
            dma16(rsrc_w, wsm + (st) * WS_BYTES + (wave + NW * i) * 1024, w_goff[i], soff_);                       \
    }
    // pipeline: weight stages three deep (the DMA of step t+2 is issued at step t); the patch of the next channel block is
    // requested as soon as the last reads of the current one (the row combination of xi = 3) are done.
    // Waves w and w+4 share a SIMD: waves 0-3 form V(t) at the START of step t, waves 4-7 form V(t+1) at the END of step t, so
    // that one wave's transform (VALU) runs under its partner's MFMAs.
#define WINO_DMA_PATCH(kb_)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                                \
        const int slot_ = (wave + NW * i) * 8 + (lane >> 3);                                                         \
        const int pix_ = slot_ ^ ((slot_ >> 1) & 1);                                                                 \
        const int pr_ = pix_ / PW, pc_ = pix_ - pr_ * PW;                                                            \
        const int iy_ = oy0 - 1 + pr_, ix_ = ox0 - 1 + pc_;                                                          \
        const bool ok_ = pix_ < NPX && (unsigned)iy_ < (unsigned)H && (unsigned)ix_ < (unsigned)W;                   \
        const int u_ = pc_ >> 1;                                                                                     \
        const int lc_ = ((((lane & 7) >> 1) - 2 * (u_ >> 2)) & 3) * 2 + ((lane & 1) ^ ((u_ >> 1) & 1));               \
        const int off_ = ok_ ? ((img * H + iy_) * W + ix_) * row_bytes + p.in_coff * 4 + lc_ * 16 + (kb_) * 128      \
                             : (int)0x80000000;                                                                      \
        dma16(rsrc_in, smem + (wave + NW * i) * 1024, off_, 0);                                                      \
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
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
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
    if (fast_epi_ok(p, vec)) {
        // The operand combinations of the attack loops, branch-free (epilogue.hpp: fast_epi_*): a wave's 32 x 2 pixels go
        // through a PRIVATE LDS region (row = 16 (pixel column & 1) + tile column, padded by 16 bytes: conflict-free writes
        // from the MFMA layout, a lane keeps ONE channel quad), so that one barrier after the main loop is all the
        // synchronisation there is, and the residual / gate operands of eight pixels per lane are in flight together.
        constexpr int ROWP = BN * 4 + 16;
        const int n = n_blk + 4 * (lane & (LPP - 1));
        const bool n_ok = n < p.Cout;
        const fast_epi_t fe = make_fast_epi(p, n_ok ? n : 0);
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
            const int orow = (img * p.Hout + oy) * p.Wout + ox0;
            constexpr int EB = 8;
#pragma unroll 1
            for (int it0 = 0; it0 < 32 / PPI; it0 += EB) {
                fast_pre_t<float> pre[EB];
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    const int r = (it0 + u) * PPI + lane / LPP, px = 2 * (r & 15) + (r >> 4);
                    pre[u] = fast_epi_load<float>(fe, p, orow + px, n, n_ok && oy < p.Hout && ox0 + px < p.Wout);
                }
#pragma unroll
                for (int u = 0; u < EB; ++u) {
                    const int r = (it0 + u) * PPI + lane / LPP, px = 2 * (r & 15) + (r >> 4);
                    const f32x4 y = *reinterpret_cast<const f32x4*>(eb + r * ROWP + ((lane & (LPP - 1)) << 4));
                    fast_epi_store<float>(fe, p, orow + px, n, n_ok && oy < p.Hout && ox0 + px < p.Wout, y, pre[u]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
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
                ok[u] = oy < p.Hout && ox < p.Wout;
                oo[u] = ((size_t)img * p.Hout + (ok[u] ? oy : 0)) * p.Wout + (ok[u] ? ox : 0);
                if (vec && ok[u]) pre[u] = epi_load<float>(p, oo[u], n_blk + 4 * lc);
            }
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const int pl = PPI * ((32 / PPI) * wave + it0 + u) + lane / LPP;
                const int pc = lane & (LPP - 1);                // physical chunk
                const int lc = pc ^ ((pl & 31) >> 1);           // logical chunk: channels 4 lc .. 4 lc + 3 of this n tile
                const f32x4 y = *reinterpret_cast<const f32x4*>(smem + pl * ROWB + (pc << 4));
                if (ok[u]) {
                    float v[4] = {y[0], y[1], y[2], y[3]};
                    if (vec) store4_pre<float>(p, oo[u], n_blk + 4 * lc, v, pre[u]);
                    else store4_t<float>(p, oo[u], n_blk + 4 * lc, v, vec);
                }
            }
        }
    }
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tile 70 after the common shape checks.  The descriptor describes the layer in
// Winograd form: ONE class with 16 "taps" = the positions of U = G g G^T (tap entries unused), s_in = s_out = 1, same input and
// output size, Cin % 32 == 0.
int spaa_launch_tapconv_wino(const spaa_tapconv_t& d, hipStream_t stream) {
    if (d.w_split == nullptr || (d.Cin % 32) != 0 || d.nclass != 1 || d.cls[0].ntaps != 16 || d.cls[0].K != 16 * d.Cin || d.cls[0].Kpad < d.cls[0].K || (d.cls[0].Kpad & 7) ||
        d.s_in != 1 || d.s_out != 1 || d.Hin != d.Hout || d.Win != d.Wout || d.Hm != d.Hout || d.Wm != d.Wout || d.nfold > 1 ||
        d.ksplit > 1 || d.ksplit < 0 || d.io_dtype != 0)
        return hipErrorInvalidValue;
    if ((int64_t)((d.Cout + 127) & ~127) * d.cls[0].Kpad * 6 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    const int wg_y = (d.Hout + 2 * TY - 1) / (2 * TY), wg_x = (d.Wout + 2 * TX - 1) / (2 * TX);
    // N tile: the 64-wide instantiation for layers with at most 64 output channels, and where the 128-wide grid would leave
    // compute units without a workgroup (one workgroup per CU: ResNet layer2, 28 x 28 images)
    static int ncu[SPAA_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SPAA_MAX_DEVICES) return hipErrorInvalidValue;
    if (ncu[dev] == 0 && hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu[dev] = 256;
    const int BN = (d.Cout <= 64 || (int64_t)d.B * wg_y * wg_x * ((d.Cout + 127) / 128) < ncu[dev]) ? 64 : 128;
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const int64_t nwg = (int64_t)d.B * wg_y * wg_x * n_tiles;
    if (nwg > 0x7fffffff) return hipErrorInvalidValue;
    static bool attr_set[8][SPAA_MAX_DEVICES] = {};
    // kernel variant: bit 0 = late V (waves 4-7 transform one step ahead), bit 1 = xi groups expanded.  Default 3 / 2 (measured:
    // conv4 500 -> 462 us, conv5 461 -> 415 us against variant 0); `reserved0` bits 16-17 flip bits for A/B measurements
    const int var = (BN == 64 ? 2 : 3) ^ ((d.reserved0 >> 16) & 3);   // (64-wide tile: late V does not pay: 168 / 167 / 159 us for 0 / 3 / 2)
#define WINO_LAUNCH(N, V, SLOT)                                                                                           \
    {                                                                                                                     \
        const size_t smem = (size_t)PATCH_BYTES + 3 * (size_t)(((3 * N / 16 + 7) / 8) * 8 * 1024);                        \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<N, V>), (int)smem, attr_set[SLOT]); \
        if (e != hipSuccess) return (int)e;                                                                               \
        hipLaunchKernelGGL((wino_x6_kernel<N, V>), dim3((unsigned)nwg), dim3(512), smem, stream, d, wg_y, wg_x, n_tiles); \
    }
#ifdef SPAA_WINO_ABLATE
    const int dbg = (d.reserved0 >> 18) & 127;
#define WINO_LAUNCH_DBG(D)                                                                                                \
    if (dbg == D) {                                                                                                       \
        static bool as_[SPAA_MAX_DEVICES] = {};                                                                           \
        const size_t smem = (size_t)PATCH_BYTES + 3 * (size_t)(((3 * 128 / 16 + 7) / 8) * 8 * 1024);                      \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<128, 3, D>), (int)smem, as_);     \
        if (e != hipSuccess) return (int)e;                                                                               \
        hipLaunchKernelGGL((wino_x6_kernel<128, 3, D>), dim3((unsigned)nwg), dim3(512), smem, stream, d, wg_y, wg_x, n_tiles); \
        return (int)hipGetLastError();                                                                                    \
    }
    if (BN == 128) {
        WINO_LAUNCH_DBG(1) WINO_LAUNCH_DBG(3) WINO_LAUNCH_DBG(5) WINO_LAUNCH_DBG(7) WINO_LAUNCH_DBG(15) WINO_LAUNCH_DBG(31) WINO_LAUNCH_DBG(2) WINO_LAUNCH_DBG(4) WINO_LAUNCH_DBG(8) WINO_LAUNCH_DBG(32) WINO_LAUNCH_DBG(33) WINO_LAUNCH_DBG(64) WINO_LAUNCH_DBG(96)
    }
#undef WINO_LAUNCH_DBG
#endif
    if (BN == 64) {
        if (var == 3) WINO_LAUNCH(64, 3, 4) else if (var == 2) WINO_LAUNCH(64, 2, 5) else if (var == 1) WINO_LAUNCH(64, 1, 6) else WINO_LAUNCH(64, 0, 7)
    } else if (var == 3) WINO_LAUNCH(128, 3, 3) else if (var == 2) WINO_LAUNCH(128, 2, 2) else if (var == 1) WINO_LAUNCH(128, 1, 1) else WINO_LAUNCH(128, 0, 0)
#undef WINO_LAUNCH
    return (int)hipGetLastError();
}
