// tapconv_x6p.hip — stride-2 FRACTIONAL layers on the bf16x6 arithmetic with the input patch staged once per channel block:
// nn.ConvTranspose2d(k3, s2) forward (ShadingNetSPAA.transConv1, /root/reference/src/python/models.py:237,299) and the input
// gradient of a k3 / s2 convolution (conv2, conv2_s; the stride-2 convolutions of the classifier bodies), i.e. FOUR output-parity
// classes whose taps all lie in one 2 x 2 neighbourhood of the class-grid pixel (1 + 2 + 2 + 4 = 9 (class, tap) pairs).
//
// The implicit-GEMM kernels run the classes as four GEMMs: the input is gathered once per class AND tap (transConv1: 1.07 GB of
// operand traffic for 670 MB of tensors, 353 us at 0.72 PFLOP/s of bf16 MFMA), the folded form (classes stacked in the GEMM
// rows) multiplies 16 (class, tap) blocks of which 7 are zero.  Here
//   * a workgroup (4 waves) owns 4 x 32 class-grid pixels = 8 x 64 output pixels x BN channels; its 5 x 33-pixel input patch is
//     staged ONCE per 32-channel block by LDS-DMA (fp32, 21 KB, double-buffered; out-of-image = the out-of-range offset = the
//     zero padding; chunk swizzle: conflict-free ds_read_b128 for 16 consecutive pixels);
//   * wave w owns class-grid row w: two 16-pixel blocks x 4 classes x BN channels of accumulators; K order: channel block, window
//     position, class: the wave reads its two pixel fragments at a position ONCE (under the previous position's last products),
//     splits them into three bf16 planes (x == h + m + l exactly) and multiplies them with the weight planes of every class
//     that has a tap there -- a (class, tap) "combo" = 12 KB of planes, three LDS stages, DMA two combos ahead behind a counted
//     vmcnt and a raw s_barrier; 6 bf16 MFMAs per product block, exactly the 9 real combos;
//   * optional SECOND SOURCE: a 1 x 1 convolution of a tensor at OUTPUT resolution added to the result (in2 / w2_split /
//     Cin2: ShadingNetSPAA's `transConv1(x) + skipConv2(x1)`, models.py:293,299; backward `conv2^T(g) + skipConv2^T(g6)`):
//     every pixel of it is needed exactly once, so its fragments go from global memory straight to registers;
//   * epilogue through a wave-private LDS region in OUTPUT-pixel order (the two classes of an output row interleaved: 32
//     consecutive pixels x BN channels per pass) and the shared branch-free epilogue (bias, residual, ReLU, byte masks).
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
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
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

// A workgroup = FOUR waves (one per SIMD), TWO workgroups per compute unit (79 KB of LDS each): the two waves of a SIMD belong
// to different workgroups, so one's fragment splits, barriers, prologue and epilogue (256 KB of stores per workgroup) run under
// the other's matrix-core work -- with one 8-wave workgroup per CU every wave reaches those phases at the same time (measured:
// transConv1 288 -> 260 us; profiles/r04_x6p_time.txt).
constexpr int NW = 4;
constexpr int RY = NW, RX = 32;                // class-grid pixels of a workgroup (rows x columns): 8 x 64 output pixels
constexpr int PH = RY + 1, PW = RX + 1;        // input patch 5 x 33 (taps in a 2 x 2 window)
constexpr int NPX = PH * PW;                   // 165
constexpr int P_PIECES = (NPX + 7) / 8;        // 1-KiB pieces of 8 pixels x 32 channels (fp32): 21
constexpr int PPW = (P_PIECES + NW - 1) / NW;  // DMAs per wave and patch: 6 (pieces 21 .. 23: the out-of-range offset into DUMP)
constexpr int PATCH_BYTES = P_PIECES * 1024;   // 21504
constexpr int DUMP_OFF = 2 * PATCH_BYTES;      // 1 KiB that the pad pieces of both patch buffers write (zeros) into
constexpr int W_OFF = DUMP_OFF + 1024;         // the weight stages

// 64-byte weight rows (32 bf16), chunk swizzle as tapconv_x6d.hip swz_w<16>
__device__ __forceinline__ int swz_w16(int n) { return ((n >> 3) & 1) << 1; }
// 128-byte pixel rows (32 fp32): logical 16-byte chunk c of patch pixel q sits at chunk c ^ swz_p(q).  A ds_read_b128 is served in
// groups of 16 lanes = 8 pixels of one channel-chunk pair + the other 8 pixels of the next pair (MI355X_MICROARCH.md, LDS):
// consecutive pixels q .. q + 15 hit 16 distinct 16-byte slots of the 256-byte bank row when the lanes of ODD chunk pairs read
// their two chunks in the opposite order (frag_addr below)
__device__ __forceinline__ int swz_p(int q) { return ((q >> 1) & 3) << 1; }

#ifdef SPAA_X6P_STAMP
// timing-only diagnostic build (make stamp -> libspaa_hip_stamp.so, tools/lab/x6p_stamps.py): wave 0 of every workgroup records
// s_memtime at its phase boundaries and the cycles it spends in the per-step wait + barrier; no output value depends on them
__device__ unsigned long long* g_x6p_stamps = nullptr;
#define X6P_T() (stamp_on ? __builtin_amdgcn_s_memtime() : 0ull)
#define X6P_ABL(bit) ((abl >> (bit)) & 1)
#else
#define X6P_ABL(bit) 0
#endif

// STD (`reserved2` bit 0, vouched for by spaa_amd/convplan.py which sees the tap lists): the CANONICAL structure of a 3 x 3 / stride-2
// fractional layer -- tap window (0..1) x (0..1); class 0 = [(0,0)], class 1 = [(0,1), (0,0)], class 2 = [(1,0), (0,0)], class 3 =
// [(1,1), (1,0), (0,1), (0,0)], which is what ConvTranspose2d(k3, s2, p1, op1) and the input gradient of Conv2d(k3, s2, p1) give.
// The nine combos' (class, tap, window position) are then compile-time constants: no schedule words, no divisions by the combo
// count, no per-step scalar loads of the classes' Kpad / w_off from the argument buffer, weight-row offsets by shifts (Kpad =
// Cin x 1 / 2 / 2 / 4), and the 16-entry tap table with its 68-83 spilled scalar registers is gone.
constexpr int STD_CLS[9] = {0, 1, 2, 3, 1, 3, 2, 3, 3};     // combo n -> class
constexpr int STD_TAP[9] = {0, 1, 1, 3, 0, 2, 0, 1, 0};     //         -> tap index inside the class
constexpr int STD_TIX[4][4] = {{0, -1, -1, -1}, {1, 0, -1, -1}, {1, -1, 0, -1}, {3, 2, 1, 0}};   // [class][window position] -> tap or -1
constexpr int STD_LG[4] = {0, 1, 1, 2};                     // log2(taps of the class)
constexpr int STD_WOFF[4] = {0, 1, 3, 5};                   // w_off of the class in units of Npad x Cin

template <int BN, bool STD = false>
__global__ __launch_bounds__(64 * NW, 2) void x6p_kernel(const spaa_tapconv_t p, const int wg_y, const int wg_x, const int n_tiles) {
    constexpr int TJ = BN / 16;
    constexpr int W_PLANE = BN * 64;                  // one plane of a combo: BN rows of 32 bf16
    constexpr int W_PIECES = 3 * BN / 16;             // 1-KiB pieces of a combo's three planes: 12 (BN = 64) or 6
    constexpr int WPW = (W_PIECES + NW - 1) / NW;     // per wave: 2 or 1 (pieces past the planes: the out-of-range offset)
    constexpr int WS_BYTES = WPW * NW * 1024;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const wsm = smem + W_OFF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = p.Cin, H = p.Hin, W = p.Win;

    int n_blk, img, oy0, ox0;   // (oy0, ox0): the region's origin on the class grid
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        n_blk = (t % n_tiles) * BN;
        t /= n_tiles;
        ox0 = (t % wg_x) * RX;
        t /= wg_x;
        oy0 = (t % wg_y) * RY;
        img = t / wg_y;
    }
#ifdef SPAA_X6P_STAMP
    // reserved0 bits 8-15 (this build only): timing ablations -- wrong results -- 1 no DMA after the first stages, 2 no fragment split,
    // 4 no barrier, 8 no MFMA, 16 no epilogue, 32 no fragment loads; 128: record the stamps
    const int abl = (p.reserved0 >> 8) & 0xff;
    const bool stamp_on = (abl & 128) != 0;
    const unsigned long long ts0 = X6P_T(), tr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long ts_wait = 0, ts_first = 0;
#endif
    const int row_bytes = p.in_cstride * 4;
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(H * W) * (uint32_t)row_bytes;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int npad = (p.Cout + 127) & ~127;
    // the four classes' weight planes sit back to back in w_split: class c at element 3 * cls[c].w_off, [3][Npad][Kpad_c]
    const int64_t w_total = p.cls[3].w_off + (int64_t)npad * p.cls[3].Kpad;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_split);
    const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane((uint32_t)(6 * w_total)), 0x00020000);
    typedef const __attribute__((address_space(4))) int* cint_ptr;
    cint_ptr ctaps = (cint_ptr)(uintptr_t)p.taps;
    const int dy_min = p.tap_range[0], dx_min = p.tap_range[2];

    // combos = (class c, tap t of the class), ordered by the tap's POSITION in the 2 x 2 window (a wave reads and splits its pixel
    // fragments once per position, for every class that has a tap there), then by class.  tix[c][pos] = tap index or -1;
    // sched = the combos in that order, 4 bits each ((c << 2) | t).  Class 3 = parity (1, 1) has a tap at every position
    // (spaa_amd/convplan.py x6p_ok): the fragments of the next position are fetched under ITS products.
    int tix[4][4];
    unsigned long long sched = 0;
    int ncombo = 0;
    if constexpr (STD) {
        ncombo = 9;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) tix[c][ps] = -1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int nt = p.cls[c].ntaps, toff = p.cls[c].tap_off;
            for (int t = 0; t < nt; ++t) {
                const int ps = 2 * (ctaps[2 * (toff + t)] - dy_min) + (ctaps[2 * (toff + t) + 1] - dx_min);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (ps == q) tix[c][q] = t;
            }
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (tix[c][ps] >= 0) {
                    sched |= (unsigned long long)((c << 2) | tix[c][ps]) << (4 * ncombo);
                    ++ncombo;
                }
    }
    const int nkb = Cin >> 5;
    const int nsteps = nkb * ncombo;

    // ---- patch staging: piece i (8 consecutive patch pixels) -> wave i % 8; lane -> (pixel lane >> 3, physical chunk lane & 7)
    auto dma_patch = [&](const int buf, const int kb) {
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (per-lane constants recomputed here, not kept in registers across the K loop)
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = (wave + NW * i) * 8 + (ln >> 3);           // patch pixel
            const int pr = q / PW, pc = q - pr * PW;
            const int iy = oy0 + dy_min + pr, ix = ox0 + dx_min + pc;
            const bool ok = q < NPX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int c = (ln & 7) ^ swz_p(q);                       // logical chunk that lands in this lane's physical chunk
            const int off = ok ? ((img * H + iy) * W + ix) * row_bytes + (p.in_coff + kb * 32) * 4 + c * 16 : (int)0x80000000;
            const int pi = wave + NW * i;
            dma16(rsrc_in, smem + (pi < P_PIECES ? buf * PATCH_BYTES + pi * 1024 : DUMP_OFF), off, 0);
        }
    };
    // ---- weights of step s = kb * ncombo + n: combo n = (class c, tap t): planes [3][rows n_blk ..][columns t * Cin + kb * 32 ..]
    auto dma_w = [&](const int stage, const int s) {
        const int kb = s / ncombo, n = s - kb * ncombo;
        const int ct = (int)(sched >> (4 * n)) & 15;
        const int c = ct >> 2, t = ct & 3;
        const int kpad = p.cls[c].Kpad;
        const int soff = (int)(6 * p.cls[c].w_off) + (t * Cin + kb * 32) * 2;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int q = wave + NW * i;                             // piece: (plane q / (BN / 16), 16-row block q % (BN / 16))
            const int pl = q / (BN / 16), rb = q - pl * (BN / 16);
            const int nrow = 16 * rb + (lane >> 2);
            const int ch = (lane & 3) ^ swz_w16(nrow);
            const int voff = q < W_PIECES ? (pl * npad + n_blk + nrow) * kpad * 2 + ch * 16 : (int)0x80000000;
            dma16(rsrc_w, wsm + stage * WS_BYTES + q * 1024, voff, soff);
        }
    };
    // STD: per-lane row offsets of this wave's weight pieces for a class with ONE tap (Kpad = Cin); a class with 2 / 4 taps: << 1 / 2
    int wrow2[WPW], wch16[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int q = wave + NW * i;
        const int pl = q / (BN / 16), rb = q - pl * (BN / 16);
        const int nrow = 16 * rb + (lane >> 2);
        wrow2[i] = (pl * npad + n_blk + nrow) * Cin * 2;
        wch16[i] = (((lane & 3) ^ swz_w16(nrow)) << 4) | (q < W_PIECES ? 0 : (int)0x80000000);
    }
    const int wcls_bytes = 6 * npad * Cin;     // bytes of the three planes of a one-tap class
    auto dma_w_std = [&](const int stage, const int kb2, const int n2) {   // (n2: a compile-time constant wherever this is expanded)
        const int c = STD_CLS[n2], t = STD_TAP[n2];
        const int soff = STD_WOFF[c] * wcls_bytes + (t * Cin + kb2 * 32) * 2;
#pragma unroll
        for (int i = 0; i < WPW; ++i) dma16(rsrc_w, wsm + stage * WS_BYTES + (wave + NW * i) * 1024, (wrow2[i] << STD_LG[c]) + wch16[i], soff);
    };
    const int w_addr_l = (lane & 15) * 64 + (((lane >> 4) ^ swz_w16(lane & 15)) * 16);
    const int q8 = lane >> 4;

    f32x4 acc[4][2][TJ];   // [class][pixel block][16-channel block]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[c][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one combo's products: two pixel blocks x TJ channel blocks x six bf16 MFMAs (small terms first, tapconv_x6d.hip X6D_MFMA6)
#define X6P_MFMA(c_, wbase)                                                                                        \
    {                                                                                                              \
        /* (the weight fragments of channel block j + 1 are requested before block j's MFMAs are issued) */         \
        bf16x8 wf_[2][3];                                                                                          \
        _Pragma("unroll") for (int k = 0; k < 3; ++k) wf_[0][k] = *reinterpret_cast<const bf16x8*>((wbase) + k * W_PLANE); \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                                          \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                            \
            if (j + 1 < TJ) {                                                                                      \
                _Pragma("unroll") for (int k = 0; k < 3; ++k)                                                       \
                    wf_[(j + 1) & 1][k] = *reinterpret_cast<const bf16x8*>((wbase) + (j + 1) * 1024 + k * W_PLANE); \
            }                                                                                                      \
            const bf16x8 w0 = wf_[j & 1][0], w1 = wf_[j & 1][1], w2 = wf_[j & 1][2];                                \
            _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                         \
                f32x4 a_ = acc[c_][b][j];                                                                           \
                a_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, pf[b][0], a_, 0, 0, 0);                            \
                a_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[b][2], a_, 0, 0, 0);                            \
                a_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, pf[b][1], a_, 0, 0, 0);                            \
                a_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, pf[b][0], a_, 0, 0, 0);                            \
                a_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[b][1], a_, 0, 0, 0);                            \
                acc[c_][b][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[b][0], a_, 0, 0, 0);                 \
            }                                                                                                      \
            /* order: [the next block's three fragment reads,] then this block's twelve MFMAs, each with two VALU    \
               instructions in its shadow where there are any (class 3: the split of the next position's fragments) */ \
            if (j + 1 < TJ) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                      \
            _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {                                                     \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                  \
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }

    // ---- prologue: second source (if any), then the pipeline's first patch and weight stages
    const int nkb2 = p.in2 != nullptr ? p.Cin2 >> 5 : 0;     // (launcher: 0, 1 or 2 channel blocks)
    dma_patch(0, 0);
    if constexpr (STD) dma_w_std(0, 0, 0); else dma_w(0, 0);
    bool w1_issued = false;
    if (nkb2 < 2 && nsteps > 1) {
        if constexpr (STD) dma_w_std(1, 0, 1); else dma_w(1, 1);
        w1_issued = true;
    }
    if (nkb2 > 0) {
        // weights of the second source: [3][Npad][Cin2] bf16 planes, one "combo" per 32-channel block, into the ring's last stages
        const uint64_t w2_addr = reinterpret_cast<uint64_t>(p.w2_split);
        const uint32_t w2_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w2_addr);
        const uint32_t w2_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w2_addr >> 32));
        const auto rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w2_hi << 32) | w2_lo), 0,
                                                                (int)__builtin_amdgcn_readfirstlane((uint32_t)(6 * npad * p.Cin2)), 0x00020000);
        for (int k2 = 0; k2 < nkb2; ++k2) {
#pragma unroll
            for (int i = 0; i < WPW; ++i) {
                const int q = wave + NW * i;
                const int pl = q / (BN / 16), rb = q - pl * (BN / 16);
                const int nrow = 16 * rb + (lane >> 2);
                const int ch = (lane & 3) ^ swz_w16(nrow);
                const int voff = q < W_PIECES ? (pl * npad + n_blk + nrow) * p.Cin2 * 2 + ch * 16 : (int)0x80000000;
                dma16(rsrc_w2, wsm + (2 - k2) * WS_BYTES + q * 1024, voff, k2 * 64);
            }
        }
        const int row2 = p.in2_cstride * 4;
        const auto rsrc_in2 = rsrc_or_empty(p.in2, (int64_t)p.B * p.Hout * p.Wout * row2);
        for (int k2 = 0; k2 < nkb2; ++k2) {
            // fragments of the four classes' pixels: class (cy, cx) of class-grid pixel (y, x) = output pixel (2 y + cy, 2 x + cx)
            u4 raw[4][2][2];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int y = oy0 + wave, x = ox0 + 16 * b + (lane & 15);
                    const int oy = 2 * y + (c >> 1), ox = 2 * x + (c & 1);
                    const bool ok = y < p.Hm && x < p.Wm && oy < p.Hout && ox < p.Wout;
                    const int off = ok ? ((img * p.Hout + oy) * p.Wout + ox) * row2 + (p.in2_coff + k2 * 32 + q8 * 8) * 4 : (int)0x80000000;
                    raw[c][b][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in2, off, 0, 0);
                    raw[c][b][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in2, off, 16, 0);
                }
            if (k2 == 0) {   // (the weight stages of the second source have landed -- and the pipeline's first stages with them)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            const unsigned char* wc = wsm + (2 - k2) * WS_BYTES + w_addr_l;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bf16x8 pf[2][3];
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    split8(__builtin_bit_cast(f32x4, raw[c][b][0]), __builtin_bit_cast(f32x4, raw[c][b][1]), pf[b][0], pf[b][1], pf[b][2]);
                X6P_MFMA(c, wc)
            }
        }
    }

    // ---- main loop: channel block, window position, class
    // fragments of the wave's two 16-pixel blocks at window position (a, b) of patch buffer `pbuf`: raw fp32 (LDS reads in flight) ...
    auto frag_load = [&](const unsigned char* pbuf, const int ps, f32x4 (&raw)[2][2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int q = (wave + (ps >> 1)) * PW + 16 * b + (ps & 1) + (lane & 15);
            const unsigned char* px = pbuf + q * 128;
            const int sw = swz_p(q);
            // (odd chunk pairs read their chunks in the opposite order: see swz_p)
            raw[b][0] = *reinterpret_cast<const f32x4*>(px + (((2 * q8 + (q8 & 1)) ^ sw) << 4));
            raw[b][1] = *reinterpret_cast<const f32x4*>(px + (((2 * q8 + 1 - (q8 & 1)) ^ sw) << 4));
        }
    };
    // ... and their split into the three bf16 planes
    auto frag_split = [&](const f32x4 (&raw)[2][2], bf16x8 (&pf)[2][3]) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 lo = (q8 & 1) ? raw[b][1] : raw[b][0], hi = (q8 & 1) ? raw[b][0] : raw[b][1];
            split8(lo, hi, pf[b][0], pf[b][1], pf[b][2]);
        }
    };
#ifdef SPAA_X6P_STAMP
    const unsigned long long ts1 = X6P_T();
#endif
    int st = 0, step = 0;
    bf16x8 pfs[2][2][3];   // fragments of the current / the next window position
    f32x4 raw[2][2];
    for (int kb = 0; kb < nkb; ++kb) {
        const unsigned char* pb = smem + (kb & 1) * PATCH_BYTES;
        const unsigned char* pb_next = smem + ((kb + 1) & 1) * PATCH_BYTES;
        int nn = 0;   // (STD: the combo index, a compile-time constant in every expanded body)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if ((STD ? STD_TIX[c][ps] : tix[c][ps]) < 0) continue;   // (uniform; STD: resolved at compile time)
                const int n = STD ? nn : step - kb * ncombo;
                ++nn;
                // this wave's pieces of the step's weights (and, at a block's first combo, of its patch) have landed.  Loads
                // complete in order; issued AFTER this step's weights (two steps ago) are the next combo's pieces and -- on the two
                // combos that follow a block's first, where the next block's patch was requested right after weights(n + 2) --
                // that patch: combo 1 waits for weights issued before it, combo 2 for weights(2) issued just before it; from
                // combo 3 on the patch is OLDER than the weights waited for
#ifdef SPAA_X6P_STAMP
                const unsigned long long tw0 = X6P_T();
#endif
                // (STD: nine combos and three weight stages: the stage of combo n is n % 3, and every condition on `step` is one on
                // the channel block and the compile-time n)
                const bool last_step = STD ? (kb + 1 == nkb && n == 8) : step + 1 >= nsteps;
                const bool two_ahead = STD ? (kb + 1 < nkb || n < 7) : step + 2 < nsteps;
                const bool first_step = STD ? (kb == 0 && n == 0) : step == 0;
                if (STD) st = n % 3;
                if (last_step) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if ((n == 1 || n == 2) && kb + 1 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW + PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPW) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef SPAA_X6P_STAMP
                const unsigned long long tw1 = X6P_T();
#endif
                if (!X6P_ABL(2)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#ifdef SPAA_X6P_STAMP
                {
                    const unsigned long long tw2 = X6P_T();
                    ts_wait += tw2 - tw1;          // barrier
                    ts_first += tw1 - tw0;         // vmcnt / lgkmcnt
                }
#endif
                if constexpr (STD) {
                    if (first_step && !w1_issued && nsteps > 1) dma_w_std(1, 0, 1);   // (two blocks of second-source weights held its stage)
                    if (two_ahead && !X6P_ABL(0)) dma_w_std((n + 2) % 3, n + 2 >= 9 ? kb + 1 : kb, (n + 2) % 9);
                } else {
                    if (step == 0 && !w1_issued && nsteps > 1) dma_w(1, 1);   // (two blocks of second-source weights held its stage)
                    if (step + 2 < nsteps && !X6P_ABL(0)) dma_w(st >= 1 ? st - 1 : 2, step + 2);
                }
                if (n == 0 && kb + 1 < nkb && !X6P_ABL(0)) dma_patch((kb + 1) & 1, kb + 1);
                if (first_step) {   // the very first fragments: nothing to hide them under
                    frag_load(pb, 0, raw);
                    frag_split(raw, pfs[0]);
                }
                const bool fetch = c == 3 && (ps < 3 || kb + 1 < nkb);   // class 3: the position's last combo
                if (fetch && !X6P_ABL(5)) frag_load(ps < 3 ? pb : pb_next, (ps + 1) & 3, raw);
                const unsigned char* wc = wsm + st * WS_BYTES + w_addr_l;
                if (fetch && !X6P_ABL(1)) frag_split(raw, pfs[(ps + 1) & 1]);   // (scheduled into the shadows of the MFMAs below)
                if (!X6P_ABL(3)) {
                    bf16x8 (&pf)[2][3] = pfs[ps & 1];
                    X6P_MFMA(c, wc)
                }
                st = st == 2 ? 0 : st + 1;
                ++step;
            }
        }
    }
#undef X6P_MFMA

    // ---- epilogue.  D layout of a 16x16 block: column (lane & 15) = class-grid pixel, rows 4 (lane >> 4) + e = 4 consecutive
    // channels.  Through a wave-private LDS region in OUTPUT-pixel order: pass (cy, b) = output row 2 y + cy, output pixels
    // 2 (ox0 + 16 b) .. + 31 (the classes (cy, 0) and (cy, 1) interleaved): a store instruction writes whole BN-channel rows of
    // consecutive pixels
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    constexpr int ROWB = BN * 4 + 16;                  // (+16: the 16 pixels of a fragment write to distinct banks)
    constexpr int LPP = BN / 4, PPI = 64 / LPP;        // lanes per pixel, pixels per instruction
#ifdef SPAA_X6P_STAMP
    const unsigned long long ts2 = X6P_T();
    if (X6P_ABL(4)) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < TJ; ++j) asm volatile("" ::"v"(acc[c][b][j]));
        return;
    }
#endif
    __syncthreads();
    unsigned char* const eb = smem + wave * (32 * ROWB);
    const int ch = 4 * (lane % LPP);
    const int n = n_blk + ch;
    const bool n_ok = n < p.Cout;
    const int y = oy0 + wave;
    const bool fast = fast_epi_ok(p, vec);
    const fast_epi_t fe = make_fast_epi(p, n_ok ? n : 0);
#pragma unroll
    for (int cy = 0; cy < 2; ++cy) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int cx = 0; cx < 2; ++cx)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    *reinterpret_cast<f32x4*>(eb + (2 * (lane & 15) + cx) * ROWB + (16 * j + 4 * q8) * 4) = acc[2 * cy + cx][b][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int oy = 2 * y + cy, oxb = 2 * (ox0 + 16 * b);
            const bool row_ok = y < p.Hm && oy < p.Hout;
            const int orow = (img * p.Hout + oy) * p.Wout + oxb;
            if (fast) {
                constexpr int EB = 32 / PPI < 8 ? 32 / PPI : 8;
#pragma unroll 1
                for (int i0 = 0; i0 < 32 / PPI; i0 += EB) {
                    fast_pre_t<float> pre[EB];
#pragma unroll
                    for (int i = 0; i < EB; ++i) {
                        const int pr = (i0 + i) * PPI + lane / LPP;
                        pre[i] = fast_epi_load<float>(fe, p, orow + pr, n, row_ok && n_ok && oxb + pr < p.Wout && ox0 + 16 * b + (pr >> 1) < p.Wm);
                    }
#pragma unroll
                    for (int i = 0; i < EB; ++i) {
                        const int pr = (i0 + i) * PPI + lane / LPP;
                        const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);
                        fast_epi_store<float>(fe, p, orow + pr, n, row_ok && n_ok && oxb + pr < p.Wout && ox0 + 16 * b + (pr >> 1) < p.Wm, a, pre[i]);
                    }
                }
            } else {
                for (int i = 0; i < 32 / PPI; ++i) {
                    const int pr = i * PPI + lane / LPP;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);
                    float v[4] = {a[0], a[1], a[2], a[3]};
                    if (row_ok && oxb + pr < p.Wout && ox0 + 16 * b + (pr >> 1) < p.Wm) store4_t<float>(p, (size_t)(orow + pr), n, v, vec);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
#ifdef SPAA_X6P_STAMP
    if (stamp_on && g_x6p_stamps != nullptr && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the stores have left: the epilogue's true end)
        const unsigned long long ts3 = X6P_T(), tr3 = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = g_x6p_stamps + ((size_t)blockIdx.x * NW + wave) * 8;
        o[0] = ts0, o[1] = ts1, o[2] = ts2, o[3] = ts3, o[4] = tr0, o[5] = tr3, o[6] = ts_wait, o[7] = ts_first;
    }
#endif
}

}  // namespace

#ifdef SPAA_X6P_STAMP
// diagnostic build only: where the stamps of the NEXT launches go (device buffer of nwg * 4 * 8 uint64) or nullptr
extern "C" int spaa_x6p_set_stamp_buffer(void* buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_x6p_stamps), &buf, sizeof(buf));
}
#endif

// called by spaa_tapconv_f32 (tapconv.hip) for tile 74 after the common shape checks: FOUR output-parity classes in row-major
// order ((0,0), (0,1), (1,0), (1,1)), s_in = 1, s_out = 2, every tap inside one 2 x 2 window, fp32 storage, Cin % 32 == 0, the class
// grid = the input grid.  Optional second source: in2 [B, Hout, Wout, in2_cstride] (channels in2_coff .. + Cin2, Cin2 = 32 or 64) through
// the 1 x 1 weights w2_split ([3][Npad][Cin2] bf16 planes), added before bias / residual / activation.
int spaa_launch_tapconv_x6p(const spaa_tapconv_t& d, hipStream_t stream) {
    if (d.w_split == nullptr || (d.Cin % 32) != 0 || d.Cin < 32 || d.nclass != 4 || d.s_in != 1 || d.s_out != 2 || d.nfold > 1 ||
        d.ksplit > 1 || d.ksplit < 0 || d.io_dtype != 0 || d.Hm != d.Hin || d.Wm != d.Win || d.Hm != (d.Hout + 1) / 2 ||
        d.Wm != (d.Wout + 1) / 2)
        return hipErrorInvalidValue;
    if (d.tap_range[1] - d.tap_range[0] > 1 || d.tap_range[3] - d.tap_range[2] > 1) return hipErrorInvalidValue;
    int64_t woff = 0;
    const int npad = (d.Cout + 127) & ~127;
    for (int c = 0; c < 4; ++c) {
        const spaa_tapclass_t& cl = d.cls[c];
        if (cl.oy0 != (c >> 1) || cl.ox0 != (c & 1) || cl.ntaps < 1 || cl.ntaps > 4 || cl.K != cl.ntaps * d.Cin || cl.Kpad != cl.K ||
            cl.w_off != woff || (c == 3 && cl.ntaps != 4))   // (class (1, 1): a tap at every window position)
            return hipErrorInvalidValue;
        woff += (int64_t)npad * cl.Kpad;
    }
    if (woff * 6 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    if ((int64_t)d.B * d.Hin * d.Win * d.in_cstride * 4 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    if (d.in2 != nullptr) {
        if (d.w2_split == nullptr || (d.Cin2 != 32 && d.Cin2 != 64) || (d.in2_cstride & 3) || (d.in2_coff & 3) || d.in2_coff + d.Cin2 > d.in2_cstride ||
            (int64_t)d.B * d.Hout * d.Wout * d.in2_cstride * 4 >= (int64_t)1 << 31)
            return hipErrorInvalidValue;
    }
    const int wg_y = (d.Hm + RY - 1) / RY, wg_x = (d.Wm + RX - 1) / RX;
    const int BN = d.Cout <= 32 ? 32 : 64;
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const int64_t nwg = (int64_t)d.B * wg_y * wg_x * n_tiles;
    if (nwg > 0x7fffffff) return hipErrorInvalidValue;
    static bool attr_set[4][SPAA_MAX_DEVICES] = {};
#define X6P_LAUNCH_T(N, T, SLOT)                                                                                           \
    {                                                                                                                      \
        /* (main loop: two patch buffers + three weight stages; epilogue: NW x 32 rows of N * 4 + 16 bytes) */              \
        const size_t smem = (size_t)W_OFF + 3 * (size_t)(((3 * N / 16 + NW - 1) / NW) * NW * 1024);                        \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&x6p_kernel<N, T>), (int)smem, attr_set[SLOT]);    \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((x6p_kernel<N, T>), dim3((unsigned)nwg), dim3(64 * NW), smem, stream, d, wg_y, wg_x, n_tiles);      \
    }
    // canonical (class, tap) structure (reserved2 bit 0: the caller has checked the tap lists; here: what the descriptor shows of it)
    const bool std_ok = (d.reserved2 & 1) && d.cls[0].ntaps == 1 && d.cls[1].ntaps == 2 && d.cls[2].ntaps == 2 && d.cls[3].ntaps == 4 &&
                        d.tap_range[0] == 0 && d.tap_range[1] == 1 && d.tap_range[2] == 0 && d.tap_range[3] == 1;
    if (std_ok) {
        if (BN == 32) X6P_LAUNCH_T(32, true, 2) else X6P_LAUNCH_T(64, true, 3)
    } else {
        if (BN == 32) X6P_LAUNCH_T(32, false, 0) else X6P_LAUNCH_T(64, false, 1)
    }
#undef X6P_LAUNCH_T
    return (int)hipGetLastError();
}
