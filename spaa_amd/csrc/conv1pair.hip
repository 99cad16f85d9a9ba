// conv1pair.hip — the two stride-2 entry layers of ShadingNetSPAA with use_rough in ONE launch:
//     S1 = relu(conv1_s(cat[s, xw * s]))          (models.py:284-285 of the reference: res1_s)
//     X1 = relu(conv1(xw) + S1)                   (models.py:295)
// Both layers read the same 3-channel camera-resolution images and write 32 channels at half resolution, so they are
// HBM-bound on their outputs.  Run separately they read xw twice, the 8-channel concatenation [s, xw * s] (which the
// warp kernel had to write first) and S1 back as conv1's residual; fused, a workgroup stages the xw and s patches of
// its 32 x 8 output tile once (LDS-DMA, zero padding as out-of-range offsets), forms xw * s in registers, and keeps
// S1 in registers for conv1's epilogue.  Per camera pixel: 32 B read, and per output pixel 2 x 128 B + 16 mask bytes
// written — 410 MB at batch 64 instead of 870 MB (+ the 134 MB the warp kernel no longer writes).
//
// Arithmetic is the exact fp32 matrix instruction of smallcin.hip (v_mfma_f32_32x32x2_f32), weights in registers:
// per tap 2 MFMAs for conv1 and 4 for conv1_s (s quad, xw * s quad).
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/spaa_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TW = 32, TH = 8;                 // output pixels per workgroup (4 waves x 2 rows of 32)
constexpr int PH = 2 * (TH - 1) + 3;           // 17 input rows
constexpr int PW = 2 * (TW - 1) + 3;           // 65 input columns
constexpr int NPIX = PH * PW;                  // 1105 staged pixels of 16 B per source
constexpr int NPIECE = (NPIX + 63) / 64;       // 18 one-KiB DMA pieces per source
constexpr int SRC_BYTES = NPIECE * 1024;
constexpr int LDS_BYTES = 2 * SRC_BYTES;       // 36 KiB
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct c1p_args {
    const float* xw;
    const float* s;
    const float* w;      // [3 groups: conv1 | conv1_s on s | conv1_s on xw*s][32 n][9 taps][4 channels]
    const float* b1;
    const float* bs;
    void* S1;
    void* X1;
    uint8_t* mS1;
    uint8_t* mX1;
    int B, H, W, Hm, Wm, tiles_x, tiles_y;
};

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* ptr, uint32_t bytes) {
    const uint64_t addr = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* base, const size_t byte_off, const int bytes) {
    const uint64_t addr = reinterpret_cast<uint64_t>(base) + byte_off;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// nibble q of u (q = 0..3) -> byte q: the gate format is one byte per 4 channels
__device__ __forceinline__ uint32_t spread_nibbles(const uint32_t u) {
    return (u & 0xFu) | ((u & 0xF0u) << 4) | ((u & 0xF00u) << 8) | ((u & 0xF000u) << 12);
}
// lane `l` of `old` := the wave-uniform value `v` (v_writelane_b32 through the compiler, which then keeps the wait states between
// the compare that produces `v` and this read of it; clang has no builtin of that name for the intrinsic)
extern "C" __device__ uint32_t spaa_llvm_writelane(uint32_t v, uint32_t l, uint32_t old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t writelane(const uint32_t v, const int l, const uint32_t old) { return spaa_llvm_writelane(v, (uint32_t)l, old); }
template <typename T>
__device__ __forceinline__ void st1(const __amdgpu_buffer_rsrc_t r, const int voff, const int imm, const float v) {
    if constexpr (sizeof(T) == 2)
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), r, voff, imm, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, imm, 0);
}

// T = storage type of S1 / X1 (fp32, or fp16 in the fp16-storage mode: the residual added is then the ROUNDED S1, as
// when conv1 reads it back from memory)
template <typename T>
__global__ __launch_bounds__(256, 2) void conv1pair_kernel(const c1p_args p) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    int tile;
    {   // consecutive tiles on the same XCD (workgroups are dealt round-robin over the 8 XCDs): neighbours share halo rows in L2
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tx = tile % p.tiles_x;
    const int ty = (tile / p.tiles_x) % p.tiles_y;
    const int b = tile / (p.tiles_x * p.tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- stage the two input patches: rows 2 y0 - 1 .., columns 2 x0 - 1 .. (one 16-byte pixel per lane)
    {
        const uint32_t bytes = (uint32_t)p.B * (uint32_t)(p.H * p.W) * 16u;
        const auto rx = make_rsrc(p.xw, bytes), rs = make_rsrc(p.s, bytes);
        for (int i = wave; i < NPIECE; i += 4) {
            const int q = i * 64 + lane;
            const int py = q / PW, px = q - py * PW;
            const int iy = 2 * y0 - 1 + py, ix = 2 * x0 - 1 + px;
            const bool v = q < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int off = v ? ((b * p.H + iy) * p.W + ix) * 16 : (int)0x80000000;
            dma16(rx, smem + i * 1024, off);
            dma16(rs, smem + SRC_BYTES + i * 1024, off);
        }
    }

    // ---- weights in registers: lane -> output channel (lane & 31), channels 2 * (lane >> 5) + {0, 1} of every tap
    float wA[3][9], wB[3][9];
    {
        const float* wr = p.w + (size_t)(lane & 31) * 36 + 2 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f2 w2 = *reinterpret_cast<const f2*>(wr + g * (32 * 36) + t * 4);
                wA[g][t] = w2.x;
                wB[g][t] = w2.y;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lx = lane & 31, h = lane >> 5, n = lane & 31;
    const float b1n = p.b1[n], bsn = p.bs[n];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ly = wave * 2 + r;
        const unsigned char* pp = smem + ((2 * ly) * PW + 2 * lx) * 16 + h * 8;
        f32x16 acc1, acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[i] = acc2[i] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = ((t / 3) * PW + (t % 3)) * 16;
            const f2 vx = *reinterpret_cast<const f2*>(pp + toff);
            const f2 vs = *reinterpret_cast<const f2*>(pp + SRC_BYTES + toff);
            const f2 vxs = vx * vs;
            // pixels as the ROWS of the product (A operand), channels as its columns: a lane then holds ONE channel of 16 pixels
            // and a store instruction writes two complete 128-byte channel rows
            acc1 = mfma2(vx.x, wA[0][t], acc1);
            acc2 = mfma2(vs.x, wA[1][t], acc2);
            acc1 = mfma2(vx.y, wB[0][t], acc1);
            acc2 = mfma2(vs.y, wB[1][t], acc2);
            acc2 = mfma2(vxs.x, wA[2][t], acc2);
            acc2 = mfma2(vxs.y, wB[2][t], acc2);
        }
        // ---- epilogue: lane = channel n (both halves), element 4 g + e = pixel 8 g + 4 h + e of the row.  Out-of-image pixels fall
        // outside the row's buffer descriptor (its records end at the last valid pixel) and are dropped by the hardware.
        const int y = y0 + ly;
        const int npx = y < p.Hm ? (p.Wm - x0 < TW ? p.Wm - x0 : TW) : 0;
        const size_t o0 = ((size_t)b * p.Hm + (y < p.Hm ? y : 0)) * p.Wm + x0;
        const auto rS = row_rsrc(p.S1, o0 * 32 * sizeof(T), npx * 32 * (int)sizeof(T));
        const auto rX = row_rsrc(p.X1, o0 * 32 * sizeof(T), npx * 32 * (int)sizeof(T));
        const int voff = (h * 4 * 32 + n) * (int)sizeof(T);
        uint32_t gS = 0, gX = 0;      // lane j < 32: the 32 gate bits of pixel j
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = fmaxf(acc2[4 * g + e] + bsn, 0.f);
                if constexpr (sizeof(T) == 2) a = (float)(_Float16)a;
                float v = fmaxf(acc1[4 * g + e] + b1n + a, 0.f);
                if constexpr (sizeof(T) == 2) v = (float)(_Float16)v;
                st1<T>(rS, voff, (8 * g + e) * 32 * (int)sizeof(T), a);
                st1<T>(rX, voff, (8 * g + e) * 32 * (int)sizeof(T), v);
                const uint64_t ba = __builtin_amdgcn_ballot_w64(a > 0.f), bv = __builtin_amdgcn_ballot_w64(v > 0.f);
                gS = writelane((uint32_t)ba, 8 * g + e, gS);
                gS = writelane((uint32_t)(ba >> 32), 8 * g + 4 + e, gS);
                gX = writelane((uint32_t)bv, 8 * g + e, gX);
                gX = writelane((uint32_t)(bv >> 32), 8 * g + 4 + e, gX);
            }
        const int moff = lane < 32 ? lane * 8 : (int)0x80000000;
        if (p.mS1 != nullptr) {
            const auto rm = row_rsrc(p.mS1, o0 * 8, npx * 8);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{spread_nibbles(gS & 0xFFFFu), spread_nibbles(gS >> 16)}, rm, moff, 0, 0);
        }
        if (p.mX1 != nullptr) {
            const auto rm = row_rsrc(p.mX1, o0 * 8, npx * 8);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{spread_nibbles(gX & 0xFFFFu), spread_nibbles(gX >> 16)}, rm, moff, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the same pair on the 16 x 16 x 32 matrix instructions.  The 27 products (9 taps x 3 channels) of an output pixel are ONE
// K = 32 step: k = 8 g + j holds tap 2 g (j < 3) and tap 2 g + 1 (4 <= j < 7); the pad lanes of the 4-float pixels (k = 3, 7, 11) carry
// tap 8's three channels.  Per 16 pixels: conv1 = 2 products (two blocks of 16 output channels), conv1_s = 4 (s step, xw * s step).
//   F16 (fp16-STORAGE mode): "fp16 operands like every other layer of the mode" -- xw, s and xw * s (formed in fp32) rounded to fp16 in
//       registers, the weights once per workgroup, one v_mfma_f32_16x16x32_f16 per product: 6 MFMAs per 16 pixels;
//   fp32: both operands split exactly into three bf16 planes, six of the nine partial products (the arithmetic of every bf16x6 kernel
//       of this library): 36 v_mfma_f32_16x16x32_bf16 per 16 pixels
// instead of 54 v_mfma_f32_32x32x2_f32 of four times the length (46 us of matrix time on 5.4 GFLOP in the kernel above).  The epilogue
// writes a lane's four channels of a pixel as one 8- / 16-byte store and one gate byte.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int cvt2(float a, float b) {
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned int p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned int p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
// 8 fp32 -> three bf16x8 with x == h + m + l exactly
__device__ __forceinline__ void split8(const f4 a, const f4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
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
// six of the nine partial products of (w0 + w1 + w2) . (p0 + p1 + p2), small terms first
__device__ __forceinline__ f32x4 mfma6(const bf16x8 (&w)[3], const bf16x8 (&q)[3], f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], q[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], q[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], q[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], q[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], q[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], q[0], acc, 0, 0, 0);
    return acc;
}

template <bool F16>
__global__ __launch_bounds__(256, 2) void conv1pair_mfma_kernel(const c1p_args p) {
    typedef typename std::conditional<F16, _Float16, float>::type T;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tx = tile % p.tiles_x;
    const int ty = (tile / p.tiles_x) % p.tiles_y;
    const int b = tile / (p.tiles_x * p.tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;
    {   // the two input patches, as the kernel above stages them
        const uint32_t bytes = (uint32_t)p.B * (uint32_t)(p.H * p.W) * 16u;
        const auto rx = make_rsrc(p.xw, bytes), rs = make_rsrc(p.s, bytes);
        for (int i = wave; i < NPIECE; i += 4) {
            const int q = i * 64 + lane;
            const int py = q / PW, px = q - py * PW;
            const int iy = 2 * y0 - 1 + py, ix = 2 * x0 - 1 + px;
            const bool v = q < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const int off = v ? ((b * p.H + iy) * p.W + ix) * 16 : (int)0x80000000;
            dma16(rx, smem + i * 1024, off);
            dma16(rs, smem + SRC_BYTES + i * 1024, off);
        }
    }
    // weights as A operands: row = output channel 16 nb + (lane & 15), k = 8 (lane >> 4) + j as above; group 0 conv1, 1 conv1_s on s,
    // 2 conv1_s on xw * s  (p.w: [3][32 n][9 taps][4 channels] fp32)
    const int r16 = lane & 15, g = lane >> 4;
    h8 A[3][2];
    bf16x8 Ab[3][2][3];
#pragma unroll
    for (int grp = 0; grp < 3; ++grp)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float* wr = p.w + (size_t)grp * (32 * 36) + (size_t)(16 * nb + r16) * 36;
            float wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int tap = 2 * g + (j >> 2), c = j & 3;
                if (c == 3) {
                    const int k = 8 * g + j;
                    tap = 8;
                    c = k == 3 ? 0 : (k == 7 ? 1 : (k == 11 ? 2 : 3));
                }
                wv[j] = (c < 3 && tap < 9) ? wr[tap * 4 + c] : 0.f;
            }
            if constexpr (F16) {
#pragma unroll
                for (int j = 0; j < 8; ++j) A[grp][nb][j] = (_Float16)wv[j];
            } else {
                split8(f4{wv[0], wv[1], wv[2], wv[3]}, f4{wv[4], wv[5], wv[6], wv[7]}, Ab[grp][nb][0], Ab[grp][nb][1], Ab[grp][nb][2]);
            }
        }
    f32x4 b1q[2], bsq[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        b1q[nb] = *reinterpret_cast<const f32x4*>(p.b1 + 16 * nb + 4 * g);
        bsq[nb] = *reinterpret_cast<const f32x4*>(p.bs + 16 * nb + 4 * g);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int t0 = 2 * g, t1 = 2 * g + 1;
    const int o0 = ((t0 / 3) * PW + t0 % 3) * 16, o1 = ((t1 / 3) * PW + t1 % 3) * 16, o8 = (2 * PW + 2) * 16;
#pragma unroll 1
    for (int grp = 0; grp < 4; ++grp) {       // 2 rows x 2 groups of 16 output pixels per wave
        const int ly = wave * 2 + (grp >> 1), lx = 16 * (grp & 1) + r16;
        const unsigned char* pp = smem + ((2 * ly) * PW + 2 * lx) * 16;
        f4 x0v = *reinterpret_cast<const f4*>(pp + o0), x1v = *reinterpret_cast<const f4*>(pp + o1);
        const f4 x8v = *reinterpret_cast<const f4*>(pp + o8);
        f4 s0v = *reinterpret_cast<const f4*>(pp + SRC_BYTES + o0), s1v = *reinterpret_cast<const f4*>(pp + SRC_BYTES + o1);
        const f4 s8v = *reinterpret_cast<const f4*>(pp + SRC_BYTES + o8);
        x0v[3] = g == 0 ? x8v[0] : (g == 1 ? x8v[2] : 0.f);
        x1v[3] = g == 0 ? x8v[1] : 0.f;
        s0v[3] = g == 0 ? s8v[0] : (g == 1 ? s8v[2] : 0.f);
        s1v[3] = g == 0 ? s8v[1] : 0.f;
        const f4 p0v = x0v * s0v, p1v = x1v * s1v;      // xw * s in fp32 (models.py:342)
        h8 X, S, XS;
        bf16x8 Xb[3], Sb[3], XSb[3];
        if constexpr (F16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                X[e] = (_Float16)x0v[e], X[4 + e] = (_Float16)x1v[e];
                S[e] = (_Float16)s0v[e], S[4 + e] = (_Float16)s1v[e];
                XS[e] = (_Float16)p0v[e], XS[4 + e] = (_Float16)p1v[e];
            }
        } else {
            split8(x0v, x1v, Xb[0], Xb[1], Xb[2]);
            split8(s0v, s1v, Sb[0], Sb[1], Sb[2]);
            split8(p0v, p1v, XSb[0], XSb[1], XSb[2]);
        }
        const int y = y0 + ly, x = x0 + lx;
        const bool ok = y < p.Hm && x < p.Wm;
        const size_t o = ((size_t)b * p.Hm + (ok ? y : 0)) * p.Wm + (ok ? x : 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (F16) {
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0][nb], X, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1][nb], S, a2, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[2][nb], XS, a2, 0, 0, 0);
            } else {
                a1 = mfma6(Ab[0][nb], Xb, a1);
                a2 = mfma6(Ab[1][nb], Sb, a2);
                a2 = mfma6(Ab[2][nb], XSb, a2);
            }
            // D: column = pixel, rows = channels 16 nb + 4 g + e
            float sv[4], xv[4];
            unsigned int ms = 0, mx = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = fmaxf(a2[e] + bsq[nb][e], 0.f);
                if constexpr (F16) a = (float)(_Float16)a;      // (the residual added is the ROUNDED S1, as when conv1 reads it back)
                float v = fmaxf(a1[e] + b1q[nb][e] + a, 0.f);
                if constexpr (F16) v = (float)(_Float16)v;
                sv[e] = a, xv[e] = v;
                ms |= (a > 0.f ? 1u : 0u) << e;
                mx |= (v > 0.f ? 1u : 0u) << e;
            }
            if (ok) {
                const size_t oc = o * 32 + 16 * nb + 4 * g;
                if constexpr (F16) {
                    *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(p.S1) + oc) = h4{(_Float16)sv[0], (_Float16)sv[1], (_Float16)sv[2], (_Float16)sv[3]};
                    *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(p.X1) + oc) = h4{(_Float16)xv[0], (_Float16)xv[1], (_Float16)xv[2], (_Float16)xv[3]};
                } else {
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.S1) + oc) = f32x4{sv[0], sv[1], sv[2], sv[3]};
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.X1) + oc) = f32x4{xv[0], xv[1], xv[2], xv[3]};
                }
                if (p.mS1 != nullptr) p.mS1[o * 8 + 4 * nb + g] = (uint8_t)ms;
                if (p.mX1 != nullptr) p.mX1[o * 8 + 4 * nb + g] = (uint8_t)mx;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The ADJOINT of the pair in fp16-storage mode (round 6): the input gradients of conv1 and conv1_s meet at the warped image,
//     g_xw = conv1^T(gX1) + s * conv1_s^T(gS1)[rough channels]          (models.py:284-285,295,342 of the reference under autograd)
// until now two thin-output launches with a 67 MB round trip of the rough part between them (402 MB per step at batch 64).  Here a wave
// owns an INPUT row segment of 16 half-resolution pixels; they feed the 2 x 2 output pixels (2 y + cy, 2 x + cx) through at most four
// operands -- I00 = in[y][x], I01 = in[y][x + 1], I10 = in[y + 1][x], I11 = in[y + 1][x + 1] -- and with the MFMA's rows = (parity class,
// channel) = 4 (2 cy + cx) + c ONE v_mfma_f32_16x16x32_f16 per operand (K = the 32 channels) adds that operand's taps to all four classes:
// class (cy, cx) takes operand I_rq iff r <= cy and q <= cx, through tap ky = (cy == 0 ? 1 : r == 0 ? 2 : 0), kx likewise.  Four MFMAs
// per source for 64 output pixels, and every lane ends with ONE output pixel's three channels of both sources: scene load, multiply-add,
// one 16-byte store.  The operands are 16-byte-per-lane global loads of whole 1 KB row segments (no LDS at all); the weights (fp16, as
// the separate launches round them) live in registers.  HBM: read gX1, gS1 (fp16) and the scene, write g_xw: 268 MB.
struct c1b_args {
    const _Float16* gx1;
    const _Float16* gs1;
    const float* scene;
    const _Float16* wimg;   // [2 sources][4 operands][64 lanes][8] fp16: the MFMA A operands (see the kernel)
    float* g_xw;
    int B, H2, W2, ngrp;   // half-resolution size; 16-pixel column groups per row
};

__global__ __launch_bounds__(256, 4) void conv1pair_bwd_h16_kernel(const c1b_args p) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15, g = lane >> 4;
    // A operands: row = 4 cl + c (cl = 2 cy + cx), k = input channel n = 8 g + e; operand (r, q); source 0 = conv1, 1 = conv1_s rough:
    // the per-lane image [source][operand][lane][8 fp16] is packed once per model by the host (spaa_amd/models.py: pack_pair1_bwd)
    h8 A[2][4];
#pragma unroll
    for (int src = 0; src < 2; ++src)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) A[src][rq] = *reinterpret_cast<const h8*>(p.wimg + ((src * 4 + rq) * 64 + lane) * 8);
    const int nrows = p.B * p.H2 * p.ngrp;               // wave tasks: (image, input row, column group)
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int H = 2 * p.H2, W = 2 * p.W2;
    for (int t = wid; t < nrows; t += nw) {
        const int grp = t % p.ngrp, y = (t / p.ngrp) % p.H2, b = t / (p.ngrp * p.H2);
        const int x = 16 * grp + j;
        const bool x0ok = x < p.W2, x1ok = x + 1 < p.W2, y1ok = y + 1 < p.H2;
        const size_t row0 = ((size_t)b * p.H2 + y) * p.W2, row1 = row0 + p.W2;
        const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        h8 I[2][4];
#pragma unroll
        for (int src = 0; src < 2; ++src) {
            const _Float16* base = src == 0 ? p.gx1 : p.gs1;
            I[src][0] = x0ok ? *reinterpret_cast<const h8*>(base + (row0 + x) * 32 + 8 * g) : z;
            I[src][1] = x1ok ? *reinterpret_cast<const h8*>(base + (row0 + x + 1) * 32 + 8 * g) : z;
            I[src][2] = (x0ok && y1ok) ? *reinterpret_cast<const h8*>(base + (row1 + x) * 32 + 8 * g) : z;
            I[src][3] = (x1ok && y1ok) ? *reinterpret_cast<const h8*>(base + (row1 + x + 1) * 32 + 8 * g) : z;
        }
        // this lane's output pixel: class g of input pixel (y, x)
        const int Y = 2 * y + (g >> 1), X = 2 * x + (g & 1);
        const size_t o = ((size_t)b * H + Y) * W + X;
        f32x4 sv = {0.f, 0.f, 0.f, 0.f};
        if (x0ok) sv = *reinterpret_cast<const f32x4*>(p.scene + o * 4);
        f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0][rq], I[0][rq], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1][rq], I[1][rq], a2, 0, 0, 0);
        }
        // D: column = input pixel j, rows 4 g + e = (class g, channel e)
        if (x0ok) *reinterpret_cast<f32x4*>(p.g_xw + o * 4) = f32x4{a1[0] + a2[0] * sv[0], a1[1] + a2[1] * sv[1], a1[2] + a2[2] * sv[2], 0.f};
    }
}

}  // namespace

extern "C" int spaa_conv1_pair_bwd_f16(const void* g_x1, const void* g_s1, const float* scene, const void* w_image, float* g_xw, int B, int H,
                                       int W, spaa_stream_t stream) {
    if (!g_x1 || !g_s1 || !scene || !w_image || !g_xw || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return hipErrorInvalidValue;
    if ((uint64_t)B * H * W * 16u >= (1ull << 40)) return hipErrorInvalidValue;
    c1b_args a;
    a.gx1 = reinterpret_cast<const _Float16*>(g_x1), a.gs1 = reinterpret_cast<const _Float16*>(g_s1);
    a.scene = scene, a.wimg = reinterpret_cast<const _Float16*>(w_image), a.g_xw = g_xw;
    a.B = B, a.H2 = H / 2, a.W2 = W / 2, a.ngrp = (a.W2 + 15) / 16;
    const int64_t tasks = (int64_t)B * a.H2 * a.ngrp;
    if (tasks > 0x7fffffff) return hipErrorInvalidValue;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    int64_t nwg = (tasks + 3) / 4;                        // four waves per workgroup, a task per wave and round
    if (nwg > 4 * (int64_t)ncu) nwg = 4 * (int64_t)ncu;   // persistent: four workgroups per compute unit walk the tasks
    hipLaunchKernelGGL(conv1pair_bwd_h16_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

extern "C" int spaa_conv1_pair_fwd(const float* xw, const float* s, const float* w_pair, const float* bias1, const float* bias_s,
                                   void* S1, void* X1, uint8_t* mask_S1, uint8_t* mask_X1, int B, int H, int W, int out_f16,
                                   spaa_stream_t stream) {
    if (!xw || !s || !w_pair || !bias1 || !bias_s || !S1 || !X1 || B < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return hipErrorInvalidValue;
    if ((uint64_t)B * H * W * 16u >= (1ull << 31)) return hipErrorInvalidValue;   // 32-bit buffer offsets
    c1p_args a;
    a.xw = xw, a.s = s, a.w = w_pair, a.b1 = bias1, a.bs = bias_s;
    a.S1 = S1, a.X1 = X1, a.mS1 = mask_S1, a.mX1 = mask_X1;
    a.B = B, a.H = H, a.W = W, a.Hm = H / 2, a.Wm = W / 2;
    a.tiles_x = (a.Wm + TW - 1) / TW, a.tiles_y = (a.Hm + TH - 1) / TH;
    const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * B), 1, 1);
    hipStream_t st = (hipStream_t)stream;
    // SPAA_C1P_MFMA: 1 (default) fp16 storage on the 16 x 16 x 32 fp16 form of round 6 (100 -> 79 us at batch 64, 256 x 256), fp32 on the
    // fp32-MFMA form; 2: fp32 on the bf16x6 16 x 16 x 32 form too (measured 119 against 116 us: three 44-instruction operand splits
    // per 16 pixels cost what the shorter matrix instructions save -- kept for A/B runs and its test); 0: the fp32-MFMA form everywhere
    static const int mfma = []() { const char* e = getenv("SPAA_C1P_MFMA"); return e ? atoi(e) : 1; }();
    if (out_f16 && mfma >= 1)
        hipLaunchKernelGGL(conv1pair_mfma_kernel<true>, grid, dim3(256), LDS_BYTES, st, a);
    else if (!out_f16 && mfma >= 2)
        hipLaunchKernelGGL(conv1pair_mfma_kernel<false>, grid, dim3(256), LDS_BYTES, st, a);
    else if (out_f16)
        hipLaunchKernelGGL(conv1pair_kernel<_Float16>, grid, dim3(256), LDS_BYTES, st, a);
    else
        hipLaunchKernelGGL(conv1pair_kernel<float>, grid, dim3(256), LDS_BYTES, st, a);
    return (int)hipGetLastError();
}
