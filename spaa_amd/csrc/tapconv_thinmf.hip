// tapconv_thinmf.hip — THIN outputs (Cout <= 4: the image-side input gradients of conv1 / conv1_s and of the classifiers' first
// layers) on the matrix cores, with the output-parity classes folded into the GEMM's N dimension.
//
// The VALU kernel (thinpatch.hip) spends one 16-byte LDS read per six packed FMAs and is LDS-bandwidth bound: the ResNet stem's
// input gradient (64 -> 3, 7 x 7 / stride 2: 15.1 GFLOP) takes 354 us at 43 TFLOP/s.  A matrix-core tile is 16 wide and a thin
// layer has 3 outputs -- but a stride-2 layer's input gradient has FOUR output-parity classes that read the same input
// neighbourhood: with N = 4 classes x 4 channels = 16 rows (weight rows of taps a class does not have are zero) one MFMA tile
// produces the 2 x 2 output pixels of a class-grid position.  Stem: K = 16 taps x 64 channels, 77 % of the products are real.
//   * workgroup = 8 waves = 12 (fp32 input: 8) rows x 32 columns of the class grid; wave = 3 (2) rows x 16 columns;
//   * the input patch ((12 + TBH - 1) x (32 + TBW - 1) pixels of a 32-channel block; TBH x TBW <= 4 x 4 = the tap box) is staged
//     once by LDS-DMA (out-of-image = out-of-range offset = zeros), double-buffered over the channel blocks;
//   * K order: channel block, then tap COLUMN dx: a wave reads the 3 + TBH - 1 pixel-row fragments of that column once (fp32: two
//     16-byte reads and one exact 3-way bf16 split each) and uses each for up to TBH taps x 3 rows; the column's weights (TBH taps x
//     16 rows x 32 channels, 1 KiB per plane and tap, chunk swizzle baked in by the host) are LDS-DMA'd one step ahead;
//   * fp32 input: bf16x6 arithmetic (exact operands, six MFMAs per product: tapconv_x6d.hip); fp16 input (fp16-STORAGE mode): fp16
//     weights, one MFMA per product; fp32 accumulation and fp32 output either way;
//   * epilogue: a lane holds the 4 channels of ONE output pixel (class = lane >> 4): a 16-byte store, 512 contiguous bytes per
//     output row and wave; residual and multiplicative gate (the two forms the PCNet engine uses) branch-free, anything else through
//     the shared store4_t.
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
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
// 8 fp32 -> three bf16x8 with x == h + m + l exactly (tapconv_x6d.hip: split8)
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
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

constexpr int TW = 32;   // class-grid columns of a workgroup (rows: 4 RW, RW = rows of a wave: template parameter)
constexpr int NW = 8;

// HIN: fp16 input (one weight plane, one MFMA per product); TBH: rows of the tap box (2..4); RW: class-grid rows per wave (the
// workgroup covers 4 RW rows); PDB: the patch double-buffered over the channel blocks.  fp32 input runs RW = 2 with a SINGLE patch
// buffer (49 + 24 KiB: two workgroups per CU cover each other's load phases; 3 / double-buffered needs 156 KiB = one per CU)
// POOL (fp32 input, single patch buffer): `in` is the gradient w.r.t. the OUTPUT of a 3 x 3 / stride 2 / padding 1 max-pool
// ([B, Hp, Wp, Cin], Hp = ceil(Hin / 2)) and `in2` that pool's arg-max bytes (code | 0x80 = maximum positive, pool_ops.hip); the
// patch of the pool's INPUT gradient (the Hin x Win tensor this layer's transposed taps read) is formed here from the up to four
// windows that cover a pixel -- spaa_maxpool3s2_bwd's gather with the same order of additions -- instead of being read from a tensor
// that a separate launch wrote: ResNet-18's `maxpool` adjoint as the prologue of the stem's input gradient
// (/root/reference/src/python/classifier.py:26-28,59-60: 205 MB written and read again per batch-64 iteration otherwise).
template <bool HIN, int TBH, int RW, bool PDB, bool POOL = false>
__global__ __launch_bounds__(512, PDB ? 1 : 2) void thinmf_kernel(const spaa_tapconv_t p, const int tiles_y, const int tiles_x, const int TBW,
                                                        const int npieces) {
    static_assert(!POOL || (!HIN && !PDB), "the pool-adjoint prologue serves the fp32 single-buffer form");
    constexpr int PB = HIN ? 64 : 128;     // bytes of a staged pixel (32 channels)
    constexpr int CPP = PB / 16;           // 16-byte chunks per pixel
    constexpr int PPP = 1024 / PB;         // pixels per 1-KiB piece
    constexpr int NPL = HIN ? 1 : 3;       // weight planes
    constexpr int WST = NPL * TBH;         // 1-KiB weight pieces per step (one tap column of one channel block)
    constexpr int NB = RW + TBH - 1;       // pixel-row fragments a wave reads per step
    constexpr int TR = 4 * RW;
    constexpr int PPWMAX = ((TR + 3) * (TW + 3) + PPP - 1) / PPP / NW + 1;   // patch pieces per wave at most
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = wave >> 1, pb = wave & 1;
    const int S = p.s_out;
    const int dy0 = p.tap_range[0], dx0 = p.tap_range[2];
    const int PW = TW + TBW - 1;
    const int NPX = (TR + TBH - 1) * PW;
    const int pbuf = npieces * 1024;                       // bytes of a patch buffer
    unsigned char* const wsm = smem + (PDB ? 2 : 1) * pbuf;   // two weight stages of WST KiB

    int img, y0, x0;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        x0 = (t % tiles_x) * TW;
        t /= tiles_x;
        y0 = (t % tiles_y) * TR;
        img = t / tiles_y;
    }
    const int EB = HIN ? 2 : 4;
    const int row_bytes = p.in_cstride * EB;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uint64_t)in_hi << 32) | in_lo), 0,
        (int)__builtin_amdgcn_readfirstlane((uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)row_bytes), 0x00020000);
    const int nkb = p.Cin >> 5;
    const int nsteps = nkb * TBW;
    const void* wptr = HIN ? p.w_half : (const void*)p.w_split;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(wptr);
    const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane((uint32_t)(nsteps * WST * 1024)), 0x00020000);

    // ---- patch staging: piece i of this wave = piece wave + 8 i = PPP consecutive patch pixels; lane -> (pixel, physical chunk)
    int pvoff[PPWMAX];
#pragma unroll
    for (int i = 0; i < PPWMAX; ++i) {
        const int q = (wave + NW * i) * PPP + lane / CPP;   // patch pixel
        const int pr = q / PW, pc = q - pr * PW;
        const int iy = y0 + dy0 + pr, ix = x0 + dx0 + pc;
        const bool ok = q < NPX && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        const int c = (lane % CPP) ^ (HIN ? (q >> 2) & 3 : (q >> 1) & 7);   // logical chunk held at this lane's slot
        pvoff[i] = ok ? ((img * p.Hin + iy) * p.Win + ix) * row_bytes + p.in_coff * EB + c * 16 : (int)0x80000000;
    }
    auto dma_patch = [&](const int buf, const int kb, const int share, const int nshare) {
#pragma unroll
        for (int i = 0; i < PPWMAX; ++i)
            if (wave + NW * i < npieces && (nshare == 1 || i % nshare == share))
                dma16(rsrc_in, smem + buf * pbuf + (wave + NW * i) * 1024, pvoff[i], kb * 32 * EB);
    };
    // POOL: the patch of channel block kb computed from the pooled gradient (item = (patch pixel q, 4-channel group cq): the four
    // candidate windows' (arg-max byte, gradient) pairs are loaded unconditionally, all of a thread's items in flight together)
    auto pool_patch = [&](const int kb) {
        if constexpr (POOL) {
            const int Hp = p.in2_cstride, Wp = p.in2_coff, C4 = p.Cin >> 2;
            const unsigned char need = p.Cin2 ? 0x80 : 0x00;    // (Cin2 != 0: only windows whose maximum is positive pass -- the ReLU gate)
            const uchar4* am_p = reinterpret_cast<const uchar4*>(p.in2);
            const float4* g_p = reinterpret_cast<const float4*>(p.in);
            constexpr int NIT = 2;                               // items in flight per thread (2 x 4 windows x 20 bytes)
            const int nitems = npieces * PPP * 8;
#pragma unroll 1
            for (int it0 = 0; it0 < nitems; it0 += 512 * NIT) {
            uchar4 am[NIT][4];
            float4 gv[NIT][4];
            bool wok[NIT][4];
            unsigned char kk[NIT][4];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int item = it0 + tid + 512 * it, q = item >> 3, cq = item & 7;
                const int pr = q / PW, pc = q - pr * PW;
                const int iy = y0 + dy0 + pr, ix = x0 + dx0 + pc;
                const bool in_img = q < NPX && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                int oyc[2], kyc[2], oxc[2], kxc[2];
                bool yok[2], xok[2];
                if (iy & 1) { oyc[0] = (iy + 1) >> 1; kyc[0] = 0; oyc[1] = (iy - 1) >> 1; kyc[1] = 2; yok[0] = oyc[0] < Hp; yok[1] = true; }
                else        { oyc[0] = iy >> 1; kyc[0] = 1; oyc[1] = 0; kyc[1] = 0; yok[0] = oyc[0] < Hp; yok[1] = false; }
                if (ix & 1) { oxc[0] = (ix + 1) >> 1; kxc[0] = 0; oxc[1] = (ix - 1) >> 1; kxc[1] = 2; xok[0] = oxc[0] < Wp; xok[1] = true; }
                else        { oxc[0] = ix >> 1; kxc[0] = 1; oxc[1] = 0; kxc[1] = 0; xok[0] = oxc[0] < Wp; xok[1] = false; }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool ok = in_img && yok[i] && xok[j];
                        const size_t o = (((size_t)img * Hp + (ok ? oyc[i] : 0)) * Wp + (ok ? oxc[j] : 0)) * C4 + kb * 8 + cq;
                        wok[it][2 * i + j] = ok;
                        kk[it][2 * i + j] = (unsigned char)(kyc[i] * 3 + kxc[j]);
                        am[it][2 * i + j] = am_p[o];
                        gv[it][2 * i + j] = g_p[o];
                    }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int item = it0 + tid + 512 * it, q = item >> 3, cq = item & 7;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uchar4 a = am[it][w];
                    const float4 g = gv[it][w];
                    const unsigned char k = kk[it][w];
                    if (wok[it][w] && (a.x & 0x7f) == k && (a.x & need) == need) acc[0] += g.x;
                    if (wok[it][w] && (a.y & 0x7f) == k && (a.y & need) == need) acc[1] += g.y;
                    if (wok[it][w] && (a.z & 0x7f) == k && (a.z & need) == need) acc[2] += g.z;
                    if (wok[it][w] && (a.w & 0x7f) == k && (a.w & need) == need) acc[3] += g.w;
                }
                if (q < npieces * PPP) *reinterpret_cast<f32x4*>(smem + q * 128 + ((cq ^ ((q >> 1) & 7)) << 4)) = acc;
            }
            }
        }
    };
    auto dma_w = [&](const int stage, const int s) {
#pragma unroll
        for (int i = 0; i < (WST + NW - 1) / NW; ++i)
            if (wave + NW * i < WST) dma16(rsrc_w, wsm + stage * (WST * 1024) + (wave + NW * i) * 1024, lane * 16, (s * WST + wave + NW * i) * 1024);
    };
    // weight fragment address inside a 1-KiB (tap, plane) block: row lane & 15, k-chunk lane >> 4 (host-side swizzle: swz64)
    const int w_addr_l = (lane & 15) * 64 + ((((lane >> 4) ^ (((lane >> 3) & 1) << 1))) << 4);

    f32x4 acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (POOL) pool_patch(0); else dma_patch(0, 0, 0, 1);
    dma_w(0, 0);
    for (int s = 0; s < nsteps; ++s) {
        const int kb = s / TBW, dxi = s - kb * TBW;
        if (!PDB && dxi == 0 && kb > 0) {   // single patch buffer: everybody is done with the previous block's patch
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (POOL) pool_patch(kb); else dma_patch(0, kb, 0, 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this step's weights, and the patch pieces requested so far, have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < nsteps) dma_w((s + 1) & 1, s + 1);
        if (PDB && kb + 1 < nkb) dma_patch((kb + 1) & 1, kb + 1, dxi, TBW);   // (a share per step: every wait above stays short)
        const unsigned char* pbase = smem + (PDB ? (kb & 1) : 0) * pbuf;
        const unsigned char* wbase = wsm + (s & 1) * (WST * 1024) + w_addr_l;
        if constexpr (HIN) {
            h8 bf[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int q = (RW * wrow + j) * PW + 16 * pb + dxi + (lane & 15);
                bf[j] = *reinterpret_cast<const h8*>(pbase + q * 64 + (((lane >> 4) ^ ((q >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int dyi = 0; dyi < TBH; ++dyi) {
                const h8 wf = *reinterpret_cast<const h8*>(wbase + dyi * 1024);
#pragma unroll
                for (int r = 0; r < RW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, bf[r + dyi], acc[r], 0, 0, 0);
            }
        } else {
            bf16x8 bfr[NB][3];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int q = (RW * wrow + j) * PW + 16 * pb + dxi + (lane & 15);
                const int sw = (q >> 1) & 7, c0 = 2 * (lane >> 4);
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(pbase + q * 128 + ((c0 ^ sw) << 4));
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(pbase + q * 128 + (((c0 + 1) ^ sw) << 4));
#ifdef SPAA_THINMF_ABLATE
                if ((p.reserved0 >> 27) & 1) {   // timing only: no operand split
                    bfr[j][0] = __builtin_bit_cast(bf16x8, v0);
                    bfr[j][1] = __builtin_bit_cast(bf16x8, v1);
                    bfr[j][2] = bfr[j][0];
                } else
#endif
                split8(v0, v1, bfr[j][0], bfr[j][1], bfr[j][2]);
            }
#pragma unroll
            for (int dyi = 0; dyi < TBH; ++dyi) {
                bf16x8 wf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wf[pl] = *reinterpret_cast<const bf16x8*>(wbase + (pl * TBH + dyi) * 1024);
#pragma unroll
                for (int r = 0; r < RW; ++r) {   // small terms first (tapconv_x6d.hip: X6D_MFMA6)
                    f32x4 a = acc[r];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2], bfr[r + dyi][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], bfr[r + dyi][2], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], bfr[r + dyi][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1], bfr[r + dyi][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], bfr[r + dyi][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0], bfr[r + dyi][0], a, 0, 0, 0);
                    acc[r] = a;
                }
            }
        }
    }

    // ---- epilogue: lane = (class-grid column lane & 15, class lane >> 4): the 4 channels of output pixel (S y + cls / S, S x + cls % S)
    const int cls = lane >> 4;
    const int cy = cls / S, cx = cls - cy * S;
    const bool cls_ok = cls < S * S;
    const int xg = x0 + 16 * pb + (lane & 15);
    const int ox = S * xg + cx;
    const bool simple = p.out_cstride == 4 && p.out_coff == 0 && p.Cout >= 3 && p.act == SPAA_ACT_NONE && p.gate2 == nullptr &&
                        p.gate_bits == nullptr && p.gate2_bits == nullptr && p.mask_out == nullptr && p.aux_out == nullptr &&
                        (p.add == nullptr || (p.add_cstride == 4 && p.add_coff == 0)) &&
                        (p.gate == nullptr || (p.gate_cstride == 4 && p.gate_coff == 0 && p.gate_mode == SPAA_GATE_MUL)) &&
                        (int64_t)p.B * p.Hout * p.Wout * 16 < ((int64_t)1 << 31);
    if (simple) {
        // out = (acc + bias + add) * gate, four channels per 16-byte access, absent operands = zero-record descriptors
        const int64_t nb = (int64_t)p.B * p.Hout * p.Wout * 16;
        const auto r_out = rsrc_or_empty(p.out, nb), r_add = rsrc_or_empty(p.add, nb), r_gate = rsrc_or_empty(p.gate, nb);
        const auto r_bias = rsrc_or_empty(p.bias, (int64_t)p.Cout * 4);
        float bias[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) bias[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_bias, e < p.Cout ? e * 4 : (int)0x80000000, 0, 0));
        const bool has_gate = p.gate != nullptr;
        int off[RW];
        u32x4 av[RW], gv[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int oy = S * (y0 + RW * wrow + r) + cy;
            const bool ok = cls_ok && oy < p.Hout && ox < p.Wout && xg < p.Wm && y0 + RW * wrow + r < p.Hm;
            off[r] = ok ? ((img * p.Hout + oy) * p.Wout + ox) * 16 : (int)0x80000000;
            av[r] = __builtin_amdgcn_raw_buffer_load_b128(r_add, off[r], 0, 0);
            gv[r] = __builtin_amdgcn_raw_buffer_load_b128(r_gate, off[r], 0, 0);
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[r][e] + bias[e];
                v += __uint_as_float(av[r][e]);
                v = has_gate ? v * __uint_as_float(gv[r][e]) : v;
                o[e] = __float_as_uint(e < p.Cout ? v : 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b128(o, r_out, off[r], 0, 0);
        }
        return;
    }
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int yg = y0 + RW * wrow + r, oy = S * yg + cy;
        if (cls_ok && yg < p.Hm && xg < p.Wm && oy < p.Hout && ox < p.Wout) {
            float v[4] = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
            store4_t<float>(p, ((size_t)img * p.Hout + oy) * p.Wout + ox, 0, v, vec);
        }
    }
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tile 72.  The descriptor describes the layer as usual (its classes are used only
// to check the geometry); the weights come from `w_split` (fp32 input: three bf16 planes) or `w_half` (fp16 input) in the FOLDED
// layout [channel block][tap column][plane][tap row][16 rows = class * 4 + channel][32 channels] that
// spaa_amd/convplan.py: ConvPlan.thin_fold() packs (rows of taps a class does not have are zero; chunk swizzle swz64 baked in).
int spaa_launch_tapconv_thinmf(const spaa_tapconv_t& d, hipStream_t stream) {
    const bool hin = (d.io_dtype & SPAA_IO_IN_F16) != 0;
    if ((d.io_dtype & SPAA_IO_OUT_F16) || d.Cout > 4 || (d.Cin % 32) != 0 || d.s_in != 1 || (d.s_out != 1 && d.s_out != 2) ||
        d.nclass != d.s_out * d.s_out || d.nfold > 1 || d.ksplit > 1 || d.ksplit < 0 || (hin ? d.w_half == nullptr : d.w_split == nullptr))
        return hipErrorInvalidValue;
    for (int c = 0; c < d.nclass; ++c)   // class c writes output pixels (S y + c / S, S x + c % S)
        if (d.cls[c].oy0 != c / d.s_out || d.cls[c].ox0 != c % d.s_out) return hipErrorInvalidValue;
    const int tbh = d.tap_range[1] - d.tap_range[0] + 1, tbw = d.tap_range[3] - d.tap_range[2] + 1;
    if (tbh < 1 || tbh > 4 || tbw < 1 || tbw > 4) return hipErrorInvalidValue;
    if (d.Hm != (d.Hout + d.s_out - 1) / d.s_out || d.Wm != (d.Wout + d.s_out - 1) / d.s_out) return hipErrorInvalidValue;
    if ((int64_t)d.B * d.Hin * d.Win * d.in_cstride * (hin ? 2 : 4) >= (int64_t)1 << 31) return hipErrorInvalidValue;
    const int th = tbh < 2 ? 2 : tbh;   // (a one-row box runs as two rows with zero weights: convplan packs it that way)
    const int rw = hin ? 3 : 2, tr = 4 * rw;
    const int pw = TW + tbw - 1, npx = (tr + th - 1) * pw;
    const int ppp = hin ? 16 : 8;
    const int npieces = (npx + ppp - 1) / ppp;
    const int tiles_y = (d.Hm + tr - 1) / tr, tiles_x = (d.Wm + TW - 1) / TW;
    const int64_t nwg = (int64_t)d.B * tiles_y * tiles_x;
    if (nwg > 0x7fffffff) return hipErrorInvalidValue;
    const size_t smem = (hin ? 2 : 1) * (size_t)npieces * 1024 + 2 * (size_t)(hin ? 1 : 3) * th * 1024;
    static bool attr_set[6][SPAA_MAX_DEVICES] = {};
#define THINMF_LAUNCH(H, T, R, D, SLOT)                                                                                    \
    {                                                                                                                      \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&thinmf_kernel<H, T, R, D>), 160 * 1024, attr_set[SLOT]); \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((thinmf_kernel<H, T, R, D>), dim3((unsigned)nwg), dim3(512), smem, stream, d, tiles_y, tiles_x, tbw, npieces); \
    }
    if (d.in2 != nullptr) {
        // pool-adjoint prologue (see the kernel): fp32, `in` = the pooled gradient [B, Hp, Wp, Cin] contiguous, in2 = arg-max bytes,
        // in2_cstride / in2_coff = Hp / Wp of a 3 x 3 / stride 2 / padding 1 pool over Hin x Win, Cin2 = ReLU gate on / off
        if (hin || d.in_cstride != d.Cin || d.in_coff != 0 || d.in2_cstride != (d.Hin + 2 - 3) / 2 + 1 || d.in2_coff != (d.Win + 2 - 3) / 2 + 1)
            return hipErrorInvalidValue;
#define THINMF_LAUNCH_POOL(T, SLOT)                                                                                        \
    {                                                                                                                      \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&thinmf_kernel<false, T, 2, false, true>), 160 * 1024, attr_pool[SLOT]); \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((thinmf_kernel<false, T, 2, false, true>), dim3((unsigned)nwg), dim3(512), smem, stream, d, tiles_y, tiles_x, tbw, npieces); \
    }
        static bool attr_pool[3][SPAA_MAX_DEVICES] = {};
        if (th == 2) THINMF_LAUNCH_POOL(2, 0) else if (th == 3) THINMF_LAUNCH_POOL(3, 1) else THINMF_LAUNCH_POOL(4, 2)
#undef THINMF_LAUNCH_POOL
        return (int)hipGetLastError();
    }
    if (hin) {
        if (th == 2) THINMF_LAUNCH(true, 2, 3, true, 0) else if (th == 3) THINMF_LAUNCH(true, 3, 3, true, 1) else THINMF_LAUNCH(true, 4, 3, true, 2)
    } else {
        if (th == 2) THINMF_LAUNCH(false, 2, 2, false, 3) else if (th == 3) THINMF_LAUNCH(false, 3, 2, false, 4) else THINMF_LAUNCH(false, 4, 2, false, 5)
    }
#undef THINMF_LAUNCH
    return (int)hipGetLastError();
}
