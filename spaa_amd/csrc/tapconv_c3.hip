// tapconv_c3.hip — the FIRST convolution of a network: a 3-channel image (NHWC4 fp32, lane 3 = 0) in, up to 64 channels out:
// torchvision's ResNet-18 `conv1` (7 x 7 / stride 2, /root/reference/src/python/classifier.py:26-28), VGG-16 `features.0` (3 x 3,
// :21-24) and Inception-v3 `Conv2d_1a_3x3` (3 x 3 / stride 2, :29-33), behind `model(im)` at classifier.py:60.
//
// The implicit-GEMM kernels pad the 3 channels to 4 and K = taps x 4 to their 32-deep steps (7 x 7: 224 for 147 real products per
// output; 3 x 3: 64 for 27) and gather one 16-byte fragment per pixel and tap through registers; VGG-16's first layer ran at
// 34 TFLOP/s (326 us for a 411 MB fp16 output).  Here
//   * K = 3 taps-major products only: k = 3 t + c, padded to 32 NK (3 x 3: 27 -> 32, one step; 7 x 7: 147 -> 160, five steps);
//   * a workgroup (8 waves, persistent: it walks tiles) owns 16 x 32 output pixels at a time; the tile's input patch is staged ONCE by
//     LDS-DMA (16 bytes per pixel, out-of-image = the out-of-range offset = the zero padding), one tile ahead; the whole weight matrix
//     (NK x 3 bf16 planes x BN rows x 64 B, chunk swizzle baked in by the host: ConvPlan.c3_pack) once per workgroup;
//   * a lane builds its B fragment -- 8 consecutive k of one pixel -- from eight 4-byte LDS reads at per-lane offsets fixed for the
//     kernel ((tap, channel) of k; k past the last product reads the image's zero lane), splits it exactly into three bf16 planes
//     and multiplies: bf16x6, fp32 accumulation (tapconv_x6d.hip); a weight fragment read from LDS serves two of the wave's four pixel blocks;
//   * epilogue through a wave-private LDS region: whole channel rows per store, the shared branch-free epilogue (bias, ReLU, byte
//     masks; fp32 or fp16 output).
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
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

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

constexpr int OH = 16, OW = 32, NW = 8;
__device__ __forceinline__ int swz_w16(int n) { return ((n >> 3) & 1) << 1; }

// ONE TILE PER WORKGROUP (the 3 x 3 layers: 12 KB of weights, two workgroups per compute unit cover each other's prologue and epilogue;
// the persistent form below measured slower there -- VGG-16 features.0 217 -> 276 us: its tile loop costs registers the two-per-CU
// occupancy does not have)
// HO (round 5, fp16-STORAGE mode: `reserved1` bit 0): the image operand rounded to fp16 in registers and ONE plane of fp16 weights on
// v_mfma_f32_16x16x32_f16 -- what every other layer of the mode multiplies (fp16 operands, fp32 accumulation); the bf16x6 form spent six MFMAs
// and a 44-instruction split per fragment on an output that is rounded to fp16.
template <int NK, int BN, bool HO = false>
__global__ __launch_bounds__(512, 4) void c3conv_tile_kernel(const spaa_tapconv_t p, const int tiles_y, const int tiles_x, const int PH, const int PW,
                                                             const int w_bytes, const int patch_off) {
    constexpr int TJ = BN / 16;
    constexpr int KS_BYTES = (HO ? 1 : 3) * BN * 64;   // one K-step of weights: three planes (HO: one) of BN rows x 32 bf16 / fp16
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const wl = smem;
    unsigned char* const pl = smem + patch_off;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const spaa_tapclass_t cl = p.cls[0];
    const int S = p.s_in, dy0 = p.tap_range[0], dx0 = p.tap_range[2];

    int img, oy0, ox0;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        ox0 = (t % tiles_x) * OW;
        t /= tiles_x;
        oy0 = (t % tiles_y) * OH;
        img = t / tiles_y;
    }
    // ---- stage the weights (linear copy of the host's packed planes) and the patch (64 pixels of 16 bytes per piece)
    {
        const auto rsrc_w = rsrc_or_empty(p.w_split, w_bytes);
        for (int piece = wave; piece < w_bytes / 1024; piece += NW) dma16(rsrc_w, wl + piece * 1024, lane * 16, piece * 1024);
        const auto rsrc_in = rsrc_or_empty(p.in, (int64_t)p.B * p.Hin * p.Win * p.in_cstride * 4);
        const int npx = PH * PW;
        for (int piece = wave; piece * 64 < npx; piece += NW) {
            const int q = piece * 64 + lane;
            const int pr = q / PW, pc = q - pr * PW;
            const int iy = oy0 * S + dy0 + pr, ix = ox0 * S + dx0 + pc;
            const bool ok = q < npx && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            dma16(rsrc_in, pl + piece * 1024, ok ? (((img * p.Hin + iy) * p.Win + ix) * p.in_cstride + p.in_coff) * 4 : (int)0x80000000, 0);
        }
    }
    // ---- per-lane fragment offsets: k = 32 ks + 8 (lane >> 4) + e  ->  (tap t, channel c) = (k / 3, k % 3); past the last product: the
    // zero lane (channel 3) of the patch's first pixel
    const int kq = lane >> 4;
    int koff[NK][8];
    {
        const int kreal = 3 * cl.ntaps;
        const int32_t* tp = p.taps + 2 * cl.tap_off;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * ks + 8 * kq + e;
                const int t = k / 3, c = k - 3 * t;
                const bool real = k < kreal;
                const int dy = real ? tp[2 * t] : dy0, dx = real ? tp[2 * t + 1] : dx0;
                koff[ks][e] = ((dy - dy0) * PW + (dx - dx0)) * 16 + (real ? 4 * c : 12);
            }
    }
    f32x4 acc[4][TJ];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // pixel blocks of this wave: rows 2 wave, 2 wave + 1; columns 0-15, 16-31
    int pbase[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) pbase[b] = (((2 * wave + (b >> 1)) * S) * PW + (16 * (b & 1) + (lane & 15)) * S) * 16;
    const int w_addr_l = (lane & 15) * 64 + ((kq ^ swz_w16(lane & 15)) << 4);

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        bf16x8 pf[4][3];
        h8 pfh[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(pl + pbase[b] + koff[ks][e]);
            if constexpr (HO) pfh[b] = h8{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
            else split8(v, pf[b][0], pf[b][1], pf[b][2]);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const unsigned char* wc = wl + ks * KS_BYTES + j * 1024 + w_addr_l;
            if constexpr (HO) {
                const h8 wh = *reinterpret_cast<const h8*>(wc);
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, pfh[b], acc[b][j], 0, 0, 0);
                continue;
            }
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wc);
            const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wc + BN * 64);
            const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wc + 2 * BN * 64);
#pragma unroll
            for (int b = 0; b < 4; ++b) {   // small terms first (tapconv_x6d.hip: X6D_MFMA6)
                f32x4 a = acc[b][j];
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, pf[b][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[b][2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, pf[b][1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, pf[b][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[b][1], a, 0, 0, 0);
                acc[b][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[b][0], a, 0, 0, 0);
            }
        }
    }

    // ---- epilogue.  D layout of a 16 x 16 block: column (lane & 15) = pixel, rows 4 (lane >> 4) + e = 4 consecutive channels.  Through a
    // wave-private LDS region (free once every wave has left the K loop): a lane then owns 4 channels of a pixel and BN / 4 consecutive
    // lanes its whole channel row (tapconv_h16p.hip)
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) && (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) && (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    constexpr int ROWB = BN * 4 + 16;
    constexpr int LPP = BN / 4, PPI = 64 / LPP;
    __syncthreads();
    unsigned char* const eb = smem + wave * (32 * ROWB);
    const int ch = 4 * (lane % LPP);
    const bool fast = fast_epi_ok(p, vec);
    const bool n_ok = ch < p.Cout;
    const fast_epi_t fe = make_fast_epi(p, n_ok ? ch : 0);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                *reinterpret_cast<f32x4*>(eb + (16 * bb + (lane & 15)) * ROWB + (16 * j + 4 * kq) * 4) = acc[2 * hb + bb][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int oy = oy0 + 2 * wave + hb;
        if (oy < p.Hout) {
            const int orow = (img * p.Hout + oy) * p.Wout + ox0;
            if (fast) {
#define C3_EPI(T)                                                                                                          \
    _Pragma("unroll 1") for (int i0 = 0; i0 < 32 / PPI; i0 += 4) {                                                         \
        fast_pre_t<T> pre[4];                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
            const int pr = (i0 + i) * PPI + lane / LPP;                                                                    \
            pre[i] = fast_epi_load<T>(fe, p, orow + pr, ch, n_ok && ox0 + pr < p.Wout);                                    \
        }                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
            const int pr = (i0 + i) * PPI + lane / LPP;                                                                    \
            const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);                                      \
            fast_epi_store<T>(fe, p, orow + pr, ch, n_ok && ox0 + pr < p.Wout, a, pre[i]);                                 \
        }                                                                                                                  \
    }
                if (p.io_dtype & SPAA_IO_OUT_F16) C3_EPI(_Float16) else C3_EPI(float)
#undef C3_EPI
            } else {
                for (int i = 0; i < 32 / PPI; ++i) {
                    const int pr = i * PPI + lane / LPP;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);
                    float v[4] = {a[0], a[1], a[2], a[3]};
                    if (ox0 + pr < p.Wout && n_ok) store4(p, (size_t)(orow + pr), ch, v, vec);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// PERSISTENT: a workgroup walks tiles  blockIdx.x, + gridDim.x, ...  -- the weights are staged once per workgroup, the patch of
// tile i + 1 is requested (second patch buffer: DB) before the arithmetic of tile i, or (7 x 7: one buffer fits) right after it, under
// tile i's epilogue; the epilogue's stores are still in flight when the next tile's fragments are read.
template <int NK, int BN, bool DB, bool HO = false>
__global__ __launch_bounds__(512, 2) void c3conv_kernel(const spaa_tapconv_t p, const int tiles_y, const int tiles_x, const int PH, const int PW,
                                                        const int w_bytes, const int patch_bytes, const int ntiles) {
    constexpr int TJ = BN / 16;
    constexpr int KS_BYTES = (HO ? 1 : 3) * BN * 64;   // one K-step of weights: three planes (HO: one) of BN rows x 32 bf16 / fp16
    constexpr int ROWB = BN * 4 + 16;       // epilogue: 16 pixels x (BN channels + pad) per wave
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* const wl = smem;
    unsigned char* const pl0 = smem + w_bytes;
    unsigned char* const eb = smem + w_bytes + (DB ? 2 : 1) * patch_bytes + (threadIdx.x >> 6) * (16 * ROWB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const spaa_tapclass_t cl = p.cls[0];
    const int S = p.s_in, dy0 = p.tap_range[0], dx0 = p.tap_range[2];
    const auto rsrc_in = rsrc_or_empty(p.in, (int64_t)p.B * p.Hin * p.Win * p.in_cstride * 4);
    const int npx = PH * PW;

    auto tile_origin = [&](const int t_, int& img, int& oy0, int& ox0) {
        int t = t_;
        ox0 = (t % tiles_x) * OW;
        t /= tiles_x;
        oy0 = (t % tiles_y) * OH;
        img = t / tiles_y;
    };
    // the patch of a tile: 64 pixels of 16 bytes per 1-KiB piece, out-of-image pixels = the out-of-range offset = zeros
    auto dma_patch = [&](unsigned char* dst, const int t_) {
        int img, oy0, ox0;
        tile_origin(t_, img, oy0, ox0);
        int ln = lane;
        asm volatile("" : "+v"(ln));   // (per-lane constants recomputed here, not kept in registers across the tile loop)
        for (int piece = wave; piece * 64 < npx; piece += NW) {
            const int q = piece * 64 + ln;
            const int pr = q / PW, pc = q - pr * PW;
            const int iy = oy0 * S + dy0 + pr, ix = ox0 * S + dx0 + pc;
            const bool ok = q < npx && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            dma16(rsrc_in, dst + piece * 1024, ok ? (((img * p.Hin + iy) * p.Win + ix) * p.in_cstride + p.in_coff) * 4 : (int)0x80000000, 0);
        }
    };
    {   // the weights: a linear copy of the host's packed planes
        const auto rsrc_w = rsrc_or_empty(p.w_split, w_bytes);
        for (int piece = wave; piece < w_bytes / 1024; piece += NW) dma16(rsrc_w, wl + piece * 1024, lane * 16, piece * 1024);
    }
    int tile = blockIdx.x;
    dma_patch(pl0, tile);
    // ---- per-lane fragment offsets: k = 32 ks + 8 (lane >> 4) + e  ->  (tap t, channel c) = (k / 3, k % 3); past the last product: the
    // zero lane (channel 3) of the patch's first pixel
    const int kq = lane >> 4;
    int koff[NK][8];
    {
        const int kreal = 3 * cl.ntaps;
        const int32_t* tp = p.taps + 2 * cl.tap_off;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * ks + 8 * kq + e;
                const int t = k / 3, c = k - 3 * t;
                const bool real = k < kreal;
                const int dy = real ? tp[2 * t] : dy0, dx = real ? tp[2 * t + 1] : dx0;
                koff[ks][e] = ((dy - dy0) * PW + (dx - dx0)) * 16 + (real ? 4 * c : 12);
            }
    }
    // pixel blocks of this wave: rows 2 wave, 2 wave + 1; columns 0-15, 16-31
    int pbase[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) pbase[b] = (((2 * wave + (b >> 1)) * S) * PW + (16 * (b & 1) + (lane & 15)) * S) * 16;
    const int w_addr_l = (lane & 15) * 64 + ((kq ^ swz_w16(lane & 15)) << 4);
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) && (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) && (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    constexpr int LPP = BN / 4, PPI = 64 / LPP;
    const int ch = 4 * (lane % LPP);
    const bool fast = fast_epi_ok(p, vec);
    const bool n_ok = ch < p.Cout;

    for (int it = 0; tile < ntiles; tile += gridDim.x, ++it) {
        const unsigned char* pl = pl0 + (DB ? (it & 1) * patch_bytes : 0);
        // this tile's patch (and, the first time, the weights) has landed -- and the previous tile's stores have left
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (DB && tile + gridDim.x < ntiles) dma_patch(pl0 + ((it + 1) & 1) * patch_bytes, tile + gridDim.x);
        f32x4 acc[4][TJ];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[b][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
            for (int bp = 0; bp < 2; ++bp) {   // (two pixel blocks at a time: 24 fragment registers live instead of 48)
                bf16x8 pf[2][3];
                h8 pfh[2];
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float*>(pl + pbase[2 * bp + bb] + koff[ks][e]);
                    if constexpr (HO) pfh[bb] = h8{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3], (_Float16)v[4], (_Float16)v[5], (_Float16)v[6], (_Float16)v[7]};
                    else split8(v, pf[bb][0], pf[bb][1], pf[bb][2]);
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const unsigned char* wc = wl + ks * KS_BYTES + j * 1024 + w_addr_l;
                    if constexpr (HO) {
                        const h8 wh = *reinterpret_cast<const h8*>(wc);
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) acc[2 * bp + bb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, pfh[bb], acc[2 * bp + bb][j], 0, 0, 0);
                        continue;
                    }
                    const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wc);
                    const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wc + BN * 64);
                    const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wc + 2 * BN * 64);
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {   // small terms first (tapconv_x6d.hip: X6D_MFMA6)
                        f32x4 a = acc[2 * bp + bb][j];
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, pf[bb][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[bb][2], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, pf[bb][1], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, pf[bb][0], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[bb][1], a, 0, 0, 0);
                        acc[2 * bp + bb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, pf[bb][0], a, 0, 0, 0);
                    }
                }
            }
        }
        if (!DB && tile + gridDim.x < ntiles) {   // one patch buffer: everybody is done reading it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
            dma_patch(pl0, tile + gridDim.x);
        }
        // ---- epilogue.  D layout of a 16 x 16 block: column (lane & 15) = pixel, rows 4 (lane >> 4) + e = 4 consecutive channels.
        // Through a wave-private LDS region of 16 pixels: a lane then owns 4 channels of a pixel and BN / 4 consecutive lanes its
        // whole channel row (tapconv_h16p.hip)
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        const fast_epi_t fe = make_fast_epi(p, n_ok ? ch : 0);   // (per tile: its registers are free during the arithmetic)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) *reinterpret_cast<f32x4*>(eb + (lane & 15) * ROWB + (16 * j + 4 * kq) * 4) = acc[b][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int oy = oy0 + 2 * wave + (b >> 1), oxb = ox0 + 16 * (b & 1);
            const int orow = (img * p.Hout + oy) * p.Wout + oxb;
            const bool row_ok = oy < p.Hout;
            if (fast) {
#define C3_EPI(T)                                                                                                          \
    {                                                                                                                      \
        fast_pre_t<T> pre[16 / PPI];                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 16 / PPI; ++i) {                                                             \
            const int pr = i * PPI + lane / LPP;                                                                           \
            pre[i] = fast_epi_load<T>(fe, p, orow + pr, ch, row_ok && n_ok && oxb + pr < p.Wout);                          \
        }                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 16 / PPI; ++i) {                                                             \
            const int pr = i * PPI + lane / LPP;                                                                           \
            const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);                                      \
            fast_epi_store<T>(fe, p, orow + pr, ch, row_ok && n_ok && oxb + pr < p.Wout, a, pre[i]);                       \
        }                                                                                                                  \
    }
                if (p.io_dtype & SPAA_IO_OUT_F16) C3_EPI(_Float16) else C3_EPI(float)
#undef C3_EPI
            } else {
                for (int i = 0; i < 16 / PPI; ++i) {
                    const int pr = i * PPI + lane / LPP;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(eb + pr * ROWB + ch * 4);
                    float v[4] = {a[0], a[1], a[2], a[3]};
                    if (row_ok && oxb + pr < p.Wout && n_ok) store4(p, (size_t)(orow + pr), ch, v, vec);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tile 76 after the common shape checks: ONE class, Cin = 4 (three image channels + the
// zero lane: the CALLER vouches for the zero lane -- spaa_amd/convplan.py offers the tile only to plans built from 3-channel weights),
// stride 1 or 2, Cout <= 64, 3 x taps <= 32 (one K-step) or <= 160 (five); `w_split` = the planes ConvPlan.c3_pack() lays out:
// [NK][3 planes][BN rows][32 bf16], 16-byte chunk c of row n at chunk c ^ swz_w16(n), BN = 32 (Cout <= 32) or 64.
int spaa_launch_tapconv_c3(const spaa_tapconv_t& d, hipStream_t stream) {
    if (d.w_split == nullptr || d.Cin != 4 || d.nclass != 1 || d.s_out != 1 || d.Hm != d.Hout || d.Wm != d.Wout || d.s_in < 1 || d.s_in > 2 ||
        d.Cout > 64 || d.cls[0].ntaps < 1 || d.nfold > 1 || d.ksplit > 1 || d.ksplit < 0 || (d.io_dtype & SPAA_IO_IN_F16) || d.in2 != nullptr)
        return hipErrorInvalidValue;
    const int nk = 3 * d.cls[0].ntaps <= 32 ? 1 : (3 * d.cls[0].ntaps <= 160 ? 5 : 0);
    if (!nk) return hipErrorInvalidValue;
    const int kh = d.tap_range[1] - d.tap_range[0] + 1, kw = d.tap_range[3] - d.tap_range[2] + 1;
    if (kh < 1 || kw < 1 || kh > 8 || kw > 8) return hipErrorInvalidValue;
    if ((int64_t)d.B * d.Hin * d.Win * d.in_cstride * 4 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    const int PH = (OH - 1) * d.s_in + kh, PW = (OW - 1) * d.s_in + kw;
    const int BN = d.Cout <= 32 ? 32 : 64;
    const bool ho = (d.reserved1 & 1) != 0;      // fp16 operands (fp16-storage mode): one fp16 weight plane
    if (ho && !(d.io_dtype & SPAA_IO_OUT_F16)) return hipErrorInvalidValue;
    const int w_bytes = nk * (ho ? 1 : 3) * BN * 64;
    const int patch_bytes = (PH * PW * 16 + 1023) & ~1023;
    const int tiles_y = (d.Hout + OH - 1) / OH, tiles_x = (d.Wout + OW - 1) / OW;
    const int64_t ntiles = (int64_t)d.B * tiles_y * tiles_x;
    if (ntiles > 0x7fffffff) return hipErrorInvalidValue;
    static bool attr_set[8][SPAA_MAX_DEVICES] = {};
    if (nk == 1) {
        // one tile per workgroup: weights + patch, then (aliased) 32 pixels x (BN + 4) floats per wave for the epilogue
        const int epi_bytes = NW * 32 * (BN * 4 + 16), main_bytes = w_bytes + patch_bytes;
        const size_t smem = (size_t)(main_bytes > epi_bytes ? main_bytes : epi_bytes);
        if (smem > 80 * 1024) return hipErrorInvalidValue;
#define C3_LAUNCH_TILE(N, HO_, SLOT)                                                                                       \
    {                                                                                                                      \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&c3conv_tile_kernel<1, N, HO_>), 80 * 1024, attr_set[SLOT]); \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((c3conv_tile_kernel<1, N, HO_>), dim3((unsigned)ntiles), dim3(512), smem, stream, d, tiles_y, tiles_x, PH, PW, w_bytes, w_bytes); \
    }
        if (ho) {
            if (BN == 32) C3_LAUNCH_TILE(32, true, 4) else C3_LAUNCH_TILE(64, true, 5)
        } else if (BN == 32) C3_LAUNCH_TILE(32, false, 0) else C3_LAUNCH_TILE(64, false, 1)
#undef C3_LAUNCH_TILE
        return (int)hipGetLastError();
    }
    // 7 x 7: persistent, one patch buffer (60 KB of weights + 40 KB + 35 KB of epilogue rows: one workgroup per compute unit)
    const int epi_bytes = NW * 16 * (BN * 4 + 16);
    const size_t smem = (size_t)w_bytes + patch_bytes + epi_bytes;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    const int slots = (smem <= 80 * 1024 ? 2 : 1) * ncu;
    const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
#define C3_LAUNCH(N, HO_, SLOT)                                                                                            \
    {                                                                                                                      \
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&c3conv_kernel<5, N, false, HO_>), 160 * 1024, attr_set[SLOT]); \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((c3conv_kernel<5, N, false, HO_>), dim3(grid), dim3(512), smem, stream, d, tiles_y, tiles_x, PH, PW, w_bytes, patch_bytes, \
                           (int)ntiles);                                                                                   \
    }
    if (ho) {
        if (BN == 32) C3_LAUNCH(32, true, 6) else C3_LAUNCH(64, true, 7)
    } else if (BN == 32) C3_LAUNCH(32, false, 2) else C3_LAUNCH(64, false, 3)
#undef C3_LAUNCH
    return (int)hipGetLastError();
}
