// smallcin.hip — tap-list convolution for layers with FEW INPUT channels (Cin = 4 or 8 after padding) and up to 32 (one pass) or
// 64 (two 32-channel halves over one staged patch) output channels: conv1 (3 -> 32, stride 2), conv1_s (6 -> 32, stride 2), the
// input-gradient of conv6 (3 -> 32) and VGG-16's first layer (3 -> 64).
//
// K = taps * Cin is 36 or 72: the general implicit-GEMM kernels spend their time gathering 16-byte im2col fragments
// (one bounds-checked load per pixel and tap) and padding K to their 32-deep steps.  These layers are HBM-bound (the
// 32-channel side is 8-16x larger than the 3-channel side), so the arithmetic stays on the exact fp32 matrix
// instruction (v_mfma_f32_32x32x2_f32, 18-36 of them per 32 pixels) and the work goes into the data path:
//   * the input patch of a 32 x 8 pixel tile (with halo, stride 1 or 2) is staged ONCE by LDS-DMA, out-of-image pixels
//     as zeros (out-of-range buffer offset);
//   * the whole weight matrix lives in registers (2 per tap and channel quad: a lane holds W[n = lane & 31][k] for the
//     two k it feeds);
//   * per tap and channel quad a lane reads 8 bytes from LDS: lanes 0-31 channels (0,1), lanes 32-63 channels (2,3) of
//     pixel (lane & 31) — the two k-slices of two consecutive MFMAs;
//   * epilogue: the 32 x 32 result tile of a wave goes through a wave-private LDS slab, so that a store instruction
//     writes 8 pixels' complete 128-byte channel rows (and the gate / residual loads read complete rows) instead of
//     sixteen-byte pieces of 32 different rows.
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TW = 32, TH = 8;
constexpr int MAXT = 9;  // taps held in registers
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(4))) int* cint_ptr;

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// CG = channel quads per pixel (Cin / 4); NH = 32-channel halves of the output (2: Cout <= 64 -- VGG-16's first layer, 3 -> 64 at
// 224 x 224, classifier.py:21-24: the halves share the patch and its LDS reads)
template <int CG, bool SLAB, int NH = 1>
__global__ __launch_bounds__(256) void smallcin_kernel(const spaa_tapconv_t p, const int tiles_x, const int tiles_y,
                                                       const int dymin, const int dxmin, const int PH, const int PW) {
    constexpr int PIXB = 16 * CG;              // bytes per staged pixel
    constexpr int PIX_PER_PIECE = 1024 / PIXB;  // pixels per 1-KiB DMA piece
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const spaa_tapclass_t cl = p.cls[0];

    int tile;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- stage the input patch: rows y0*s_in + dymin .., columns x0*s_in + dxmin ..
    {
        const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)p.in_cstride * 4u;
        const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
        const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
        const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
        const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo),
                                                                0, (int)__builtin_amdgcn_readfirstlane(in_bytes),
                                                                0x00020000);
        const int npix = PH * PW;
        const int npieces = (npix + PIX_PER_PIECE - 1) / PIX_PER_PIECE;
        const int row_bytes = p.in_cstride * 4;
        int q = wave * PIX_PER_PIECE + lane / CG;
        int py = q / PW, px = q - py * PW;
        for (int i = wave; i < npieces; i += 4, q += 4 * PIX_PER_PIECE) {
            if (i != wave) {
                px += 4 * PIX_PER_PIECE;
                while (px >= PW) {
                    px -= PW;
                    py += 1;
                }
            }
            const int iy = y0 * p.s_in + dymin + py, ix = x0 * p.s_in + dxmin + px;
            const bool v = q < npix && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const int off = ((b * p.Hin + iy) * p.Win + ix) * row_bytes + (p.in_coff + 4 * (lane % CG)) * 4;
            dma16(rsrc_in, smem + i * 1024, v ? off : (int)0x80000000);
        }
    }

    // ---- the weight matrix in registers: lane -> output channel (lane & 31), channels 2*(lane>>5) + {0,1} of every
    // (tap, quad); rows >= Cout of the packed matrix are zero
    float wA[NH][MAXT][CG], wB[NH][MAXT][CG];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const float* wr = p.weights + cl.w_off + (size_t)(32 * h + (lane & 31)) * cl.Kpad + 2 * (lane >> 5);
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int g = 0; g < CG; ++g) {
                const bool has = t < cl.ntaps;
                const f2 w2 = has ? *reinterpret_cast<const f2*>(wr + t * (4 * CG) + 4 * g) : f2{0.f, 0.f};
                wA[h][t][g] = w2.x;
                wB[h][t][g] = w2.y;
            }
    }
    cint_ptr taps = (cint_ptr)(uintptr_t)(p.taps + 2 * cl.tap_off);
    int toff[MAXT];  // LDS byte offset of tap t relative to the lane's own pixel
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int tt = t < cl.ntaps ? t : 0;
        toff[t] = ((taps[2 * tt] - dymin) * PW + (taps[2 * tt + 1] - dxmin)) * PIXB;
    }
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int lx = lane & 31;
    const int slab_off = ((PH * PW * PIXB + 1023) / 1024) * 1024;
    const bool fast = SLAB && fast_epi_ok(p, vec);
    fast_epi_t fes[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) fes[h] = make_fast_epi(p, 32 * h + 4 * (lane & 7) < p.Cout ? 32 * h + 4 * (lane & 7) : 0);
#pragma unroll
    for (int r = 0; r < TH / 4; ++r) {
        const int ly = wave * (TH / 4) + r;
        const unsigned char* pp = smem + ((ly * p.s_in) * PW + lx * p.s_in) * PIXB + (lane >> 5) * 8;
        f32x16 accs[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) accs[h][i] = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t < cl.ntaps) {
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    const f2 v = *reinterpret_cast<const f2*>(pp + toff[t] + 16 * g);
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        accs[h] = mfma2(wA[h][t][g], v.x, accs[h]);
                        accs[h] = mfma2(wB[h][t][g], v.y, accs[h]);
                    }
                }
            }
        }
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const f32x16 acc = accs[h];
        const fast_epi_t& fe = fes[h];
        const int nb = 32 * h;   // first output channel of this half
        if constexpr (!SLAB) {
            const int y = y0 + ly, x = x0 + lx;
            size_t o;
            if (y >= p.Hm || x >= p.Wm || !out_pixel(p, cl, (b * p.Hm + y) * p.Wm + x, M, HWm, o)) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                store4(p, o, nb + 8 * g + 4 * (lane >> 5), v, vec);
            }
            continue;
        }
        // transpose through the wave's slab: [32 pixels][36 floats]; write own pixel's 4 x 4 channels ...
        float* slab = reinterpret_cast<float*>(smem + slab_off) + wave * (32 * 36);
        if (NH > 1 && h > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the previous half's reads of the slab are done)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f4*>(slab + lx * 36 + 8 * g + 4 * (lane >> 5)) = f4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        // ... read back 8 lanes per pixel, 4 channels per lane
        const int y = y0 + ly;
        if (fast) {   // branch-free operand accesses, the row's four pixel groups in flight (epilogue.hpp: fast_epi_*)
#define SC_FAST(T)                                                                                                     \
    {                                                                                                                  \
        fast_pre_t<T> pre[4];                                                                                          \
        size_t oo[4] = {0, 0, 0, 0};                                                                                   \
        bool ok[4];                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
            const int x = x0 + 8 * i + (lane >> 3);                                                                    \
            ok[i] = y < p.Hm && x < p.Wm && out_pixel(p, cl, (b * p.Hm + y) * p.Wm + x, M, HWm, oo[i]) && nb + 4 * (lane & 7) < p.Cout; \
            pre[i] = fast_epi_load<T>(fe, p, (int)oo[i], nb + 4 * (lane & 7), ok[i]);                                  \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
            const f4 t = *reinterpret_cast<const f4*>(slab + (8 * i + (lane >> 3)) * 36 + 4 * (lane & 7));             \
            fast_epi_store<T>(fe, p, (int)oo[i], nb + 4 * (lane & 7), ok[i], t, pre[i]);                               \
        }                                                                                                              \
    }
            if (p.io_dtype & SPAA_IO_OUT_F16) SC_FAST(_Float16) else SC_FAST(float)
#undef SC_FAST
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = 8 * i + (lane >> 3), x = x0 + px;
            const f4 t = *reinterpret_cast<const f4*>(slab + px * 36 + 4 * (lane & 7));
            size_t o;
            if (y >= p.Hm || x >= p.Wm || !out_pixel(p, cl, (b * p.Hm + y) * p.Wm + x, M, HWm, o)) continue;
            float v[4] = {t.x, t.y, t.z, t.w};
            store4(p, o, nb + 4 * (lane & 7), v, vec);
        }
      }
    }
}

template <int CG, bool SLAB, int NH = 1>
int launch_sc(const spaa_tapconv_t& d, int dymin, int dxmin, int PH, int PW, hipStream_t stream) {
    const int tiles_x = (d.Wm + TW - 1) / TW, tiles_y = (d.Hm + TH - 1) / TH;
    const size_t smem = ((size_t)PH * PW * (16 * CG) + 1023) / 1024 * 1024 + (SLAB ? 4 * 32 * 36 * sizeof(float) : 0);
    if (smem > 64 * 1024) return hipErrorInvalidValue;
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&smallcin_kernel<CG, SLAB, NH>), 64 * 1024, attr_set);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B), 1, 1);
    hipLaunchKernelGGL((smallcin_kernel<CG, SLAB, NH>), grid, dim3(256), smem, stream, d, tiles_x, tiles_y, dymin, dxmin, PH, PW);
    return (int)hipGetLastError();
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tile 38 after the common shape checks
int spaa_launch_smallcin(const spaa_tapconv_t& d, hipStream_t stream) {
    if (d.nclass != 1 || d.Cout > 64 || (d.Cin != 4 && d.Cin != 8) || d.cls[0].ntaps > MAXT || d.cls[0].ntaps < 1 ||
        d.s_in < 1 || d.s_in > 2 || d.ksplit > 1 || d.nfold > 1)
        return hipErrorInvalidValue;
    const int dymin = d.tap_range[0], dymax = d.tap_range[1], dxmin = d.tap_range[2], dxmax = d.tap_range[3];
    if (dymax < dymin || dxmax < dxmin || dymax - dymin > 8 || dxmax - dxmin > 8) return hipErrorInvalidValue;
    const int PH = (TH - 1) * d.s_in + dymax - dymin + 1, PW = (TW - 1) * d.s_in + dxmax - dxmin + 1;
    // stride-1 layers write 8x more bytes than they read: coalesce the epilogue through LDS (whole 128-byte channel rows, the
    // branch-free operand accesses of epilogue.hpp); a stride-2 layer pays only when its epilogue reads a residual as well
    // (conv1: 145 -> 96 us; conv1_s without one: 112 -> 128 us)
    if (d.Cout > 32)   // two 32-channel halves per workgroup (always through the slab: 64-channel rows)
        return d.Cin == 4 ? launch_sc<1, true, 2>(d, dymin, dxmin, PH, PW, stream) : launch_sc<2, true, 2>(d, dymin, dxmin, PH, PW, stream);
    if (d.s_in == 1 || (d.add != nullptr && !((d.reserved0 >> 26) & 1)))
        return d.Cin == 4 ? launch_sc<1, true>(d, dymin, dxmin, PH, PW, stream) : launch_sc<2, true>(d, dymin, dxmin, PH, PW, stream);
    return d.Cin == 4 ? launch_sc<1, false>(d, dymin, dxmin, PH, PW, stream) : launch_sc<2, false>(d, dymin, dxmin, PH, PW, stream);
}
