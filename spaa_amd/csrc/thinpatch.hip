// thinpatch.hip — tap-list convolution for THIN outputs (Cout <= 4) with stride-1 input sampling: conv6 (32 -> 3), the
// input-gradients of conv1 / conv1_s (32 -> 3, four output-parity classes) and of the ResNet stem (64 -> 3, 7x7).
//
// These layers are tiny in FLOPs but wide in bytes; the older thin kernel (tapconv.hip: thinconv_kernel) re-reads the
// input once per tap through the vector L1 / L2 (conv6: 9 x 537 MB per launch) and is bound by that path.  Here a
// workgroup stages the input patch of its 32 x 8 pixel tile ONCE, by LDS-DMA (`buffer_load_dwordx4 ... lds`,
// out-of-image pixels written as zeros by the out-of-range offset), and every tap of every output-parity class reads
// it from LDS:  HBM sees the input once.  One lane = one pixel (all its channels), so there is no cross-lane
// reduction; the weights of a (tap, 16 channels) slice are wave-uniform and arrive through scalar loads (SGPRs feed the
// packed FMAs directly, no LDS traffic for weights).  Per 16-byte LDS read (4 channels of one pixel): 2 x Cout
// v_pk_fma_f32 (even / odd channel partial sums, added at the end).
//
// LDS image: patch pixels in row-major order, Cin*4 bytes each, the 16-byte chunks of a pixel XOR-swizzled so that 16
// neighbouring pixels reading the same channel quad hit 16 different bank groups; the DMA writes lane-linear, so the
// swizzle is applied on the source side (which chunk a lane fetches).
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int TW = 32, TH1 = 8;  // pixel tile of a workgroup (class-grid coordinates): TW x (TH1 * P), P pixels per lane
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(4))) int* cint_ptr;
typedef const __attribute__((address_space(4))) f16v* cf16_ptr;

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
}

constexpr int MAXCLS = SPAA_MAX_CLASSES;

// L = 16-byte chunks per staged pixel (the patch holds CCH = 4 L channels at a time); chunk swizzle of patch pixel q
template <int L>
__device__ __forceinline__ int swz(int q) {
    return L == 8 ? (q >> 1) & 7 : (q >> 2) & 3;
}

// HIN: the input is an fp16 activation (fp16-STORAGE mode: the image-side input gradients of conv1 / conv1_s / the ResNet stem
// read fp16 gradients and write the fp32 image gradient); a 16-byte chunk then holds 8 channels, the arithmetic stays fp32
template <int L, int NOUT, int P, bool HIN = false>
__global__ __launch_bounds__(256) void thinpatch_kernel(const spaa_tapconv_t p, const int tiles_x, const int tiles_y,
                                                        const int dymin, const int dxmin, const int PH, const int PW) {
    constexpr int PIX_PER_PIECE = 64 / L;
    constexpr int CCH = (HIN ? 8 : 4) * L;
    constexpr int EB = HIN ? 2 : 4;   // bytes per input element
    constexpr int TH = TH1 * P;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware order: the workgroups of one XCD take a contiguous range of tiles (halo rows come from its L2)
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

    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)p.in_cstride * (uint32_t)EB;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int npix = PH * PW;
    const int npieces = (npix + PIX_PER_PIECE - 1) / PIX_PER_PIECE;
    const int row_bytes = p.in_cstride * EB;

    const int lx = tid & (TW - 1), ly = tid / TW;  // the lane's pixels: (lx, ly + TH1 * pi), pi < P
    const int x = x0 + lx;
    const bool vec4 = p.out_cstride == 4 && p.out_coff == 0 && (p.add == nullptr || (p.add_cstride == 4 && p.add_coff == 0)) &&
                      (p.gate == nullptr || (p.gate_cstride == 4 && p.gate_coff == 0)) &&
                      (p.gate2 == nullptr || (p.gate2_cstride == 4 && p.gate2_coff == 0));

    f2 acc[MAXCLS][P][NOUT];
#pragma unroll
    for (int ci = 0; ci < MAXCLS; ++ci)
#pragma unroll
        for (int pi = 0; pi < P; ++pi)
#pragma unroll
            for (int n = 0; n < NOUT; ++n) acc[ci][pi][n] = f2{0.f, 0.f};

    for (int c0 = 0; c0 < p.Cin; c0 += CCH) {  // 32 input channels per pass
        if (c0 > 0) __syncthreads();           // everybody is done reading the previous pass's patch
        // ---- stage the patch: rows y0+dymin .. , columns x0+dxmin .. (s_in == 1), channels c0 .. c0+31
        int q = wave * PIX_PER_PIECE + lane / L;
        int py = q / PW, px = q - py * PW;  // one division; the pieces of a wave advance by 4 * PIX_PER_PIECE pixels
        for (int i = wave; i < npieces; i += 4, q += 4 * PIX_PER_PIECE) {
            if (i != wave) {
                px += 4 * PIX_PER_PIECE;
                while (px >= PW) {
                    px -= PW;
                    py += 1;
                }
            }
            const int c = (lane % L) ^ swz<L>(q);
            const int iy = y0 + dymin + py, ix = x0 + dxmin + px;
            const bool v = q < npix && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const int off = ((b * p.Hin + iy) * p.Win + ix) * row_bytes + (p.in_coff + c0) * EB + 16 * c;
            dma16(rsrc_in, smem + i * 1024, v ? off : (int)0x80000000);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

#pragma unroll
        for (int ci = 0; ci < MAXCLS; ++ci) {
            if (ci >= p.nclass) break;
            const spaa_tapclass_t cl = p.cls[ci];
            cint_ptr taps = (cint_ptr)(uintptr_t)(p.taps + 2 * cl.tap_off);
            const float* wbase = p.weights + cl.w_off + c0;
            for (int t = 0; t < cl.ntaps; ++t) {
                const int dy = taps[2 * t], dx = taps[2 * t + 1];
                const unsigned char* pp[P];
                int sw[P];
#pragma unroll
                for (int pi = 0; pi < P; ++pi) {
                    const int q = (ly + TH1 * pi + dy - dymin) * PW + (lx + dx - dxmin);
                    pp[pi] = smem + q * (16 * L);
                    sw[pi] = swz<L>(q);
                }
#pragma unroll
                for (int g = 0; g < CCH / 16; ++g) {  // 16 channels at a time: one s_load_dwordx16 per output channel
                    f16v w[NOUT];
#pragma unroll
                    for (int n = 0; n < NOUT; ++n)
                        w[n] = *(cf16_ptr)(uintptr_t)(wbase + (size_t)n * cl.Kpad + t * p.Cin + 16 * g);
                    if constexpr (HIN) {
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
#pragma unroll
                            for (int pi = 0; pi < P; ++pi) {
                                const h8 ah = *reinterpret_cast<const h8*>(pp[pi] + (((2 * g + u) ^ sw[pi]) * 16));
#pragma unroll
                                for (int v2 = 0; v2 < 4; ++v2) {
                                    const f2 a2 = {(float)ah[2 * v2], (float)ah[2 * v2 + 1]};
#pragma unroll
                                    for (int n = 0; n < NOUT; ++n) {
                                        const f2 w2 = {w[n][8 * u + 2 * v2], w[n][8 * u + 2 * v2 + 1]};
                                        acc[ci][pi][n] = __builtin_elementwise_fma(a2, w2, acc[ci][pi][n]);
                                    }
                                }
                            }
                        }
                        continue;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int pi = 0; pi < P; ++pi) {  // the scalar weights feed every pixel of the lane
                            const f4 a = *reinterpret_cast<const f4*>(pp[pi] + (((4 * g + u) ^ sw[pi]) * 16));
                            const f2 a01 = {a.x, a.y}, a23 = {a.z, a.w};
#pragma unroll
                            for (int n = 0; n < NOUT; ++n) {
                                const f2 w01 = {w[n][4 * u], w[n][4 * u + 1]}, w23 = {w[n][4 * u + 2], w[n][4 * u + 3]};
                                acc[ci][pi][n] = __builtin_elementwise_fma(a01, w01, acc[ci][pi][n]);
                                acc[ci][pi][n] = __builtin_elementwise_fma(a23, w23, acc[ci][pi][n]);
                            }
                        }
                    }
                }
            }
        }
    }

#pragma unroll
    for (int cp_ = 0; cp_ < MAXCLS * P; ++cp_) {
        const int ci = cp_ / P, pi = cp_ % P;
        if (ci >= p.nclass) break;
        const spaa_tapclass_t cl = p.cls[ci];
        const int y = y0 + ly + TH1 * pi;
        if (!(y < p.Hm && x < p.Wm)) continue;
        const int oy = cl.oy0 + y * p.s_out, ox = cl.ox0 + x * p.s_out;
        if (oy >= p.Hout || ox >= p.Wout) continue;
        const size_t o = ((size_t)b * p.Hout + oy) * p.Wout + ox;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NOUT; ++n) v[n] = acc[ci][pi][n].x + acc[ci][pi][n].y;
        if (vec4) {
            // NHWC4 everywhere: one 16-byte access per operand; channels >= Cout stay 0
            f4 addv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f}, g2v = {1.f, 1.f, 1.f, 1.f};
            if (p.add != nullptr) addv = *reinterpret_cast<const f4*>(p.add + o * 4);
            if (p.gate != nullptr) gv = *reinterpret_cast<const f4*>(p.gate + o * 4);
            if (p.gate2 != nullptr) g2v = *reinterpret_cast<const f4*>(p.gate2 + o * 4);
            f4 outv = {0.f, 0.f, 0.f, 0.f}, auxv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < NOUT; ++n) {
                if (n >= p.Cout) break;
                float t = v[n] + (p.bias != nullptr ? p.bias[n] : 0.f) + addv[n];
                if (p.act == SPAA_ACT_RELU) {
                    t = fmaxf(t, 0.f);
                } else if (p.act == SPAA_ACT_RELU_CLAMP1) {
                    t = fmaxf(t, 0.f);
                    auxv[n] = t;
                    t = fminf(t, 1.f);
                } else if (p.act == SPAA_ACT_LEAKY01) {
                    t = t > 0.f ? t : 0.1f * t;
                }
                if (p.gate != nullptr) {
                    if (p.gate_mode == SPAA_GATE_MUL) {
                        t *= gv[n];
                    } else {
                        const bool pass = (p.gate_mode == SPAA_GATE_POS_LE1) ? (gv[n] > 0.f && gv[n] <= 1.f) : (gv[n] > 0.f);
                        t = pass ? t : 0.f;
                    }
                }
                outv[n] = t;
                if (p.gate2 != nullptr) auxv[n] = (g2v[n] > 0.f) ? t : 0.f;
            }
            *reinterpret_cast<f4*>(p.out + o * 4) = outv;
            if (p.aux_out != nullptr && (p.act == SPAA_ACT_RELU_CLAMP1 || p.gate2 != nullptr))
                *reinterpret_cast<f4*>(p.aux_out + o * 4) = auxv;
        } else {
#pragma unroll
            for (int n = 0; n < NOUT; ++n) {
                if (n >= p.Cout) break;
                float t = v[n] + (p.bias != nullptr ? p.bias[n] : 0.f);
                if (p.add != nullptr) t += p.add[o * p.add_cstride + p.add_coff + n];
                if (p.act == SPAA_ACT_RELU) {
                    t = fmaxf(t, 0.f);
                } else if (p.act == SPAA_ACT_RELU_CLAMP1) {
                    t = fmaxf(t, 0.f);
                    if (p.aux_out != nullptr) p.aux_out[o * p.out_cstride + p.out_coff + n] = t;
                    t = fminf(t, 1.f);
                } else if (p.act == SPAA_ACT_LEAKY01) {
                    t = t > 0.f ? t : 0.1f * t;
                }
                if (p.gate != nullptr) {
                    const float g = p.gate[o * p.gate_cstride + p.gate_coff + n];
                    const bool pass = (p.gate_mode == SPAA_GATE_POS_LE1) ? (g > 0.f && g <= 1.f) : (g > 0.f);
                    t = (p.gate_mode == SPAA_GATE_MUL) ? t * g : (pass ? t : 0.f);
                }
                p.out[o * p.out_cstride + p.out_coff + n] = t;
                if (p.gate2 != nullptr) {
                    const float g2 = p.gate2[o * p.gate2_cstride + p.gate2_coff + n];
                    p.aux_out[o * p.out_cstride + p.out_coff + n] = (g2 > 0.f) ? t : 0.f;
                }
            }
        }
    }
}

template <int L, int NOUT, int P, bool HIN = false>
int launch_tp(const spaa_tapconv_t& d, int dymin, int dxmin, int PH, int PW, hipStream_t stream) {
    constexpr int TH = TH1 * P;
    const int tiles_x = (d.Wm + TW - 1) / TW, tiles_y = (d.Hm + TH - 1) / TH;
    const size_t smem = ((size_t)PH * PW * (16 * L) + 1023) / 1024 * 1024;
    if (smem > 64 * 1024) return hipErrorInvalidValue;
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&thinpatch_kernel<L, NOUT, P, HIN>), 64 * 1024, attr_set);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((unsigned)(tiles_x * tiles_y * d.B), 1, 1);
    hipLaunchKernelGGL((thinpatch_kernel<L, NOUT, P, HIN>), grid, dim3(256), smem, stream, d, tiles_x, tiles_y, dymin, dxmin, PH, PW);
    return (int)hipGetLastError();
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tiles 28 (32 channels per pass) and 29 (16 per pass) after the common shape checks.  `d.tap_range` = (dymin, dymax,
// dxmin, dxmax) over the taps of all classes, filled in by the host (the tap list itself lives in device memory).
int spaa_launch_thinpatch(const spaa_tapconv_t& d, hipStream_t stream) {
    // tile 28: 32 channels per pass; 29: 16 per pass; 47: 16 per pass, two pixels per lane (the scalar weights and their
    // latency are shared by twice the FMAs)
    const int L = d.tile == 28 ? 8 : 4;
    const int P = d.tile == 47 ? 2 : 1;
    const bool hin = (d.io_dtype & SPAA_IO_IN_F16) != 0;   // fp16 activation in, fp32 image gradient out (tile 29 only)
    if (hin && (d.tile != 29 || (d.io_dtype & SPAA_IO_OUT_F16))) return hipErrorInvalidValue;
    if (d.Cout > 4 || d.s_in != 1 || (d.Cin % ((hin ? 8 : 4) * L)) != 0) return hipErrorInvalidValue;
    const int dymin = d.tap_range[0], dymax = d.tap_range[1], dxmin = d.tap_range[2], dxmax = d.tap_range[3];
    if (dymax < dymin || dxmax < dxmin || dymax - dymin > 16 || dxmax - dxmin > 16) return hipErrorInvalidValue;
    const int PH = TH1 * P + dymax - dymin, PW = TW + dxmax - dxmin;
    for (int c = 0; c < d.nclass; ++c)
        if (d.cls[c].Kpad % 16) return hipErrorInvalidValue;
    const bool n3 = d.Cout <= 3;
    if (hin) return n3 ? launch_tp<4, 3, 1, true>(d, dymin, dxmin, PH, PW, stream) : launch_tp<4, 4, 1, true>(d, dymin, dxmin, PH, PW, stream);
    if (L == 8) return n3 ? launch_tp<8, 3, 1>(d, dymin, dxmin, PH, PW, stream) : launch_tp<8, 4, 1>(d, dymin, dxmin, PH, PW, stream);
    if (P == 2) return n3 ? launch_tp<4, 3, 2>(d, dymin, dxmin, PH, PW, stream) : launch_tp<4, 4, 2>(d, dymin, dxmin, PH, PW, stream);
    return n3 ? launch_tp<4, 3, 1>(d, dymin, dxmin, PH, PW, stream) : launch_tp<4, 4, 1>(d, dymin, dxmin, PH, PW, stream);
}
