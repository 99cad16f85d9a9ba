// epilogue.hpp — the fused tap-list convolution epilogue shared by the DMA-staged kernels (tapconv_x6d.hip,
// smallcin.hip): bias + residual + activation (+ pre-clamp second output) + ReLU-backward gate(s), 4 channels at a time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// fused epilogue for 4 consecutive output channels n0..n0+3 of output pixel o
__device__ __forceinline__ void store4(const spaa_tapconv_t& p, const size_t o, const int n0, float (&v)[4], const bool vec) {
    if (n0 >= p.Cout) return;
    if (vec) {
        if (p.bias != nullptr) {
            const f4 bb = *reinterpret_cast<const f4*>(p.bias + n0);
            v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        if (p.add != nullptr) {
            const f4 aa = *reinterpret_cast<const f4*>(p.add + o * p.add_cstride + p.add_coff + n0);
            v[0] += aa.x; v[1] += aa.y; v[2] += aa.z; v[3] += aa.w;
        }
        float* outp = p.out + o * p.out_cstride + p.out_coff + n0;
        if (p.act == SPAA_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == SPAA_ACT_RELU_CLAMP1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            if (p.aux_out != nullptr)
                *reinterpret_cast<f4*>(p.aux_out + o * p.out_cstride + p.out_coff + n0) = f4{v[0], v[1], v[2], v[3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fminf(v[e], 1.f);
        } else if (p.act == SPAA_ACT_LEAKY01) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
        }
        if (p.gate != nullptr) {
            const f4 gg = *reinterpret_cast<const f4*>(p.gate + o * p.gate_cstride + p.gate_coff + n0);
            const float ga[4] = {gg.x, gg.y, gg.z, gg.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool pass = (p.gate_mode == SPAA_GATE_POS_LE1) ? (ga[e] > 0.f && ga[e] <= 1.f) : (ga[e] > 0.f);
                v[e] = (p.gate_mode == SPAA_GATE_MUL) ? v[e] * ga[e] : (pass ? v[e] : 0.f);
            }
        } else if (p.gate_bits != nullptr) {  // the same ReLU gate as 1 byte per 4 channels (written through mask_out)
            const unsigned int mb = p.gate_bits[(o * p.gate_cstride + p.gate_coff + n0) >> 2];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
        }
        *reinterpret_cast<f4*>(outp) = f4{v[0], v[1], v[2], v[3]};
        if (p.mask_out != nullptr)
            p.mask_out[(o * p.out_cstride + p.out_coff + n0) >> 2] =
                (uint8_t)((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u));
        if (p.gate2 != nullptr) {
            const f4 gg = *reinterpret_cast<const f4*>(p.gate2 + o * p.gate2_cstride + p.gate2_coff + n0);
            *reinterpret_cast<f4*>(p.aux_out + o * p.out_cstride + p.out_coff + n0) =
                f4{gg.x > 0.f ? v[0] : 0.f, gg.y > 0.f ? v[1] : 0.f, gg.z > 0.f ? v[2] : 0.f, gg.w > 0.f ? v[3] : 0.f};
        } else if (p.gate2_bits != nullptr) {
            const unsigned int mb = p.gate2_bits[(o * p.gate2_cstride + p.gate2_coff + n0) >> 2];
            *reinterpret_cast<f4*>(p.aux_out + o * p.out_cstride + p.out_coff + n0) =
                f4{(mb & 1u) ? v[0] : 0.f, (mb & 2u) ? v[1] : 0.f, (mb & 4u) ? v[2] : 0.f, (mb & 8u) ? v[3] : 0.f};
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + e;
            if (n >= p.Cout) continue;
            float t = v[e] + (p.bias != nullptr ? p.bias[n] : 0.f);
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
                const float gv = p.gate[o * p.gate_cstride + p.gate_coff + n];
                const bool pass = (p.gate_mode == SPAA_GATE_POS_LE1) ? (gv > 0.f && gv <= 1.f) : (gv > 0.f);
                t = (p.gate_mode == SPAA_GATE_MUL) ? t * gv : (pass ? t : 0.f);
            }
            p.out[o * p.out_cstride + p.out_coff + n] = t;
            if (p.gate2 != nullptr) {
                const float g2 = p.gate2[o * p.gate2_cstride + p.gate2_coff + n];
                p.aux_out[o * p.out_cstride + p.out_coff + n] = (g2 > 0.f) ? t : 0.f;
            }
        }
    }
}

// output pixel index of tile row m (class grid -> output grid); false when the pixel does not exist
__device__ __forceinline__ bool out_pixel(const spaa_tapconv_t& p, const spaa_tapclass_t& cl, const int m, const int M,
                                          const int HWm, size_t& o) {
    if (m >= M) return false;
    if ((p.s_out == 1) && (cl.oy0 == 0) && (cl.ox0 == 0) && (p.Hm == p.Hout) && (p.Wm == p.Wout)) {
        o = (size_t)m;
        return true;
    }
    const int b = m / HWm;
    const int rr = m - b * HWm;
    const int y = rr / p.Wm;
    const int x = rr - y * p.Wm;
    const int oy = cl.oy0 + y * p.s_out;
    const int ox = cl.ox0 + x * p.s_out;
    if (oy >= p.Hout || ox >= p.Wout) return false;
    o = ((size_t)b * p.Hout + oy) * p.Wout + ox;
    return true;
}

}  // namespace
