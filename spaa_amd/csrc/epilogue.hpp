// epilogue.hpp — the fused tap-list convolution epilogue shared by the DMA-staged kernels (tapconv_x6d.hip,
// smallcin.hip): bias + residual + activation (+ pre-clamp second output) + ReLU-backward gate(s), 4 channels at a time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// 4 consecutive elements of a tensor stored as fp32 or as fp16 (fp16-STORAGE mode: spaa_tapconv_t.io_dtype)
template <typename T>
struct io4;
template <>
struct io4<float> {
    static __device__ __forceinline__ f4 ld(const void* base, size_t idx) {
        return *reinterpret_cast<const f4*>(reinterpret_cast<const float*>(base) + idx);
    }
    static __device__ __forceinline__ void st(void* base, size_t idx, f4 v) {
        *reinterpret_cast<f4*>(reinterpret_cast<float*>(base) + idx) = v;
    }
    static __device__ __forceinline__ float ld1(const void* base, size_t idx) { return reinterpret_cast<const float*>(base)[idx]; }
    static __device__ __forceinline__ void st1(void* base, size_t idx, float v) { reinterpret_cast<float*>(base)[idx] = v; }
};
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
template <>
struct io4<_Float16> {
    static __device__ __forceinline__ f4 ld(const void* base, size_t idx) {
        const h4 h = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(base) + idx);
        return f4{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    }
    static __device__ __forceinline__ void st(void* base, size_t idx, f4 v) {
        *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(base) + idx) =
            h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    }
    static __device__ __forceinline__ float ld1(const void* base, size_t idx) {
        return (float)reinterpret_cast<const _Float16*>(base)[idx];
    }
    static __device__ __forceinline__ void st1(void* base, size_t idx, float v) {
        reinterpret_cast<_Float16*>(base)[idx] = (_Float16)v;
    }
};

// fused epilogue for 4 consecutive output channels n0..n0+3 of output pixel o; T = storage type of out / add / gate /
// aux_out / gate2 (bias is always fp32).  The mask bits and the value handed back in `v` are those of the STORED value.
template <typename T>
__device__ __forceinline__ void store4_t(const spaa_tapconv_t& p, const size_t o, const int n0, float (&v)[4], const bool vec) {
    if (n0 >= p.Cout) return;
    if (vec) {
        if (p.bias != nullptr) {
            const f4 bb = *reinterpret_cast<const f4*>(p.bias + n0);
            v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        if (p.add != nullptr) {
            const f4 aa = io4<T>::ld(p.add, o * p.add_cstride + p.add_coff + n0);
            v[0] += aa.x; v[1] += aa.y; v[2] += aa.z; v[3] += aa.w;
        }
        const size_t oi = o * p.out_cstride + p.out_coff + n0;
        if (p.act == SPAA_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == SPAA_ACT_RELU_CLAMP1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            if (p.aux_out != nullptr) io4<T>::st(p.aux_out, oi, f4{v[0], v[1], v[2], v[3]});
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fminf(v[e], 1.f);
        } else if (p.act == SPAA_ACT_LEAKY01) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
        }
        if (p.gate != nullptr) {
            const f4 gg = io4<T>::ld(p.gate, o * p.gate_cstride + p.gate_coff + n0);
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
        if (sizeof(T) == 2) {  // what is stored is the fp16 rounding: gate bits must describe that value
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)(_Float16)v[e];
        }
        io4<T>::st(p.out, oi, f4{v[0], v[1], v[2], v[3]});
        if (p.mask_out != nullptr)
            p.mask_out[(o * p.out_cstride + p.out_coff + n0) >> 2] =
                (uint8_t)((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u));
        if (p.gate2 != nullptr) {
            const f4 gg = io4<T>::ld(p.gate2, o * p.gate2_cstride + p.gate2_coff + n0);
            io4<T>::st(p.aux_out, oi, f4{gg.x > 0.f ? v[0] : 0.f, gg.y > 0.f ? v[1] : 0.f, gg.z > 0.f ? v[2] : 0.f,
                                        gg.w > 0.f ? v[3] : 0.f});
        } else if (p.gate2_bits != nullptr) {
            const unsigned int mb = p.gate2_bits[(o * p.gate2_cstride + p.gate2_coff + n0) >> 2];
            io4<T>::st(p.aux_out, oi, f4{(mb & 1u) ? v[0] : 0.f, (mb & 2u) ? v[1] : 0.f, (mb & 4u) ? v[2] : 0.f,
                                        (mb & 8u) ? v[3] : 0.f});
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + e;
            if (n >= p.Cout) continue;
            float t = v[e] + (p.bias != nullptr ? p.bias[n] : 0.f);
            if (p.add != nullptr) t += io4<T>::ld1(p.add, o * p.add_cstride + p.add_coff + n);
            const size_t oi = o * p.out_cstride + p.out_coff + n;
            if (p.act == SPAA_ACT_RELU) {
                t = fmaxf(t, 0.f);
            } else if (p.act == SPAA_ACT_RELU_CLAMP1) {
                t = fmaxf(t, 0.f);
                if (p.aux_out != nullptr) io4<T>::st1(p.aux_out, oi, t);
                t = fminf(t, 1.f);
            } else if (p.act == SPAA_ACT_LEAKY01) {
                t = t > 0.f ? t : 0.1f * t;
            }
            if (p.gate != nullptr) {
                const float gv = io4<T>::ld1(p.gate, o * p.gate_cstride + p.gate_coff + n);
                const bool pass = (p.gate_mode == SPAA_GATE_POS_LE1) ? (gv > 0.f && gv <= 1.f) : (gv > 0.f);
                t = (p.gate_mode == SPAA_GATE_MUL) ? t * gv : (pass ? t : 0.f);
            }
            io4<T>::st1(p.out, oi, t);
            if (p.gate2 != nullptr) {
                const float g2 = io4<T>::ld1(p.gate2, o * p.gate2_cstride + p.gate2_coff + n);
                io4<T>::st1(p.aux_out, oi, (g2 > 0.f) ? t : 0.f);
            }
        }
    }
}

// The same epilogue in two steps, so that a kernel can have the operand loads of several pixels in flight before it
// finishes (and stores) any of them: epi_load() fetches what the vector path of store4_t reads besides the accumulators
// (residual, gates), store4_pre() is store4_t's vector path on those values.
struct epi_pre_t {
    f4 add, gate, gate2;
    unsigned int gbits, g2bits;
};

template <typename T>
__device__ __forceinline__ epi_pre_t epi_load(const spaa_tapconv_t& p, const size_t o, const int n0) {
    epi_pre_t r;
    r.add = r.gate = r.gate2 = f4{0.f, 0.f, 0.f, 0.f};
    r.gbits = r.g2bits = 0;
    if (n0 >= p.Cout) return r;
    if (p.add != nullptr) r.add = io4<T>::ld(p.add, o * p.add_cstride + p.add_coff + n0);
    if (p.gate != nullptr) r.gate = io4<T>::ld(p.gate, o * p.gate_cstride + p.gate_coff + n0);
    else if (p.gate_bits != nullptr) r.gbits = p.gate_bits[(o * p.gate_cstride + p.gate_coff + n0) >> 2];
    if (p.gate2 != nullptr) r.gate2 = io4<T>::ld(p.gate2, o * p.gate2_cstride + p.gate2_coff + n0);
    else if (p.gate2_bits != nullptr) r.g2bits = p.gate2_bits[(o * p.gate2_cstride + p.gate2_coff + n0) >> 2];
    return r;
}

template <typename T>
__device__ __forceinline__ void store4_pre(const spaa_tapconv_t& p, const size_t o, const int n0, float (&v)[4], const epi_pre_t& r) {
    if (n0 >= p.Cout) return;
    if (p.bias != nullptr) {
        const f4 bb = *reinterpret_cast<const f4*>(p.bias + n0);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
    }
    if (p.add != nullptr) { v[0] += r.add.x; v[1] += r.add.y; v[2] += r.add.z; v[3] += r.add.w; }
    const size_t oi = o * p.out_cstride + p.out_coff + n0;
    if (p.act == SPAA_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == SPAA_ACT_RELU_CLAMP1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        if (p.aux_out != nullptr) io4<T>::st(p.aux_out, oi, f4{v[0], v[1], v[2], v[3]});
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fminf(v[e], 1.f);
    } else if (p.act == SPAA_ACT_LEAKY01) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.1f * v[e];
    }
    if (p.gate != nullptr) {
        const float ga[4] = {r.gate.x, r.gate.y, r.gate.z, r.gate.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool pass = (p.gate_mode == SPAA_GATE_POS_LE1) ? (ga[e] > 0.f && ga[e] <= 1.f) : (ga[e] > 0.f);
            v[e] = (p.gate_mode == SPAA_GATE_MUL) ? v[e] * ga[e] : (pass ? v[e] : 0.f);
        }
    } else if (p.gate_bits != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((r.gbits >> e) & 1u) ? v[e] : 0.f;
    }
    if (sizeof(T) == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)(_Float16)v[e];
    }
    io4<T>::st(p.out, oi, f4{v[0], v[1], v[2], v[3]});
    if (p.mask_out != nullptr)
        p.mask_out[(o * p.out_cstride + p.out_coff + n0) >> 2] =
            (uint8_t)((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u));
    if (p.gate2 != nullptr) {
        io4<T>::st(p.aux_out, oi, f4{r.gate2.x > 0.f ? v[0] : 0.f, r.gate2.y > 0.f ? v[1] : 0.f, r.gate2.z > 0.f ? v[2] : 0.f,
                                    r.gate2.w > 0.f ? v[3] : 0.f});
    } else if (p.gate2_bits != nullptr) {
        io4<T>::st(p.aux_out, oi, f4{(r.g2bits & 1u) ? v[0] : 0.f, (r.g2bits & 2u) ? v[1] : 0.f, (r.g2bits & 4u) ? v[2] : 0.f,
                                    (r.g2bits & 8u) ? v[3] : 0.f});
    }
}

// ---- the same epilogue WITHOUT a branch, for the operand combinations of the attack loops (bias, residual, ReLU, byte-mask
// gate, byte mask out, second output gated by a second byte mask): absent tensors are buffer descriptors with ZERO records
// (loads give 0, stores are dropped) and out-of-range pixels / channels use the out-of-bounds offset, so that a kernel can keep
// the operand loads of several pixels per lane in flight -- the conditional loads of epi_load() make the compiler wait for
// each one (s_waitcnt vmcnt(0) at every join).  32-bit byte offsets: fits_32bit_offsets().
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct fast_epi_t {
    __amdgpu_buffer_rsrc_t out, add, gbits, g2bits, mask, aux, rbias;
    float bias[4];
    bool has_gate, relu;
    bool out_sc1;   // `out` is written with agent-scope (sc1) stores: K-range partial sums another XCD's workgroup reads back (fast_epi_store<.., SC1 = true>)
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_or_empty(const void* ptr, const int64_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const int n = ptr != nullptr ? (int)bytes : 0;
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(n), 0x00020000);
}
// (the launcher checks that every operand of the fast path is below 2 GiB: fits_32bit_offsets)
__device__ __forceinline__ bool fits_32bit_offsets(const spaa_tapconv_t& p) {
    const int64_t npix = (int64_t)p.B * p.Hout * p.Wout;
    const int64_t widest = p.io_dtype & SPAA_IO_OUT_F16 ? 2 : 4;
    const int cs = p.out_cstride > p.add_cstride ? p.out_cstride : p.add_cstride;
    const int cg = p.gate_cstride > p.gate2_cstride ? p.gate_cstride : p.gate2_cstride;
    return npix * (cs > cg ? cs : cg) * widest < ((int64_t)1 << 31);
}
__device__ __forceinline__ fast_epi_t make_fast_epi(const spaa_tapconv_t& p, const int n) {
    fast_epi_t f;
    const int64_t npix = (int64_t)p.B * p.Hout * p.Wout;
    const int64_t eb = p.io_dtype & SPAA_IO_OUT_F16 ? 2 : 4;
    f.out = rsrc_or_empty(p.out, npix * p.out_cstride * eb);
    f.aux = rsrc_or_empty(p.gate2_bits != nullptr ? p.aux_out : nullptr, npix * p.out_cstride * eb);
    f.add = rsrc_or_empty(p.add, npix * p.add_cstride * eb);
    f.gbits = rsrc_or_empty(p.gate_bits, npix * p.gate_cstride / 4);
    f.g2bits = rsrc_or_empty(p.gate2_bits, npix * p.gate2_cstride / 4);
    f.mask = rsrc_or_empty(p.mask_out, npix * p.out_cstride / 4);
    f.rbias = rsrc_or_empty(p.bias, (int64_t)p.Cout * 4);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(f.rbias, n * 4, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) f.bias[e] = __uint_as_float(b[e]);
    f.has_gate = p.gate_bits != nullptr;
    f.relu = p.act == SPAA_ACT_RELU;
    f.out_sc1 = false;
    return f;
}
template <typename T> struct fast_io;
template <> struct fast_io<_Float16> {
    typedef u32x2 vec_t;
    static __device__ __forceinline__ vec_t ld(const __amdgpu_buffer_rsrc_t r, const int off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
    static __device__ __forceinline__ void to_float(const vec_t x, float (&v)[4]) {
        const h4 h = __builtin_bit_cast(h4, x);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)h[e];
    }
    template <int AUX = 0>
    static __device__ __forceinline__ void st(const __amdgpu_buffer_rsrc_t r, const int off, const float (&v)[4]) {
        const h4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), r, off, 0, AUX);
    }
};
template <> struct fast_io<float> {
    typedef u32x4 vec_t;
    static __device__ __forceinline__ vec_t ld(const __amdgpu_buffer_rsrc_t r, const int off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
    static __device__ __forceinline__ void to_float(const vec_t x, float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(x[e]);
    }
    template <int AUX = 0>
    static __device__ __forceinline__ void st(const __amdgpu_buffer_rsrc_t r, const int off, const float (&v)[4]) {
        const u32x4 x = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        __builtin_amdgcn_raw_buffer_store_b128(x, r, off, 0, AUX);
    }
};

template <typename T>
struct fast_pre_t {
    typename fast_io<T>::vec_t add;
    u32x4 bias;            // (LOADB only)
    unsigned int gb, g2;
};
// eligibility (uniform): 4-channel vectors, no float gates, plain or ReLU activation, every operand below 2 GiB
__device__ __forceinline__ bool fast_epi_ok(const spaa_tapconv_t& p, const bool vec) {
    return vec && p.gate == nullptr && p.gate2 == nullptr && (p.act == SPAA_ACT_NONE || p.act == SPAA_ACT_RELU) && fits_32bit_offsets(p);
}
// operands of output pixel o (index over B x Hout x Wout), channels n .. n + 3; ok = the pixel / channels exist.
// LOADB: the lane's channels change from call to call (MFMA-layout epilogues): the bias is fetched per call instead of once
// per lane (make_fast_epi)
template <typename T, bool LOADB = false>
__device__ __forceinline__ fast_pre_t<T> fast_epi_load(const fast_epi_t& fe, const spaa_tapconv_t& p, const int o, const int n, const bool ok) {
    constexpr int OOB = (int)0x80000000;
    fast_pre_t<T> r;
    r.add = fast_io<T>::ld(fe.add, ok ? (o * p.add_cstride + p.add_coff + n) * (int)sizeof(T) : OOB);
    r.gb = __builtin_amdgcn_raw_buffer_load_b8(fe.gbits, ok ? (o * p.gate_cstride + p.gate_coff + n) >> 2 : OOB, 0, 0);
    r.g2 = __builtin_amdgcn_raw_buffer_load_b8(fe.g2bits, ok ? (o * p.gate2_cstride + p.gate2_coff + n) >> 2 : OOB, 0, 0);
    if constexpr (LOADB) r.bias = __builtin_amdgcn_raw_buffer_load_b128(fe.rbias, ok ? n * 4 : OOB, 0, 0);
    return r;
}
// store4_pre()'s arithmetic in its order: bias, residual, ReLU, gate, rounding to the storage type, out, mask of the STORED
// value, second output gated by the second mask
// SC1 (the canvas / K-range kernels): when fe.out_sc1 is set, `out` (a K range's partial sums) is written with agent-scope stores -- cache
// policy bit 4 = sc1 on gfx950: written through to the level all XCDs share, so that the workgroup that adds the K ranges, on whichever
// XCD it runs, reads them back (with sc1 loads) without any cache flush
constexpr int SPAA_AUX_SC1 = 16;
template <typename T, typename ACC, bool LOADB = false, bool SC1 = false>
__device__ __forceinline__ void fast_epi_store(const fast_epi_t& fe, const spaa_tapconv_t& p, const int o, const int n, const bool ok,
                                               const ACC& a, const fast_pre_t<T>& r) {
    constexpr int OOB = (int)0x80000000;
    const int oi = o * p.out_cstride + p.out_coff + n;
    float ad[4], v[4], u[4];
    fast_io<T>::to_float(r.add, ad);
    const unsigned int g = fe.has_gate ? r.gb : 15u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = a[e] + (LOADB ? __uint_as_float(r.bias[e]) : fe.bias[e]);
        v[e] += ad[e];
        v[e] = fe.relu ? fmaxf(v[e], 0.f) : v[e];
        v[e] = ((g >> e) & 1u) ? v[e] : 0.f;
        v[e] = (float)(T)v[e];
        u[e] = ((r.g2 >> e) & 1u) ? v[e] : 0.f;
    }
    if constexpr (SC1) {
        if (fe.out_sc1) fast_io<T>::template st<SPAA_AUX_SC1>(fe.out, ok ? oi * (int)sizeof(T) : OOB, v);
        else fast_io<T>::st(fe.out, ok ? oi * (int)sizeof(T) : OOB, v);
    } else {
        fast_io<T>::st(fe.out, ok ? oi * (int)sizeof(T) : OOB, v);
    }
    const unsigned char mb = (unsigned char)((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u));
    __builtin_amdgcn_raw_buffer_store_b8(mb, fe.mask, ok ? oi >> 2 : OOB, 0, 0);
    fast_io<T>::st(fe.aux, ok ? oi * (int)sizeof(T) : OOB, u);
}

// storage type chosen at run time (kernels that read fp32 IMAGES and may write fp16 activations: smallcin, x6v2/v3; the
// fp32-only bf16x6 kernels call store4_t<float> directly and keep their register budget)
__device__ __forceinline__ void store4(const spaa_tapconv_t& p, const size_t o, const int n0, float (&v)[4], const bool vec) {
    if (p.io_dtype & SPAA_IO_OUT_F16) store4_t<_Float16>(p, o, n0, v, vec);
    else store4_t<float>(p, o, n0, v, vec);
}

// N-folded stride-2 transposed convolution: GEMM row group n0 -> (parity class c, channel n) and the class's pixel
template <typename T>
__device__ __forceinline__ void store4_fold_t(const spaa_tapconv_t& p, const int m, const int M, const int HWm, const int n0,
                                              float (&v)[4], const bool vec) {
    if (m >= M) return;
    const int c = n0 / p.Cout;
    if (c >= p.nfold) return;
    const int b = m / HWm;
    const int rr = m - b * HWm;
    const int y = rr / p.Wm;
    const int x = rr - y * p.Wm;
    const int oy = 2 * y + (c >> 1), ox = 2 * x + (c & 1);
    if (oy >= p.Hout || ox >= p.Wout) return;
    store4_t<T>(p, ((size_t)b * p.Hout + oy) * p.Wout + ox, n0 - c * p.Cout, v, vec);
}

// the same mapping for the branch-free epilogue: output pixel o and channel n of GEMM (row m, column quad n0); false when absent
__device__ __forceinline__ bool fold_pixel(const spaa_tapconv_t& p, const int m, const int M, const int HWm, const int n0, int& o, int& n) {
    const int c = n0 / p.Cout;
    const int b = m / HWm;
    const int rr = m - b * HWm;
    const int y = rr / p.Wm;
    const int x = rr - y * p.Wm;
    const int oy = 2 * y + (c >> 1), ox = 2 * x + (c & 1);
    o = (b * p.Hout + oy) * p.Wout + ox;
    n = n0 - c * p.Cout;
    return m < M && c < p.nfold && oy < p.Hout && ox < p.Wout;
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
