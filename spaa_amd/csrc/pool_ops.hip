// pool_ops.hip — generic NHWC pooling layers of the torchvision classifier bodies (VGG-16, Inception-v3), forward and
// input-gradient.  Backward passes are gathers (each input element sums the windows that cover it): deterministic,
// no atomics.  Replaces max_pool2d / avg_pool2d / adaptive_avg_pool2d behind classifier.py:60 and their autograd.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"  // io4<T>: 4 consecutive elements stored as fp32 or fp16

namespace {

__device__ __forceinline__ int a_start(int i, int out, int in) { return (int)floorf((float)(i * in) / (float)out); }
__device__ __forceinline__ int a_end(int i, int out, int in) { return (int)ceilf((float)((i + 1) * in) / (float)out); }

struct Geo {
    int B, Hin, Win, C4, Hout, Wout, k, s, p;
};

// max_pool2d(k, s, p), first maximum in row-major window order wins (ATen CPU rule); 4 channels per thread
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ in, T* __restrict__ out,
                                   uchar4* __restrict__ argmax, Geo g, int out_c4stride, int out_c4off) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * g.Hout * g.Wout * g.C4) return;
    const int c = idx % g.C4;
    int r = idx / g.C4;
    const int ox = r % g.Wout;
    r /= g.Wout;
    const int oy = r % g.Hout;
    const int b = r / g.Hout;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 am = make_uchar4(0, 0, 0, 0);
    for (int ky = 0; ky < g.k; ++ky) {
        const int iy = oy * g.s - g.p + ky;
        if ((unsigned)iy >= (unsigned)g.Hin) continue;
        for (int kx = 0; kx < g.k; ++kx) {
            const int ix = ox * g.s - g.p + kx;
            if ((unsigned)ix >= (unsigned)g.Win) continue;
            const f4 v = io4<T>::ld(in, 4 * ((((size_t)b * g.Hin + iy) * g.Win + ix) * g.C4 + c));
            const unsigned char kk = (unsigned char)(ky * g.k + kx);
            if (v.x > best.x || v.x != v.x) { best.x = v.x; am.x = kk; }
            if (v.y > best.y || v.y != v.y) { best.y = v.y; am.y = kk; }
            if (v.z > best.z || v.z != v.z) { best.z = v.z; am.z = kk; }
            if (v.w > best.w || v.w != v.w) { best.w = v.w; am.w = kk; }
        }
    }
    // bit 7: the maximum is positive — the ReLU gate of the pooled tensor's producer, for the backward pass
    am.x |= best.x > 0.f ? 0x80 : 0;
    am.y |= best.y > 0.f ? 0x80 : 0;
    am.z |= best.z > 0.f ? 0x80 : 0;
    am.w |= best.w > 0.f ? 0x80 : 0;
    io4<T>::st(out, 4 * ((((size_t)b * g.Hout + oy) * g.Wout + ox) * out_c4stride + out_c4off + c),
               f4{best.x, best.y, best.z, best.w});
    argmax[idx] = am;
}

// kernel 3 / stride 2 / padding 1 (the ResNet stem pool), a thread per 2 x 2 block of input pixels: the block draws on four
// windows, so four (arg-max byte, gradient) pairs serve four outputs (classifier_ops.hip: maxpool_bwd_quad_kernel); same order of
// additions as the per-pixel form below
template <typename T>
__global__ void maxpool_bwd_quad_kernel(const T* __restrict__ g_out, const uchar4* __restrict__ argmax, const int relu_gate,
                                        T* __restrict__ g_in, Geo g, int gout_c4stride, int gout_c4off, int Hq, int Wq) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * Hq * Wq * g.C4) return;
    const int c = idx % g.C4;
    int r = idx / g.C4;
    const int bq = r % Wq;
    r /= Wq;
    const int a = r % Hq;
    const int b = r / Hq;
    const unsigned char need = relu_gate ? 0x80 : 0x00;
    uchar4 am[2][2];
    f4 gv[2][2];
    bool wok[2][2];
#pragma unroll
    for (int p_ = 0; p_ < 2; ++p_)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wok[p_][q] = a + p_ < g.Hout && bq + q < g.Wout;
            const size_t opix = ((size_t)b * g.Hout + (wok[p_][q] ? a + p_ : 0)) * g.Wout + (wok[p_][q] ? bq + q : 0);
            am[p_][q] = argmax[opix * g.C4 + c];
            gv[p_][q] = io4<T>::ld(g_out, 4 * (opix * gout_c4stride + gout_c4off + c));
        }
    auto take = [&](f4& acc, const int p_, const int q, const unsigned char k) {
        const uchar4 m = am[p_][q];
        const f4 v = gv[p_][q];
        if (wok[p_][q] && (m.x & 0x7f) == k && (m.x & need) == need) acc.x += v.x;
        if (wok[p_][q] && (m.y & 0x7f) == k && (m.y & need) == need) acc.y += v.y;
        if (wok[p_][q] && (m.z & 0x7f) == k && (m.z & need) == need) acc.z += v.z;
        if (wok[p_][q] && (m.w & 0x7f) == k && (m.w & need) == need) acc.w += v.w;
    };
    const int iy = 2 * a, ix = 2 * bq;
    const size_t base = 4 * ((((size_t)b * g.Hin + iy) * g.Win + ix) * g.C4 + c);
    const size_t dx = 4 * (size_t)g.C4, dy = 4 * (size_t)g.Win * g.C4;
    {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        take(acc, 0, 0, 4);
        io4<T>::st(g_in, base, acc);
    }
    if (ix + 1 < g.Win) {
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        take(acc, 0, 1, 3);
        take(acc, 0, 0, 5);
        io4<T>::st(g_in, base + dx, acc);
    }
    if (iy + 1 < g.Hin) {
        {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            take(acc, 1, 0, 1);
            take(acc, 0, 0, 7);
            io4<T>::st(g_in, base + dy, acc);
        }
        if (ix + 1 < g.Win) {
            f4 acc = {0.f, 0.f, 0.f, 0.f};
            take(acc, 1, 1, 0);
            take(acc, 1, 0, 2);
            take(acc, 0, 1, 6);
            take(acc, 0, 0, 8);
            io4<T>::st(g_in, base + dy + dx, acc);
        }
    }
}

// CK/CS/CP > 0: kernel / stride / padding known at compile time (3 / 2 / 1: the ResNet stem pool in fp16 storage -- the runtime
// divisions of the generic form cost more than the loads)
template <typename T, int CK = 0, int CS = 0, int CP = 0>
__global__ void maxpool_bwd_kernel(const T* __restrict__ g_out, const uchar4* __restrict__ argmax,
                                   const int relu_gate, T* __restrict__ g_in, Geo g,
                                   int gout_c4stride, int gout_c4off) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * g.Hin * g.Win * g.C4) return;
    const int c = idx % g.C4;
    int r = idx / g.C4;
    const int ix = r % g.Win;
    r /= g.Win;
    const int iy = r % g.Hin;
    const int b = r / g.Hin;
    const int K = CK ? CK : g.k, S = CS ? CS : g.s, P = CK ? CP : g.p;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned char need = relu_gate ? 0x80 : 0x00;  // relu_gate: only windows whose maximum is positive pass
    if constexpr (CK == 3 && CS == 2 && CP == 1) {
        // branch-free (classifier_ops.hip: maxpool_bwd_kernel): the two candidate output rows / columns, four unconditional loads
        int oyc[2], kyc[2], oxc[2], kxc[2];
        bool yok[2], xok[2];
        if (iy & 1) { oyc[0] = (iy + 1) >> 1; kyc[0] = 0; oyc[1] = (iy - 1) >> 1; kyc[1] = 2; yok[0] = oyc[0] < g.Hout; yok[1] = true; }
        else        { oyc[0] = iy >> 1; kyc[0] = 1; oyc[1] = 0; kyc[1] = 0; yok[0] = oyc[0] < g.Hout; yok[1] = false; }
        if (ix & 1) { oxc[0] = (ix + 1) >> 1; kxc[0] = 0; oxc[1] = (ix - 1) >> 1; kxc[1] = 2; xok[0] = oxc[0] < g.Wout; xok[1] = true; }
        else        { oxc[0] = ix >> 1; kxc[0] = 1; oxc[1] = 0; kxc[1] = 0; xok[0] = oxc[0] < g.Wout; xok[1] = false; }
        uchar4 am[4];
        f4 gv[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const size_t opix = ((size_t)b * g.Hout + (yok[i] ? oyc[i] : 0)) * g.Wout + (xok[j] ? oxc[j] : 0);
                am[2 * i + j] = argmax[opix * g.C4 + c];
                gv[2 * i + j] = io4<T>::ld(g_out, 4 * (opix * gout_c4stride + gout_c4off + c));
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = yok[i] && xok[j];
                const unsigned char kk = (unsigned char)(kyc[i] * 3 + kxc[j]);
                const uchar4 a = am[2 * i + j];
                const f4 go = gv[2 * i + j];
                if (ok && (a.x & 0x7f) == kk && (a.x & need) == need) acc.x += go.x;
                if (ok && (a.y & 0x7f) == kk && (a.y & need) == need) acc.y += go.y;
                if (ok && (a.z & 0x7f) == kk && (a.z & need) == need) acc.z += go.z;
                if (ok && (a.w & 0x7f) == kk && (a.w & need) == need) acc.w += go.w;
            }
        io4<T>::st(g_in, 4 * (size_t)idx, f4{acc.x, acc.y, acc.z, acc.w});
        return;
    }
    for (int ky = 0; ky < K; ++ky) {
        const int t = iy + P - ky;
        if (t < 0 || (t % S)) continue;
        const int oy = t / S;
        if (oy >= g.Hout) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int u = ix + P - kx;
            if (u < 0 || (u % S)) continue;
            const int ox = u / S;
            if (ox >= g.Wout) continue;
            const size_t opix = ((size_t)b * g.Hout + oy) * g.Wout + ox;
            const uchar4 am = argmax[opix * g.C4 + c];
            const f4 go = io4<T>::ld(g_out, 4 * (opix * gout_c4stride + gout_c4off + c));
            const unsigned char kk = (unsigned char)(ky * K + kx);
            if ((am.x & 0x7f) == kk && (am.x & need) == need) acc.x += go.x;
            if ((am.y & 0x7f) == kk && (am.y & need) == need) acc.y += go.y;
            if ((am.z & 0x7f) == kk && (am.z & need) == need) acc.z += go.z;
            if ((am.w & 0x7f) == kk && (am.w & need) == need) acc.w += go.w;
        }
    }
    io4<T>::st(g_in, 4 * (size_t)idx, f4{acc.x, acc.y, acc.z, acc.w});
}

// avg_pool2d(k, s, p), count_include_pad=True (divisor k*k), optional output channel window of a concat buffer
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, Geo g, int out_c4stride, int out_c4off) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * g.Hout * g.Wout * g.C4) return;
    const int c = idx % g.C4;
    int r = idx / g.C4;
    const int ox = r % g.Wout;
    r /= g.Wout;
    const int oy = r % g.Hout;
    const int b = r / g.Hout;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < g.k; ++ky) {
        const int iy = oy * g.s - g.p + ky;
        if ((unsigned)iy >= (unsigned)g.Hin) continue;
        for (int kx = 0; kx < g.k; ++kx) {
            const int ix = ox * g.s - g.p + kx;
            if ((unsigned)ix >= (unsigned)g.Win) continue;
            const f4 v = io4<T>::ld(in, 4 * ((((size_t)b * g.Hin + iy) * g.Win + ix) * g.C4 + c));
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    const float d = (float)(g.k * g.k);
    io4<T>::st(out, 4 * ((((size_t)b * g.Hout + oy) * g.Wout + ox) * out_c4stride + out_c4off + c),
               f4{a.x / d, a.y / d, a.z / d, a.w / d});
}

template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ g_out, T* __restrict__ g_in, Geo g, int gout_c4stride, int gout_c4off) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * g.Hin * g.Win * g.C4) return;
    const int c = idx % g.C4;
    int r = idx / g.C4;
    const int ix = r % g.Win;
    r /= g.Win;
    const int iy = r % g.Hin;
    const int b = r / g.Hin;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < g.k; ++ky) {
        const int t = iy + g.p - ky;
        if (t < 0 || (t % g.s)) continue;
        const int oy = t / g.s;
        if (oy >= g.Hout) continue;
        for (int kx = 0; kx < g.k; ++kx) {
            const int u = ix + g.p - kx;
            if (u < 0 || (u % g.s)) continue;
            const int ox = u / g.s;
            if (ox >= g.Wout) continue;
            const f4 go = io4<T>::ld(g_out, 4 * ((((size_t)b * g.Hout + oy) * g.Wout + ox) * gout_c4stride + gout_c4off + c));
            acc.x += go.x; acc.y += go.y; acc.z += go.z; acc.w += go.w;
        }
    }
    const float d = (float)(g.k * g.k);
    io4<T>::st(g_in, 4 * (size_t)idx, f4{acc.x / d, acc.y / d, acc.z / d, acc.w / d});
}

// adaptive_avg_pool2d to (Hout, Wout) with ATen's window rule
__global__ void adaptive_fwd_kernel(const float4* __restrict__ in, float4* __restrict__ out, Geo g) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * g.Hout * g.Wout * g.C4) return;
    const int c = idx % g.C4;
    int r = idx / g.C4;
    const int ox = r % g.Wout;
    r /= g.Wout;
    const int oy = r % g.Hout;
    const int b = r / g.Hout;
    const int ys = a_start(oy, g.Hout, g.Hin), ye = a_end(oy, g.Hout, g.Hin);
    const int xs = a_start(ox, g.Wout, g.Win), xe = a_end(ox, g.Wout, g.Win);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = ys; iy < ye; ++iy)
        for (int ix = xs; ix < xe; ++ix) {
            const float4 v = in[(((size_t)b * g.Hin + iy) * g.Win + ix) * g.C4 + c];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    const float d = (float)((ye - ys) * (xe - xs));
    out[idx] = make_float4(a.x / d, a.y / d, a.z / d, a.w / d);
}

__global__ void adaptive_bwd_kernel(const float4* __restrict__ g_out, const float4* __restrict__ gate_in,
                                    float4* __restrict__ g_in, Geo g) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * g.Hin * g.Win * g.C4) return;
    const int c = idx % g.C4;
    int r = idx / g.C4;
    const int ix = r % g.Win;
    r /= g.Win;
    const int iy = r % g.Hin;
    const int b = r / g.Hin;
    const int oy_lo = max(0, (iy * g.Hout) / g.Hin - 1), oy_hi = min(g.Hout - 1, ((iy + 1) * g.Hout + g.Hin - 1) / g.Hin + 1);
    const int ox_lo = max(0, (ix * g.Wout) / g.Win - 1), ox_hi = min(g.Wout - 1, ((ix + 1) * g.Wout + g.Win - 1) / g.Win + 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const int ys = a_start(oy, g.Hout, g.Hin), ye = a_end(oy, g.Hout, g.Hin);
        if (iy < ys || iy >= ye) continue;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const int xs = a_start(ox, g.Wout, g.Win), xe = a_end(ox, g.Wout, g.Win);
            if (ix < xs || ix >= xe) continue;
            const float inv = 1.f / (float)((ye - ys) * (xe - xs));
            const float4 go = g_out[(((size_t)b * g.Hout + oy) * g.Wout + ox) * g.C4 + c];
            acc.x += go.x * inv; acc.y += go.y * inv; acc.z += go.z * inv; acc.w += go.w * inv;
        }
    }
    if (gate_in != nullptr) {
        const float4 a = gate_in[idx];
        acc.x = a.x > 0.f ? acc.x : 0.f;
        acc.y = a.y > 0.f ? acc.y : 0.f;
        acc.z = a.z > 0.f ? acc.z : 0.f;
        acc.w = a.w > 0.f ? acc.w : 0.f;
    }
    g_in[idx] = acc;
}

// ReLU-gate bytes of a channel window of an activation (the `mask_out` format of spaa_tapconv_t: one byte per 4 channels, bit e =
// channel 4 q + e > 0) for activations that no convolution launch wrote: the max-pool outputs of the Inception-v3 body, which
// gate the layers that consume them like every other ReLU output (spaa_amd/inception.py).
template <typename T>
__global__ __launch_bounds__(256) void gate_mask_kernel(const T* __restrict__ act, uint8_t* __restrict__ mask, const int64_t M,
                                                        const int C4, const int cstride, const int coff) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * C4) return;
    const int64_t m = idx / C4;
    const int q = (int)(idx - m * C4);
    const T* a = act + m * cstride + coff + 4 * q;
    const unsigned int b = ((float)a[0] > 0.f ? 1u : 0u) | ((float)a[1] > 0.f ? 2u : 0u) | ((float)a[2] > 0.f ? 4u : 0u) |
                           ((float)a[3] > 0.f ? 8u : 0u);
    mask[(m * cstride + coff) / 4 + q] = (uint8_t)b;
}

// kernel 2 / stride 2 / padding 0 on even image sides (VGG-16's five pools, classifier.py:21-24): the windows do not overlap, so a
// thread owns ONE window of VW = 16 / sizeof(T) channels -- 16-byte loads and stores only, no divisions inside, and the backward
// pass writes the window's four input pixels from one (gradient, arg-max) pair instead of searching the windows that cover an
// input pixel (the generic gather: 125 us per launch on average in VGG-16's fp16 loop against ~35 at the HBM rate).  Same rule as
// maxpool_fwd_kernel: first maximum in row-major window order, NaN wins; arg-max byte = code | 0x80 when the maximum is positive.
template <typename T>
struct vec16 {
    static constexpr int VW = 16 / sizeof(T);
    T v[VW];
};
template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, uint8_t* __restrict__ argmax,
                                                             const int64_t n, const int Hout, const int Wout, const int CV,
                                                             const int out_cstride, const int out_coff) {
    constexpr int VW = vec16<T>::VW;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % CV);
    int64_t r = idx / CV;
    const int ox = (int)(r % Wout);
    r /= Wout;
    const int oy = (int)(r % Hout);
    const int64_t b = r / Hout;
    const int C = CV * VW, Win = 2 * Wout;
    const T* src = in + (((b * 2 * Hout + 2 * oy) * Win + 2 * ox) * C + c * VW);
    typedef vec16<T> V;
    const V w00 = *reinterpret_cast<const V*>(src), w01 = *reinterpret_cast<const V*>(src + C);
    const V w10 = *reinterpret_cast<const V*>(src + (int64_t)Win * C), w11 = *reinterpret_cast<const V*>(src + (int64_t)Win * C + C);
    V best;
    uint8_t am[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) {
        float bv = (float)w00.v[e];
        uint8_t k = 0;
        // (the generic kernel starts from -inf: the first element always wins its comparison unless it IS -inf, in which case
        // code 0 is kept as well)
        const float v1 = (float)w01.v[e], v2 = (float)w10.v[e], v3 = (float)w11.v[e];
        if (bv != bv) { /* NaN stays unless a later NaN replaces it */ }
        if (v1 > bv || v1 != v1) { bv = v1; k = 1; }
        if (v2 > bv || v2 != v2) { bv = v2; k = 2; }
        if (v3 > bv || v3 != v3) { bv = v3; k = 3; }
        best.v[e] = (T)bv;
        am[e] = (uint8_t)(k | (bv > 0.f ? 0x80 : 0));
    }
    const int64_t opix = (b * Hout + oy) * Wout + ox;
    *reinterpret_cast<V*>(out + opix * out_cstride + out_coff + c * VW) = best;
    uint8_t* ap = argmax + opix * C + c * VW;
    if constexpr (VW == 8) *reinterpret_cast<uint2*>(ap) = *reinterpret_cast<const uint2*>(am);
    else *reinterpret_cast<uint32_t*>(ap) = *reinterpret_cast<const uint32_t*>(am);
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const T* __restrict__ g_out, const uint8_t* __restrict__ argmax,
                                                             const int relu_gate, T* __restrict__ g_in, const int64_t n,
                                                             const int Hout, const int Wout, const int CV, const int gout_cstride,
                                                             const int gout_coff) {
    constexpr int VW = vec16<T>::VW;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % CV);
    int64_t r = idx / CV;
    const int ox = (int)(r % Wout);
    r /= Wout;
    const int oy = (int)(r % Hout);
    const int64_t b = r / Hout;
    const int C = CV * VW, Win = 2 * Wout;
    typedef vec16<T> V;
    const int64_t opix = (b * Hout + oy) * Wout + ox;
    const V go = *reinterpret_cast<const V*>(g_out + opix * gout_cstride + gout_coff + c * VW);
    uint8_t am[VW];
    const uint8_t* ap = argmax + opix * C + c * VW;
    if constexpr (VW == 8) *reinterpret_cast<uint2*>(am) = *reinterpret_cast<const uint2*>(ap);
    else *reinterpret_cast<uint32_t*>(am) = *reinterpret_cast<const uint32_t*>(ap);
    const uint8_t need = relu_gate ? 0x80 : 0x00;
    T* dst = g_in + (((b * 2 * Hout + 2 * oy) * Win + 2 * ox) * C + c * VW);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        V o;
#pragma unroll
        for (int e = 0; e < VW; ++e) o.v[e] = ((am[e] & 0x7f) == k && (am[e] & need) == need) ? go.v[e] : (T)0.f;
        *reinterpret_cast<V*>(dst + (int64_t)(k >> 1) * Win * C + (k & 1) * C) = o;
    }
}

inline int nb(int64_t n) { return (int)((n + 255) / 256); }

inline bool geo_ok(int B, int Hin, int Win, int C, int Hout, int Wout, int k, int s, int p) {
    return B > 0 && Hin > 0 && Win > 0 && C > 0 && !(C & 3) && k > 0 && k <= 15 && s > 0 && p >= 0 && 2 * p <= k &&
           Hout == (Hin + 2 * p - k) / s + 1 && Wout == (Win + 2 * p - k) / s + 1 &&
           (int64_t)B * Hin * Win * C < ((int64_t)1 << 31);
}

}  // namespace

extern "C" {

int spaa_maxpool_fwd(const float* in, float* out, uint8_t* argmax, int B, int Hin, int Win, int C, int Hout, int Wout,
                     int k, int s, int p, int out_cstride, int out_coff, spaa_stream_t stream) {
    if (!in || !out || !argmax || !geo_ok(B, Hin, Win, C, Hout, Wout, k, s, p) || (out_cstride & 3) || (out_coff & 3) ||
        out_coff + C > out_cstride)
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, k, s, p};
    if (k == 2 && s == 2 && p == 0 && Hin == 2 * Hout && Win == 2 * Wout) {   // non-overlapping windows: a thread per window
        const int64_t n = (int64_t)B * Hout * Wout * (C / 4);
        hipLaunchKernelGGL(maxpool2x2_fwd_kernel<float>, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, in, out, argmax, n, Hout, Wout,
                           C / 4, out_cstride, out_coff);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(nb((int64_t)B * Hout * Wout * g.C4)), dim3(256), 0, (hipStream_t)stream,
                       in, out, (uchar4*)argmax, g, out_cstride / 4, out_coff / 4);
    return (int)hipGetLastError();
}

int spaa_maxpool_fwd_f16(const void* in, void* out, uint8_t* argmax, int B, int Hin, int Win, int C, int Hout, int Wout,
                         int k, int s, int p, int out_cstride, int out_coff, spaa_stream_t stream) {
    if (!in || !out || !argmax || !geo_ok(B, Hin, Win, C, Hout, Wout, k, s, p) || (out_cstride & 3) || (out_coff & 3) ||
        out_coff + C > out_cstride)
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, k, s, p};
    if (k == 2 && s == 2 && p == 0 && Hin == 2 * Hout && Win == 2 * Wout && !(C & 7) && !(out_cstride & 7) && !(out_coff & 7)) {
        const int64_t n = (int64_t)B * Hout * Wout * (C / 8);
        hipLaunchKernelGGL(maxpool2x2_fwd_kernel<_Float16>, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)in,
                           (_Float16*)out, argmax, n, Hout, Wout, C / 8, out_cstride, out_coff);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel<_Float16>, dim3(nb((int64_t)B * Hout * Wout * g.C4)), dim3(256), 0,
                       (hipStream_t)stream, (const _Float16*)in, (_Float16*)out, (uchar4*)argmax, g, out_cstride / 4,
                       out_coff / 4);
    return (int)hipGetLastError();
}

int spaa_maxpool_bwd(const float* g_out, const uint8_t* argmax, int relu_gate, float* g_in, int B, int Hin,
                     int Win, int C, int Hout, int Wout, int k, int s, int p, int gout_cstride, int gout_coff,
                     spaa_stream_t stream) {
    if (!g_out || !argmax || !g_in || !geo_ok(B, Hin, Win, C, Hout, Wout, k, s, p) || (gout_cstride & 3) ||
        (gout_coff & 3) || gout_coff + C > gout_cstride)
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, k, s, p};
    if (k == 2 && s == 2 && p == 0 && Hin == 2 * Hout && Win == 2 * Wout) {
        const int64_t n = (int64_t)B * Hout * Wout * (C / 4);
        hipLaunchKernelGGL(maxpool2x2_bwd_kernel<float>, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, g_out, argmax, relu_gate, g_in, n,
                           Hout, Wout, C / 4, gout_cstride, gout_coff);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(nb((int64_t)B * Hin * Win * g.C4)), dim3(256), 0, (hipStream_t)stream,
                       g_out, (const uchar4*)argmax, relu_gate, g_in, g, gout_cstride / 4, gout_coff / 4);
    return (int)hipGetLastError();
}

int spaa_maxpool_bwd_f16(const void* g_out, const uint8_t* argmax, int relu_gate, void* g_in, int B, int Hin, int Win,
                         int C, int Hout, int Wout, int k, int s, int p, int gout_cstride, int gout_coff,
                         spaa_stream_t stream) {
    if (!g_out || !argmax || !g_in || !geo_ok(B, Hin, Win, C, Hout, Wout, k, s, p) || (gout_cstride & 3) ||
        (gout_coff & 3) || gout_coff + C > gout_cstride)
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, k, s, p};
    if (k == 2 && s == 2 && p == 0 && Hin == 2 * Hout && Win == 2 * Wout && !(C & 7) && !(gout_cstride & 7) && !(gout_coff & 7)) {
        const int64_t n = (int64_t)B * Hout * Wout * (C / 8);
        hipLaunchKernelGGL(maxpool2x2_bwd_kernel<_Float16>, dim3(nb(n)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)g_out, argmax,
                           relu_gate, (_Float16*)g_in, n, Hout, Wout, C / 8, gout_cstride, gout_coff);
        return (int)hipGetLastError();
    }
    if (k == 3 && s == 2 && p == 1) {
        const int Hq = (Hin + 1) / 2, Wq = (Win + 1) / 2;
        hipLaunchKernelGGL(maxpool_bwd_quad_kernel<_Float16>, dim3(nb((int64_t)B * Hq * Wq * g.C4)), dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)g_out, (const uchar4*)argmax, relu_gate, (_Float16*)g_in, g, gout_cstride / 4, gout_coff / 4,
                           Hq, Wq);
    } else
        hipLaunchKernelGGL(maxpool_bwd_kernel<_Float16>, dim3(nb((int64_t)B * Hin * Win * g.C4)), dim3(256), 0,
                           (hipStream_t)stream, (const _Float16*)g_out, (const uchar4*)argmax, relu_gate, (_Float16*)g_in, g,
                           gout_cstride / 4, gout_coff / 4);
    return (int)hipGetLastError();
}

int spaa_avgpool2d_fwd(const float* in, float* out, int B, int Hin, int Win, int C, int Hout, int Wout, int k, int s,
                       int p, int out_cstride, int out_coff, spaa_stream_t stream) {
    if (!in || !out || !geo_ok(B, Hin, Win, C, Hout, Wout, k, s, p) || (out_cstride & 3) || (out_coff & 3) ||
        out_coff + C > out_cstride)
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, k, s, p};
    hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(nb((int64_t)B * Hout * Wout * g.C4)), dim3(256), 0, (hipStream_t)stream,
                       in, out, g, out_cstride / 4, out_coff / 4);
    return (int)hipGetLastError();
}

int spaa_avgpool2d_fwd_f16(const void* in, void* out, int B, int Hin, int Win, int C, int Hout, int Wout, int k, int s,
                           int p, int out_cstride, int out_coff, spaa_stream_t stream) {
    if (!in || !out || !geo_ok(B, Hin, Win, C, Hout, Wout, k, s, p) || (out_cstride & 3) || (out_coff & 3) ||
        out_coff + C > out_cstride)
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, k, s, p};
    hipLaunchKernelGGL(avgpool_fwd_kernel<_Float16>, dim3(nb((int64_t)B * Hout * Wout * g.C4)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)in, (_Float16*)out, g, out_cstride / 4, out_coff / 4);
    return (int)hipGetLastError();
}

int spaa_avgpool2d_bwd(const float* g_out, float* g_in, int B, int Hin, int Win, int C, int Hout, int Wout, int k,
                       int s, int p, int gout_cstride, int gout_coff, spaa_stream_t stream) {
    if (!g_out || !g_in || !geo_ok(B, Hin, Win, C, Hout, Wout, k, s, p) || (gout_cstride & 3) || (gout_coff & 3) ||
        gout_coff + C > gout_cstride)
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, k, s, p};
    hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(nb((int64_t)B * Hin * Win * g.C4)), dim3(256), 0, (hipStream_t)stream,
                       g_out, g_in, g, gout_cstride / 4, gout_coff / 4);
    return (int)hipGetLastError();
}

int spaa_avgpool2d_bwd_f16(const void* g_out, void* g_in, int B, int Hin, int Win, int C, int Hout, int Wout, int k,
                           int s, int p, int gout_cstride, int gout_coff, spaa_stream_t stream) {
    if (!g_out || !g_in || !geo_ok(B, Hin, Win, C, Hout, Wout, k, s, p) || (gout_cstride & 3) || (gout_coff & 3) ||
        gout_coff + C > gout_cstride)
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, k, s, p};
    hipLaunchKernelGGL(avgpool_bwd_kernel<_Float16>, dim3(nb((int64_t)B * Hin * Win * g.C4)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)g_out, (_Float16*)g_in, g, gout_cstride / 4, gout_coff / 4);
    return (int)hipGetLastError();
}

int spaa_adaptive_avgpool_fwd(const float* in, float* out, int B, int Hin, int Win, int C, int Hout, int Wout,
                              spaa_stream_t stream) {
    if (!in || !out || B < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1 || C < 1 || (C & 3))
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, 0, 0, 0};
    hipLaunchKernelGGL(adaptive_fwd_kernel, dim3(nb((int64_t)B * Hout * Wout * g.C4)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)in, (float4*)out, g);
    return (int)hipGetLastError();
}

int spaa_adaptive_avgpool_bwd(const float* g_out, const float* gate_in, float* g_in, int B, int Hin, int Win, int C,
                              int Hout, int Wout, spaa_stream_t stream) {
    if (!g_out || !g_in || B < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1 || C < 1 || (C & 3))
        return hipErrorInvalidValue;
    Geo g{B, Hin, Win, C / 4, Hout, Wout, 0, 0, 0};
    hipLaunchKernelGGL(adaptive_bwd_kernel, dim3(nb((int64_t)B * Hin * Win * g.C4)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)g_out, (const float4*)gate_in, (float4*)g_in, g);
    return (int)hipGetLastError();
}

int spaa_gate_mask(const void* act, int f16, uint8_t* mask, int64_t M, int C, int cstride, int coff, spaa_stream_t stream) {
    if (!act || !mask || M < 1 || C < 4 || (C & 3) || (cstride & 3) || (coff & 3) || coff + C > cstride ||
        M * (int64_t)(C / 4) >= ((int64_t)1 << 31) * 256)
        return hipErrorInvalidValue;
    if (f16)
        hipLaunchKernelGGL(gate_mask_kernel<_Float16>, dim3(nb(M * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)act, mask, M, C / 4, cstride, coff);
    else
        hipLaunchKernelGGL(gate_mask_kernel<float>, dim3(nb(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const float*)act,
                           mask, M, C / 4, cstride, coff);
    return (int)hipGetLastError();
}

}  // extern "C"
