// classifier_ops.hip — the non-convolution pieces of the classifier path, NHWC fp32.
//
// Replaces (paths relative to /root/reference/src/python):
//   center_crop + F.interpolate(mode='area') + T.Normalize   classifier.py:59, img_proc.py:117-132, and autograd
//   torchvision max_pool2d(3,2,1) / adaptive_avg_pool2d(1)   classifier.py:60 (third-party net bodies), and autograd
// 'area' interpolation is adaptive average pooling with ATen's window rule
//   start = floor(i*in/out), end = ceil((i+1)*in/out)  (float arithmetic, as in ATen's start_index/end_index).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"

namespace {

__device__ __forceinline__ int win_start(int i, int out, int in) {
    return (int)floorf((float)(i * in) / (float)out);
}
__device__ __forceinline__ int win_end(int i, int out, int in) {
    return (int)ceilf((float)((i + 1) * in) / (float)out);
}

__global__ void preproc_fwd_kernel(const float4* __restrict__ y, float4* __restrict__ out, int B, int H, int W,
                                   int cy0, int cx0, int ch, int cw, int oh, int ow, float m0, float m1, float m2,
                                   float s0, float s1, float s2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * oh * ow) return;
    const int b = idx / (oh * ow);
    const int r = idx - b * oh * ow;
    const int oy = r / ow, ox = r - oy * ow;
    const int ys = win_start(oy, oh, ch), ye = win_end(oy, oh, ch);
    const int xs = win_start(ox, ow, cw), xe = win_end(ox, ow, cw);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int iy = ys; iy < ye; ++iy) {
        const float4* row = y + ((size_t)b * H + (cy0 + iy)) * W + cx0;
        for (int ix = xs; ix < xe; ++ix) {
            const float4 v = row[ix];
            a0 += v.x;
            a1 += v.y;
            a2 += v.z;
        }
    }
    const float cnt = (float)((ye - ys) * (xe - xs));
    out[idx] = make_float4((a0 / cnt - m0) / s0, (a1 / cnt - m1) / s1, (a2 / cnt - m2) / s2, 0.f);
}

// adjoint as a gather: every input pixel sums the output windows that cover it (deterministic, no atomics)
__global__ void preproc_bwd_kernel(const float4* __restrict__ g_out, float4* __restrict__ g_y, int B, int H, int W,
                                   int cy0, int cx0, int ch, int cw, int oh, int ow, float s0, float s1, float s2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H * W) return;
    const int b = idx / (H * W);
    const int r = idx - b * H * W;
    const int py = r / W - cy0, px = r % W - cx0;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (ch >= oh && cw >= ow) {
        // down-sampling (the attack's 240 -> 224): a pixel lies in at most TWO windows per axis.  Branch-free: the first two covering
        // outputs among four fixed candidates per axis (selects, no indexed arrays), then the four gradients loaded unconditionally
        // at clamped indices and added in row-major order (the order of the general path below: the same bits)
        const bool in = (unsigned)py < (unsigned)ch && (unsigned)px < (unsigned)cw;
        int oy2[2] = {-1, -1}, yl2[2] = {1, 1}, ox2[2] = {-1, -1}, xl2[2] = {1, 1};
        const int oy_lo = max((py * oh) / ch - 1, 0), oy_hi = min(((py + 1) * oh + ch - 1) / ch, oh - 1);
        const int ox_lo = max((px * ow) / cw - 1, 0), ox_hi = min(((px + 1) * ow + cw - 1) / cw, ow - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = oy_lo + k, ox = ox_lo + k;
            const int ys = win_start(min(oy, oh - 1), oh, ch), ye = win_end(min(oy, oh - 1), oh, ch);
            const int xs = win_start(min(ox, ow - 1), ow, cw), xe = win_end(min(ox, ow - 1), ow, cw);
            const bool cy = in && oy <= oy_hi && py >= ys && py < ye, cx = in && ox <= ox_hi && px >= xs && px < xe;
            const bool y0 = cy && oy2[0] < 0, y1 = cy && !y0 && oy2[1] < 0;
            oy2[0] = y0 ? oy : oy2[0]; yl2[0] = y0 ? ye - ys : yl2[0];
            oy2[1] = y1 ? oy : oy2[1]; yl2[1] = y1 ? ye - ys : yl2[1];
            const bool x0 = cx && ox2[0] < 0, x1 = cx && !x0 && ox2[1] < 0;
            ox2[0] = x0 ? ox : ox2[0]; xl2[0] = x0 ? xe - xs : xl2[0];
            ox2[1] = x1 ? ox : ox2[1]; xl2[1] = x1 ? xe - xs : xl2[1];
        }
        float4 gv[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) gv[2 * i + j] = g_out[((size_t)b * oh + max(oy2[i], 0)) * ow + max(ox2[j], 0)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = oy2[i] >= 0 && ox2[j] >= 0;
                const float inv = 1.f / (float)(yl2[i] * xl2[j]);
                a0 = ok ? fmaf(gv[2 * i + j].x, inv, a0) : a0;   // (fused, as the compiler contracts the general path's a += g * inv)
                a1 = ok ? fmaf(gv[2 * i + j].y, inv, a1) : a1;
                a2 = ok ? fmaf(gv[2 * i + j].z, inv, a2) : a2;
            }
    } else if ((unsigned)py < (unsigned)ch && (unsigned)px < (unsigned)cw) {
        // outputs whose window covers p: o in [floor(p*out/in), ceil((p+1)*out/in) - 1] (+-1 for the float window
        // bounds); the covering ones are collected per axis first, then combined (row-major order, as before)
        int oys[5], ylen[5], oxs[5], xlen[5], ny = 0, nx = 0;
        const int oy_lo = max((py * oh) / ch - 1, 0), oy_hi = min(((py + 1) * oh + ch - 1) / ch, oh - 1);
        for (int oy = oy_lo; oy <= oy_hi && ny < 5; ++oy) {
            const int ys = win_start(oy, oh, ch), ye = win_end(oy, oh, ch);
            if (py >= ys && py < ye) { oys[ny] = oy; ylen[ny] = ye - ys; ++ny; }
        }
        const int ox_lo = max((px * ow) / cw - 1, 0), ox_hi = min(((px + 1) * ow + cw - 1) / cw, ow - 1);
        for (int ox = ox_lo; ox <= ox_hi && nx < 5; ++ox) {
            const int xs = win_start(ox, ow, cw), xe = win_end(ox, ow, cw);
            if (px >= xs && px < xe) { oxs[nx] = ox; xlen[nx] = xe - xs; ++nx; }
        }
        for (int i = 0; i < ny; ++i)
            for (int j = 0; j < nx; ++j) {
                const float inv = 1.f / (float)(ylen[i] * xlen[j]);
                const float4 g = g_out[((size_t)b * oh + oys[i]) * ow + oxs[j]];
                a0 += g.x * inv;
                a1 += g.y * inv;
                a2 += g.z * inv;
            }
    }
    g_y[idx] = make_float4(a0 / s0, a1 / s1, a2 / s2, 0.f);
}

// The down-sampling case of the above (the attack's 240 -> 224), restructured: a thread owns ONE column of PRE_ROWS consecutive rows,
// finds its (at most two) covering output columns once, and the covering output rows are wave-uniform (scalar arithmetic); per
// pixel that leaves four gradient loads and four fused multiply-adds in the same row-major order (the same bits as the per-pixel
// kernel: 58 -> about 30 us at batch 64, 256 x 256, where the candidate search was 90 % of the instructions).
constexpr int PRE_ROWS = 8;
__device__ __forceinline__ void covering2(const int p, const int in, const int out, const bool inside, int (&o2)[2], int (&l2)[2]) {
    o2[0] = o2[1] = -1;
    l2[0] = l2[1] = 1;
    const int lo = max((p * out) / in - 1, 0), hi = min(((p + 1) * out + in - 1) / in, out - 1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int o = lo + k;
        const int s = win_start(min(o, out - 1), out, in), e = win_end(min(o, out - 1), out, in);
        const bool c = inside && o <= hi && p >= s && p < e;
        const bool c0 = c && o2[0] < 0, c1 = c && !c0 && o2[1] < 0;
        o2[0] = c0 ? o : o2[0]; l2[0] = c0 ? e - s : l2[0];
        o2[1] = c1 ? o : o2[1]; l2[1] = c1 ? e - s : l2[1];
    }
}
__global__ void preproc_bwd_down_kernel(const float4* __restrict__ g_out, float4* __restrict__ g_y, int B, int H, int W,
                                        int cy0, int cx0, int ch, int cw, int oh, int ow, float s0, float s1, float s2) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.z;
    if (x >= W) return;
    const int px = x - cx0;
    int ox2[2], xl2[2];
    covering2(px, cw, ow, (unsigned)px < (unsigned)cw, ox2, xl2);
    for (int r = 0; r < PRE_ROWS; ++r) {
        const int yy = blockIdx.y * PRE_ROWS + r;     // (uniform)
        if (yy >= H) break;
        const int py = yy - cy0;
        int oy2[2], yl2[2];
        covering2(py, ch, oh, (unsigned)py < (unsigned)ch, oy2, yl2);
        float4 gv[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) gv[2 * i + j] = g_out[((size_t)b * oh + max(oy2[i], 0)) * ow + max(ox2[j], 0)];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = oy2[i] >= 0 && ox2[j] >= 0;
                const float inv = 1.f / (float)(yl2[i] * xl2[j]);
                a0 = ok ? fmaf(gv[2 * i + j].x, inv, a0) : a0;
                a1 = ok ? fmaf(gv[2 * i + j].y, inv, a1) : a1;
                a2 = ok ? fmaf(gv[2 * i + j].z, inv, a2) : a2;
            }
        g_y[((size_t)b * H + yy) * W + x] = make_float4(a0 / s0, a1 / s1, a2 / s2, 0.f);
    }
}

// max_pool2d(kernel 3, stride 2, padding 1), NHWC, 4 channels per thread; first maximum in row-major window order
// wins (ATen CPU kernel: `val > maxval || isnan(val)`).
__global__ void maxpool_fwd_kernel(const float4* __restrict__ in, float4* __restrict__ out,
                                   uchar4* __restrict__ argmax, int B, int Hin, int Win, int C4, int Hout, int Wout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Hout * Wout * C4) return;
    const int c = idx % C4;
    int r = idx / C4;
    const int ox = r % Wout;
    r /= Wout;
    const int oy = r % Hout;
    const int b = r / Hout;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 am = make_uchar4(0, 0, 0, 0);
    // all nine window loads issued up front at clamped indices (a load under `continue` is waited for before the next is issued);
    // positions outside the image are skipped in the comparison, in the same row-major order as before
    float4 v[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = min(max(oy * 2 - 1 + ky, 0), Hin - 1), ix = min(max(ox * 2 - 1 + kx, 0), Win - 1);
            v[3 * ky + kx] = in[(((size_t)b * Hin + iy) * Win + ix) * C4 + c];
        }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
            const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
            const float4 t = v[3 * ky + kx];
            const unsigned char k = (unsigned char)(ky * 3 + kx);
            if (ok && (t.x > best.x || t.x != t.x)) { best.x = t.x; am.x = k; }
            if (ok && (t.y > best.y || t.y != t.y)) { best.y = t.y; am.y = k; }
            if (ok && (t.z > best.z || t.z != t.z)) { best.z = t.z; am.z = k; }
            if (ok && (t.w > best.w || t.w != t.w)) { best.w = t.w; am.w = k; }
        }
    // bit 7: the maximum is positive — the ReLU gate of the pooled tensor's producer, for the backward pass
    am.x |= best.x > 0.f ? 0x80 : 0;
    am.y |= best.y > 0.f ? 0x80 : 0;
    am.z |= best.z > 0.f ? 0x80 : 0;
    am.w |= best.w > 0.f ? 0x80 : 0;
    out[idx] = best;
    argmax[idx] = am;
}

// backward as a gather over the (at most 4) windows covering each input pixel; optional ReLU gate of the input.  Branch-free:
// the two candidate output rows / columns of a pixel are computed up front and all four (argmax byte, gradient) pairs are loaded
// unconditionally at clamped indices (a load inside a branch makes the compiler wait for it before the next one is issued).
__global__ void maxpool_bwd_kernel(const float4* __restrict__ g_out, const uchar4* __restrict__ argmax,
                                   const int relu_gate, float4* __restrict__ g_in, int B, int Hin,
                                   int Win, int C4, int Hout, int Wout) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Hin * Win * C4) return;
    const int c = idx % C4;
    int r = idx / C4;
    const int ix = r % Win;
    r /= Win;
    const int iy = r % Hin;
    const int b = r / Hin;
    const unsigned char need = relu_gate ? 0x80 : 0x00;  // relu_gate: only windows whose maximum is positive pass
    // window (oy, ky) covers row iy when iy = 2 oy - 1 + ky: even iy -> (iy / 2, 1); odd iy -> ((iy + 1) / 2, 0) and ((iy - 1) / 2, 2)
    int oyc[2], kyc[2], oxc[2], kxc[2];
    bool yok[2], xok[2];
    if (iy & 1) { oyc[0] = (iy + 1) >> 1; kyc[0] = 0; oyc[1] = (iy - 1) >> 1; kyc[1] = 2; yok[0] = oyc[0] < Hout; yok[1] = true; }
    else        { oyc[0] = iy >> 1; kyc[0] = 1; oyc[1] = 0; kyc[1] = 0; yok[0] = oyc[0] < Hout; yok[1] = false; }
    if (ix & 1) { oxc[0] = (ix + 1) >> 1; kxc[0] = 0; oxc[1] = (ix - 1) >> 1; kxc[1] = 2; xok[0] = oxc[0] < Wout; xok[1] = true; }
    else        { oxc[0] = ix >> 1; kxc[0] = 1; oxc[1] = 0; kxc[1] = 0; xok[0] = oxc[0] < Wout; xok[1] = false; }
    uchar4 am[4];
    float4 g[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = yok[i] ? oyc[i] : 0, ox = xok[j] ? oxc[j] : 0;
            const size_t o = (((size_t)b * Hout + oy) * Wout + ox) * C4 + c;
            am[2 * i + j] = argmax[o];
            g[2 * i + j] = g_out[o];
        }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // (summation order of the old loop: ky ascending, kx ascending -- candidate 1 of an odd coordinate has the LARGER k: add it last)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = yok[i] && xok[j];
            const unsigned char k = (unsigned char)(kyc[i] * 3 + kxc[j]);
            const uchar4 a = am[2 * i + j];
            const float4 v = g[2 * i + j];
            if (ok && (a.x & 0x7f) == k && (a.x & need) == need) acc.x += v.x;
            if (ok && (a.y & 0x7f) == k && (a.y & need) == need) acc.y += v.y;
            if (ok && (a.z & 0x7f) == k && (a.z & need) == need) acc.z += v.z;
            if (ok && (a.w & 0x7f) == k && (a.w & need) == need) acc.w += v.w;
        }
    g_in[idx] = acc;
}

// adaptive_avg_pool2d(1): one thread per (b, c); HW is small (49)
__global__ void avgpool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int HW, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    const float* p = in + (size_t)b * HW * C + c;
    float a = 0.f;
    // (eight loads in flight per thread: the 49 positions of a 7 x 7 map were 49 dependent round trips; same order of additions)
#pragma unroll 8
    for (int i = 0; i < HW; ++i) a += p[(size_t)i * C];
    out[idx] = a / (float)HW;
}

__global__ void avgpool_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ act,
                                   float* __restrict__ g_in, int B, int HW, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW * C) return;
    const int c = idx % C;
    const int b = idx / (HW * C);
    const float g = g_out[(size_t)b * C + c] / (float)HW;
    g_in[idx] = (act == nullptr || act[idx] > 0.f) ? g : 0.f;
}

// fp16-storage variants of the two global-average-pool kernels: activations / their gradients fp16, pooled features fp32
__global__ void avgpool_fwd_h_kernel(const _Float16* __restrict__ in, float* __restrict__ out, int B, int HW, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    const _Float16* p = in + (size_t)b * HW * C + c;
    float a = 0.f;
    // (eight loads in flight per thread: the 49 positions of a 7 x 7 map were 49 dependent round trips; same order of additions)
#pragma unroll 8
    for (int i = 0; i < HW; ++i) a += (float)p[(size_t)i * C];
    out[idx] = a / (float)HW;
}

__global__ void avgpool_bwd_h_kernel(const float* __restrict__ g_out, const _Float16* __restrict__ act,
                                     _Float16* __restrict__ g_in, int B, int HW, int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW * C) return;
    const int c = idx % C;
    const int b = idx / (HW * C);
    const float g = g_out[(size_t)b * C + c] / (float)HW;
    g_in[idx] = (_Float16)((act == nullptr || (float)act[idx] > 0.f) ? g : 0.f);
}

// The same adjoint with a thread per 2 x 2 block of input pixels (4 channels): the block's pixels draw on the same four windows
// (oy in {a, a + 1}, ox in {b, b + 1}), so four (arg-max byte, gradient) pairs serve four outputs instead of four pairs per output
// (the per-pixel kernel above loads all four candidates of every pixel, 2.25 of them live on average).  Same summation order.
__global__ void maxpool_bwd_quad_kernel(const float4* __restrict__ g_out, const uchar4* __restrict__ argmax, const int relu_gate,
                                        float4* __restrict__ g_in, int B, int Hin, int Win, int C4, int Hout, int Wout, int Hq, int Wq) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Hq * Wq * C4) return;
    const int c = idx % C4;
    int r = idx / C4;
    const int bq = r % Wq;
    r /= Wq;
    const int a = r % Hq;
    const int b = r / Hq;
    const unsigned char need = relu_gate ? 0x80 : 0x00;
    uchar4 am[2][2];
    float4 g[2][2];
    bool wok[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wok[p][q] = a + p < Hout && bq + q < Wout;
            const size_t o = (((size_t)b * Hout + (wok[p][q] ? a + p : 0)) * Wout + (wok[p][q] ? bq + q : 0)) * C4 + c;
            am[p][q] = argmax[o];
            g[p][q] = g_out[o];
        }
    auto take = [&](float4& acc, const int p, const int q, const unsigned char k) {
        const uchar4 m = am[p][q];
        const float4 v = g[p][q];
        if (wok[p][q] && (m.x & 0x7f) == k && (m.x & need) == need) acc.x += v.x;
        if (wok[p][q] && (m.y & 0x7f) == k && (m.y & need) == need) acc.y += v.y;
        if (wok[p][q] && (m.z & 0x7f) == k && (m.z & need) == need) acc.z += v.z;
        if (wok[p][q] && (m.w & 0x7f) == k && (m.w & need) == need) acc.w += v.w;
    };
    const int iy = 2 * a, ix = 2 * bq;
    const size_t base = (((size_t)b * Hin + iy) * Win + ix) * C4 + c;
    {   // (even row, even column): window (a, b), tap (1, 1)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        take(acc, 0, 0, 4);
        g_in[base] = acc;
    }
    if (ix + 1 < Win) {   // (even, odd): (a, b + 1) tap (1, 0), then (a, b) tap (1, 2)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        take(acc, 0, 1, 3);
        take(acc, 0, 0, 5);
        g_in[base + C4] = acc;
    }
    if (iy + 1 < Hin) {
        {   // (odd, even): (a + 1, b) tap (0, 1), then (a, b) tap (2, 1)
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            take(acc, 1, 0, 1);
            take(acc, 0, 0, 7);
            g_in[base + (size_t)Win * C4] = acc;
        }
        if (ix + 1 < Win) {   // (odd, odd): taps (0, 0), (0, 2), (2, 0), (2, 2)
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            take(acc, 1, 1, 0);
            take(acc, 1, 0, 2);
            take(acc, 0, 1, 6);
            take(acc, 0, 0, 8);
            g_in[base + (size_t)Win * C4 + C4] = acc;
        }
    }
}

inline int nblk(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace

extern "C" {

int spaa_preproc_fwd(const float* y, float* out, int B, int H, int W, int cy0, int cx0, int ch, int cw, int oh,
                     int ow, const float* mean3, const float* std3, spaa_stream_t stream) {
    if (!y || !out || !mean3 || !std3 || B < 1 || cy0 < 0 || cx0 < 0 || cy0 + ch > H || cx0 + cw > W || oh < 1 ||
        ow < 1 || ch < 1 || cw < 1)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(preproc_fwd_kernel, dim3(nblk((int64_t)B * oh * ow)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)y, (float4*)out, B, H, W, cy0, cx0, ch, cw, oh, ow, mean3[0], mean3[1],
                       mean3[2], std3[0], std3[1], std3[2]);
    return (int)hipGetLastError();
}

int spaa_preproc_bwd(const float* g_out, float* g_y, int B, int H, int W, int cy0, int cx0, int ch, int cw, int oh,
                     int ow, const float* std3, spaa_stream_t stream) {
    if (!g_out || !g_y || !std3 || B < 1 || cy0 < 0 || cx0 < 0 || cy0 + ch > H || cx0 + cw > W || oh < 1 ||
        ow < 1 || ch < 1 || cw < 1)
        return hipErrorInvalidValue;
    // the +-2 candidate search of the gather is exact while a window spans at most 3 inputs
    if (ch > 2 * oh || cw > 2 * ow) return hipErrorInvalidValue;
    // ... and an input pixel is covered by at most 5 outputs per axis (the gather's list length) up to 4x upscaling
    if (oh > 4 * ch || ow > 4 * cw) return hipErrorInvalidValue;
    if (ch >= oh && cw >= ow && B <= 65535) {   // down-sampling: a column of PRE_ROWS rows per thread
        hipLaunchKernelGGL(preproc_bwd_down_kernel, dim3((W + 255) / 256, (H + PRE_ROWS - 1) / PRE_ROWS, B), dim3(256), 0,
                           (hipStream_t)stream, (const float4*)g_out, (float4*)g_y, B, H, W, cy0, cx0, ch, cw, oh, ow, std3[0],
                           std3[1], std3[2]);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(preproc_bwd_kernel, dim3(nblk((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)g_out, (float4*)g_y, B, H, W, cy0, cx0, ch, cw, oh, ow, std3[0], std3[1],
                       std3[2]);
    return (int)hipGetLastError();
}

int spaa_maxpool3s2_fwd(const float* in, float* out, uint8_t* argmax, int B, int Hin, int Win, int C, int Hout,
                        int Wout, spaa_stream_t stream) {
    if (!in || !out || !argmax || (C & 3) || B < 1 || Hout != (Hin + 2 - 3) / 2 + 1 || Wout != (Win + 2 - 3) / 2 + 1)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nblk((int64_t)B * Hout * Wout * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)in, (float4*)out, (uchar4*)argmax, B, Hin, Win, C / 4,
                       Hout, Wout);
    return (int)hipGetLastError();
}

int spaa_maxpool3s2_bwd(const float* g_out, const uint8_t* argmax, int relu_gate, float* g_in, int B, int Hin,
                        int Win, int C, int Hout, int Wout, spaa_stream_t stream) {
    if (!g_out || !argmax || !g_in || (C & 3) || B < 1 || Hout != (Hin + 2 - 3) / 2 + 1 ||
        Wout != (Win + 2 - 3) / 2 + 1)
        return hipErrorInvalidValue;
    const int Hq = (Hin + 1) / 2, Wq = (Win + 1) / 2;
    hipLaunchKernelGGL(maxpool_bwd_quad_kernel, dim3(nblk((int64_t)B * Hq * Wq * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)g_out, (const uchar4*)argmax, relu_gate, (float4*)g_in, B, Hin, Win, C / 4, Hout, Wout, Hq, Wq);
    return (int)hipGetLastError();
}

int spaa_avgpool_fwd(const float* in, float* out, int B, int HW, int C, spaa_stream_t stream) {
    if (!in || !out || B < 1 || HW < 1 || C < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(nblk((int64_t)B * C)), dim3(256), 0, (hipStream_t)stream, in, out, B,
                       HW, C);
    return (int)hipGetLastError();
}

int spaa_avgpool_fwd_f16(const void* in, float* out, int B, int HW, int C, spaa_stream_t stream) {
    if (!in || !out || B < 1 || HW < 1 || C < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(avgpool_fwd_h_kernel, dim3(nblk((int64_t)B * C)), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)in, out, B, HW, C);
    return (int)hipGetLastError();
}

int spaa_avgpool_bwd_f16(const float* g_out, const void* act, void* g_in, int B, int HW, int C, spaa_stream_t stream) {
    if (!g_out || !g_in || B < 1 || HW < 1 || C < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(avgpool_bwd_h_kernel, dim3(nblk((int64_t)B * HW * C)), dim3(256), 0, (hipStream_t)stream, g_out,
                       (const _Float16*)act, (_Float16*)g_in, B, HW, C);
    return (int)hipGetLastError();
}

int spaa_avgpool_bwd(const float* g_out, const float* act, float* g_in, int B, int HW, int C, spaa_stream_t stream) {
    if (!g_out || !g_in || B < 1 || HW < 1 || C < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(nblk((int64_t)B * HW * C)), dim3(256), 0, (hipStream_t)stream, g_out,
                       act, g_in, B, HW, C);
    return (int)hipGetLastError();
}

}  // extern "C"
