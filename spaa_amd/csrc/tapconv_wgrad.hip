// tapconv_wgrad.hip — weight and bias gradients of the tap-list convolution (the PCNet training
// step, /root/reference/src/python/train_network.py:235-363 -> `train_loss_batch.backward()` :316).
//
// For the layer   out[b, oy0 + s_out*y, ox0 + s_out*x, n] = sum_t sum_c in[b, s_in*y + dy_t, s_in*x + dx_t, c] * W_t[n][c]
// the weight gradient is a GEMM whose K dimension is the PIXEL index:
//     dW_t[n][c] = sum_{b,y,x} gout[b, oy0 + s_out*y, ox0 + s_out*x, n] * in[b, s_in*y + dy_t, s_in*x + dx_t, c]
// (gout = gradient w.r.t. the layer's pre-activation, which the input-gradient passes already leave in the engine's
// gradient buffers).  Exact fp32 on the matrix cores: v_mfma_f32_32x32x2_f32 consumes two pixels per instruction; both
// operands are read straight from HBM/L2 as 128-byte channel rows (NHWC: 32 consecutive channels of one pixel), no LDS.
// A wave owns one (class, tap, 32 output channels, CT x 32 input channels) block over one chunk of the pixels; chunks are
// combined by a fixed-order second pass (deterministic, no float atomics).  The result has the layout of the packed
// forward weights ([class][Npad][Kpad]); the host maps it back to the parameter tensor (index plumbing).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int CT = 4;  // 32-channel input tiles per wave

// grid.x = (tile of the (class, tap, n-tile, c-group) list), grid.y = pixel chunk
__global__ __launch_bounds__(64) void wgrad_kernel(const spaa_tapconv_t p, const float* __restrict__ gout,
                                                   float* __restrict__ ws, const int nchunk, const int64_t wtotal) {
    const int lane = threadIdx.x;
    // decode blockIdx.x -> (class, tap, n tile, c group)
    const int n_tiles = (p.Cout + 31) >> 5;
    const int c_groups = (p.Cin + 32 * CT - 1) / (32 * CT);
    int rest = blockIdx.x;
    int ci = 0;
    for (; ci < p.nclass; ++ci) {
        const int cnt = p.cls[ci].ntaps * n_tiles * c_groups;
        if (rest < cnt) break;
        rest -= cnt;
    }
    if (ci >= p.nclass) return;
    const spaa_tapclass_t cl = p.cls[ci];
    const int t = rest / (n_tiles * c_groups);
    rest -= t * (n_tiles * c_groups);
    const int n0 = (rest / c_groups) * 32, c0 = (rest % c_groups) * (32 * CT);
    const int dy = p.taps[2 * (cl.tap_off + t)], dx = p.taps[2 * (cl.tap_off + t) + 1];

    const int HWm = p.Hm * p.Wm;
    const int64_t M = (int64_t)p.B * HWm;
    const int64_t per = ((M + nchunk - 1) / nchunk + 1) & ~(int64_t)1;  // even: a wave consumes pixel PAIRS
    const int64_t m_begin = (int64_t)blockIdx.y * per;
    const int64_t m_end = m_begin + per < M ? m_begin + per : M;

    f32x16 acc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int half = lane >> 5, l31 = lane & 31;
    const bool n_ok = n0 + l31 < p.Cout;
    // this lane's pixel: m_begin + half, advancing by 2
    int64_t m = m_begin + half;
    int b = (int)(m / HWm);
    int rr = (int)(m - (int64_t)b * HWm);
    int y = rr / p.Wm, x = rr - y * p.Wm;
    // operands of this lane's pixel m (zeros past the chunk, outside the image, for absent channels) ...
    auto fetch = [&](const int64_t mm, float& a, float (&bv)[CT]) {
        a = 0.f;
#pragma unroll
        for (int j = 0; j < CT; ++j) bv[j] = 0.f;
        if (mm < m_end) {
            const int oy = cl.oy0 + y * p.s_out, ox = cl.ox0 + x * p.s_out;
            const int iy = y * p.s_in + dy, ix = x * p.s_in + dx;
            if (oy < p.Hout && ox < p.Wout && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) {
                if (n_ok) a = gout[(((size_t)b * p.Hout + oy) * p.Wout + ox) * p.out_cstride + p.out_coff + n0 + l31];
                const float* ip = p.in + (((size_t)b * p.Hin + iy) * p.Win + ix) * p.in_cstride + p.in_coff + c0 + l31;
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    if (c0 + 32 * j + l31 < p.Cin) bv[j] = ip[32 * j];
            }
        }
    };
    // ... and the step to the lane's next pixel (two further in row-major order)
    auto advance = [&]() {
        x += 2;
        while (x >= p.Wm) {
            x -= p.Wm;
            y += 1;
            if (y >= p.Hm) {
                y = 0;
                b += 1;
            }
        }
    };
    // the operands of pixel pair i + 1 are requested BEFORE the products of pair i are issued (a dependent round trip to L2 / HBM per
    // pair otherwise: 13.8 ms of the 50 ms training step, profiles/r05_train_step_rocprofv3_kernel_stats.csv); same products in the
    // same order
    float a_cur, b_cur[CT];
    fetch(m, a_cur, b_cur);
    for (; m < m_end + half; m += 2) {   // (both halves run the same number of iterations: the MFMA needs the whole wave)
        float a_nxt, b_nxt[CT];
        advance();
        fetch(m + 2, a_nxt, b_nxt);
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[j], acc[j], 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int j = 0; j < CT; ++j) b_cur[j] = b_nxt[j];
    }
    // D layout: column (lane & 31) = input channel, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) = output channel
    float* dst = ws + (size_t)blockIdx.y * wtotal + cl.w_off;
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int c = c0 + 32 * j + l31;
        if (c >= p.Cin) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (n < p.Cout) dst[(size_t)n * cl.Kpad + t * p.Cin + c] = acc[j][r];
        }
    }
}

// out[i] = sum over chunks (fixed order) of ws[chunk][i]
__global__ void chunk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, const int nchunk, const int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) s += ws[(size_t)c * n + i];
    out[i] = s;
}

// bias gradient partials of one pixel chunk per block, 16-byte loads: thread = (pixel row r of R = 256 / L, channel quad q of L); a
// thread sums its pixels r, r + R, ... in order, the R rows are added in order through LDS (fixed order: deterministic).  The
// per-channel form below kept 3 ... 256 threads per block busy with one dependent 4-byte load each (conv6, 3 channels: 560 us for
// a 201 MB tensor; profiles/r05_train_step_rocprofv3_kernel_stats.csv: 5.05 ms per training step over the 21 layers)
__global__ __launch_bounds__(256) void bias_partial4_kernel(const float* __restrict__ gout, float* __restrict__ ws, const int64_t npix,
                                                            const int cstride, const int coff, const int C, const int nchunk) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    __shared__ f4v red[256];
    const int C4 = (C + 3) >> 2;
    const int L = C4 < 64 ? C4 : 64, R = 256 / L;
    const int r = threadIdx.x / L, q0 = threadIdx.x - r * L;
    const int64_t per = (npix + nchunk - 1) / nchunk;
    const int64_t m0 = (int64_t)blockIdx.x * per, m1 = m0 + per < npix ? m0 + per : npix;
    for (int qb = 0; qb < C4; qb += L) {
        const int q = qb + q0;
        f4v s = {0.f, 0.f, 0.f, 0.f};
        if (r < R && q < C4)
            for (int64_t m = m0 + r; m < m1; m += R) s += *reinterpret_cast<const f4v*>(gout + (size_t)m * cstride + coff + 4 * q);
        red[threadIdx.x] = s;
        __syncthreads();
        if (r == 0 && q < C4) {
            f4v t = red[q0];
            for (int k = 1; k < R; ++k) t += red[k * L + q0];
            float* dst = ws + (size_t)blockIdx.x * C + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * q + e < C) dst[e] = t[e];
        }
        __syncthreads();
    }
}

// (the per-channel form: layouts whose channel stride / offset is not a multiple of 4)
__global__ void bias_partial_kernel(const float* __restrict__ gout, float* __restrict__ ws, const int64_t npix, const int cstride,
                                    const int coff, const int C, const int nchunk) {
    const int64_t per = (npix + nchunk - 1) / nchunk;
    const int64_t m0 = (int64_t)blockIdx.x * per, m1 = m0 + per < npix ? m0 + per : npix;
    for (int n = threadIdx.x; n < C; n += blockDim.x) {
        float s = 0.f;
        for (int64_t m = m0; m < m1; ++m) s += gout[(size_t)m * cstride + coff + n];
        ws[(size_t)blockIdx.x * C + n] = s;
    }
}

}  // namespace

extern "C" {

int spaa_tapconv_wgrad(const spaa_tapconv_t* desc, const float* gout, float* dw_packed, float* dbias, float* workspace,
                       int nchunk, spaa_stream_t stream_) {
    if (!desc || !gout || !dw_packed || !workspace || nchunk < 1 || nchunk > 65535) return hipErrorInvalidValue;
    const spaa_tapconv_t& d = *desc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (d.in == nullptr || d.taps == nullptr || d.Cin <= 0 || d.Cout <= 0 || d.B <= 0 || d.Hm <= 0 || d.Wm <= 0 ||
        d.nclass < 1 || d.nclass > SPAA_MAX_CLASSES || d.s_in < 1 || d.s_out < 1 || d.io_dtype != 0 || d.nfold > 1)
        return hipErrorInvalidValue;
    if (d.in_coff + d.Cin > d.in_cstride || d.out_coff + d.Cout > d.out_cstride) return hipErrorInvalidValue;
    const int npad = (d.Cout + 127) & ~127;
    int64_t wtotal = 0;
    int blocks = 0;
    const int n_tiles = (d.Cout + 31) >> 5, c_groups = (d.Cin + 32 * CT - 1) / (32 * CT);
    for (int c = 0; c < d.nclass; ++c) {
        if (d.cls[c].K != d.cls[c].ntaps * d.Cin || d.cls[c].Kpad < d.cls[c].K || d.cls[c].w_off != wtotal) return hipErrorInvalidValue;
        wtotal += (int64_t)npad * d.cls[c].Kpad;
        blocks += d.cls[c].ntaps * n_tiles * c_groups;
    }
    if (blocks < 1 || wtotal >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)nchunk * wtotal * sizeof(float), stream);  // pad rows / columns stay 0
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(wgrad_kernel, dim3(blocks, nchunk), dim3(64), 0, stream, d, gout, workspace, nchunk, wtotal);
    hipLaunchKernelGGL(chunk_reduce_kernel, dim3((unsigned)((wtotal + 255) / 256)), dim3(256), 0, stream, workspace, dw_packed,
                       nchunk, wtotal);
    if (dbias != nullptr) {
        const int64_t npix = (int64_t)d.B * d.Hout * d.Wout;
        // (a 4-channel load past Cout stays inside the pixel's row when the window ends on a multiple of 4 or the row does)
        if (!(d.out_cstride & 3) && !(d.out_coff & 3) && ((d.out_coff + ((d.Cout + 3) & ~3)) <= d.out_cstride))
            hipLaunchKernelGGL(bias_partial4_kernel, dim3(nchunk), dim3(256), 0, stream, gout, workspace, npix, d.out_cstride, d.out_coff,
                               d.Cout, nchunk);
        else
            hipLaunchKernelGGL(bias_partial_kernel, dim3(nchunk), dim3(d.Cout >= 256 ? 256 : ((d.Cout + 63) & ~63)), 0, stream, gout,
                               workspace, npix, d.out_cstride, d.out_coff, d.Cout, nchunk);
        hipLaunchKernelGGL(chunk_reduce_kernel, dim3((d.Cout + 255) / 256), dim3(256), 0, stream, workspace, dbias, nchunk,
                           (int64_t)d.Cout);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
