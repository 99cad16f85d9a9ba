// attack_ops.hip — SPAA Algorithm 1's control logic and projected-gradient step, entirely on device.
//
// Replaces (projector_based_attack.py): adversarial loss :269-272, loss reductions :275-287, masks :290-299
// (the reference round-trips to the host three times per iteration: classifier.py:64, :291, :318), gradient
// normalisation and masked updates :302-315, best-so-far bookkeeping :318-328.
//
// Per sample b the reference consumes EITHER g_adv (if not best_adv) OR g_col (if best_adv), never both
// (:307,315), and samples do not interact (no BatchNorm in PCNet, classifier in eval mode).  So one backward pass
// with a per-sample-selected cotangent reproduces both of the reference's backward passes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// fixed-order block reduction (256 threads); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// state: [B][4] = succ, best_adv, best, top1      stats: [B][8] = p1, caml2, camdE, col_loss, prjl2, col_loss_best,
// target logit, reserved
__global__ __launch_bounds__(256) void decide_kernel(const float* __restrict__ logits, int ncls,
                                                     const int32_t* __restrict__ target, int targeted,
                                                     const float* __restrict__ partial, int nblk, int HW,
                                                     const float* __restrict__ prjl2, float prjl2_w, float caml2_w,
                                                     float camdE_w, float d_thr, float p_thresh, float adv_scale,
                                                     int32_t* __restrict__ state, float* __restrict__ stats,
                                                     float* __restrict__ g_logits) {
    __shared__ float red[4];
    __shared__ float s_max[4];
    __shared__ int s_arg[4];
    const int b = blockIdx.x;
    const float* lg = logits + (size_t)b * ncls;
    // argmax (first maximum) and max
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int i = threadIdx.x; i < ncls; i += 256) {
        const float v = lg[i];
        if (v > mx) {
            mx = v;
            am = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(mx, off, 64);
        const int oa = __shfl_down(am, off, 64);
        if (ov > mx || (ov == mx && oa < am)) {
            mx = ov;
            am = oa;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_max[wave] = mx;
        s_arg[wave] = am;
    }
    __syncthreads();
    mx = s_max[0];
    am = s_arg[0];
    for (int w = 1; w < 4; ++w) {
        if (s_max[w] > mx || (s_max[w] == mx && s_arg[w] < am)) {
            mx = s_max[w];
            am = s_arg[w];
        }
    }
    // softmax top-1 probability = 1 / sum exp(l - max)   (classifier.py:64)
    float se = 0.f;
    for (int i = threadIdx.x; i < ncls; i += 256) se += expf(lg[i] - mx);
    se = block_sum(se, red);
    const float p1 = 1.f / se;
    // loss sums (fixed order)
    float a = 0.f, d = 0.f;
    const float* pp = partial + 3 * (size_t)b * nblk;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        a += pp[3 * i];
        d += pp[3 * i + 1];
    }
    a = block_sum(a, red);
    d = block_sum(d, red);
    const int tgt = target[b];
    // d adv_loss / d logits: adv_loss = -/+ mean_b logit[b, target_b]   (:269-272)
    for (int i = threadIdx.x; i < ncls; i += 256)
        g_logits[(size_t)b * ncls + i] = (i == tgt) ? (targeted ? -adv_scale : adv_scale) : 0.f;
    if (threadIdx.x == 0) {
        const float caml2 = a / (float)HW;
        const float camdE = d / (float)HW;
        const float pl2 = (prjl2 != nullptr) ? prjl2[b] : 0.f;
        float col = prjl2_w * pl2;
        col += caml2_w * caml2;
        col += camdE_w * camdE;
        const bool high_conf = p1 > p_thresh;
        const bool high_pert = caml2 * 255.f > d_thr;
        const bool succ = targeted ? (am == tgt) : (am != tgt);
        const bool best_adv = targeted ? (succ && high_conf && high_pert) : (succ && high_pert);
        float* st = stats + 8 * (size_t)b;
        const bool best = best_adv && (col < st[5]);
        if (best) st[5] = col;
        st[0] = p1;
        st[1] = caml2;
        st[2] = camdE;
        st[3] = col;
        st[4] = pl2;
        st[6] = lg[tgt];
        int32_t* s = state + 4 * (size_t)b;
        s[0] = succ;
        s[1] = best_adv;
        s[2] = best;
        s[3] = am;
    }
}

// cotangent at the PCNet output: per-sample choice between the classifier path and the stealth-loss path, then the
// backward of clamp(relu(.), max=1) (models.py:301): pass where 0 < pre <= 1.
__global__ void select_grad_kernel(const float4* __restrict__ g_adv, const float4* __restrict__ g_col,
                                   const int32_t* __restrict__ state, const float4* __restrict__ ypre,
                                   float4* __restrict__ g, int B, int npix) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * npix) return;
    const int b = idx / npix;
    const bool best_adv = state[4 * b + 1] != 0;
    float4 v = best_adv ? g_col[idx] : g_adv[idx];
    if (ypre != nullptr) {
        const float4 y = ypre[idx];
        v.x = (y.x > 0.f && y.x <= 1.f) ? v.x : 0.f;
        v.y = (y.y > 0.f && y.y <= 1.f) ? v.y : 0.f;
        v.z = (y.z > 0.f && y.z <= 1.f) ? v.z : 0.f;
    }
    v.w = 0.f;
    g[idx] = v;
}

// prjl2_b = mean_px || gray - x ||_2   (:275); one workgroup per sample
__global__ __launch_bounds__(256) void prjl2_kernel(const float4* __restrict__ x, float gray,
                                                    float* __restrict__ prjl2, int HW) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    float a = 0.f;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float4 v = x[(size_t)b * HW + i];
        const float d0 = gray - v.x, d1 = gray - v.y, d2 = gray - v.z;
        a += sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    }
    a = block_sum(a, red);
    if (threadIdx.x == 0) prjl2[b] = a / (float)HW;
}

// Adds the prjl2 gradient to g for samples taking the colour step, and writes block partials of ||g_b||^2.
// grid (nblk, B)
__global__ __launch_bounds__(256) void grad_sumsq_kernel(float4* __restrict__ g, const float4* __restrict__ x,
                                                         float gray, float prjl2_scale,
                                                         const int32_t* __restrict__ state,
                                                         float* __restrict__ partial, int HW) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    float ss = 0.f;
    if (pix < HW) {
        const size_t idx = (size_t)b * HW + pix;
        float4 v = g[idx];
        if (prjl2_scale != 0.f && state[4 * b + 1] != 0) {
            const float4 xv = x[idx];
            const float d0 = gray - xv.x, d1 = gray - xv.y, d2 = gray - xv.z;
            const float n = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            if (n != 0.f) {  // d||gray - x|| / dx = -(gray - x)/n ; zero where the norm is zero (ATen norm_backward)
                const float k = -prjl2_scale / n;
                v.x += k * d0;
                v.y += k * d1;
                v.z += k * d2;
                g[idx] = v;
            }
        }
        ss = v.x * v.x + v.y * v.y + v.z * v.z;
    }
    ss = block_sum(ss, red);
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = ss;
}

// x_b -= lr_b * g_b / ||g_b||  with lr = col_lr if best_adv else adv_lr  (:307,315); then x_best_b = x_b where succ
// (read AFTER the update, Q4).   grid (nblk, B)
__global__ __launch_bounds__(256) void step_kernel(float4* __restrict__ x, const float4* __restrict__ g,
                                                   const float* __restrict__ partial, int nblk,
                                                   const int32_t* __restrict__ state, float adv_lr, float col_lr,
                                                   float4* __restrict__ x_best, int HW, uint8_t* __restrict__ clamp_bits) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += partial[(size_t)b * nblk + i];
    a = block_sum(a, red);
    const float nrm = sqrtf(a);
    const float lr = (state[4 * b + 1] != 0) ? col_lr : adv_lr;
    const bool succ = state[4 * b] != 0;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    unsigned int bits = 0;
    if (pix < HW) {
        const size_t idx = (size_t)b * HW + pix;
        float4 xv = x[idx];
        const float4 gv = g[idx];
        xv.x -= lr * (gv.x / nrm);
        xv.y -= lr * (gv.y / nrm);
        xv.z -= lr * (gv.z / nrm);
        x[idx] = xv;
        if (succ) x_best[idx] = xv;
        bits = (xv.x >= 0.f && xv.x <= 1.f ? 1u : 0u) | (xv.y >= 0.f && xv.y <= 1.f ? 2u : 0u) | (xv.z >= 0.f && xv.z <= 1.f ? 4u : 0u);
    }
    // the clamp gate of the NEXT iteration's backward pass (x.clamp(0, 1), projector_based_attack.py:265 -> models.py:337): one byte per
    // pixel, bit c = channel c inside [0, 1] -- the grid_sample adjoint then reads 1 byte per pixel instead of x's 16.  Four lanes' bytes
    // leave as one 4-byte store where the image's rows allow it (HW % 4 == 0: every group of four pixels lies in one image, aligned).
    if (clamp_bits != nullptr) {
        if ((HW & 3) == 0) {
            unsigned int w = bits | (__shfl_down(bits, 1, 64) << 8);
            w |= __shfl_down(w, 2, 64) << 16;
            if ((threadIdx.x & 3) == 0 && pix < HW) *reinterpret_cast<unsigned int*>(clamp_bits + (size_t)b * HW + pix) = w;
        } else if (pix < HW) {
            clamp_bits[(size_t)b * HW + pix] = (uint8_t)bits;
        }
    }
}

// cam_infer_best_b = cam_infer_b where succ (:324,328)
__global__ void track_cam_kernel(const float4* __restrict__ cam, float4* __restrict__ cam_best,
                                 const int32_t* __restrict__ state, int B, int HW) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    const int b = idx / HW;
    if (state[4 * b] != 0) cam_best[idx] = cam[idx];
}

}  // namespace

extern "C" {

int spaa_decide(const float* logits, int ncls, const int32_t* target, int targeted, const float* partial, int nblk,
                int HW, const float* prjl2, float prjl2_w, float caml2_w, float camdE_w, float d_thr, float p_thresh,
                float adv_scale, int32_t* state, float* stats, float* g_logits, int B, spaa_stream_t stream) {
    if (!logits || !target || !partial || !state || !stats || !g_logits || B < 1 || ncls < 1 || nblk < 1 || HW < 1)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(decide_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ncls, target, targeted,
                       partial, nblk, HW, prjl2, prjl2_w, caml2_w, camdE_w, d_thr, p_thresh, adv_scale, state, stats,
                       g_logits);
    return (int)hipGetLastError();
}

int spaa_select_grad(const float* g_adv, const float* g_col, const int32_t* state, const float* ypre, float* g, int B,
                     int npix, spaa_stream_t stream) {
    if (!g_adv || !g_col || !state || !g || B < 1 || npix < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(select_grad_kernel, dim3((int)(((int64_t)B * npix + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)g_adv, (const float4*)g_col, state, (const float4*)ypre,
                       (float4*)g, B, npix);
    return (int)hipGetLastError();
}

int spaa_prjl2_fwd(const float* x, float gray, float* prjl2, int B, int HW, spaa_stream_t stream) {
    if (!x || !prjl2 || B < 1 || HW < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(prjl2_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const float4*)x, gray, prjl2, HW);
    return (int)hipGetLastError();
}

int spaa_grad_sumsq(float* g, const float* x, float gray, float prjl2_scale, const int32_t* state, float* partial,
                    int B, int HW, spaa_stream_t stream) {
    if (!g || !x || !state || !partial || B < 1 || HW < 1) return hipErrorInvalidValue;
    dim3 grid((HW + 255) / 256, B);
    hipLaunchKernelGGL(grad_sumsq_kernel, grid, dim3(256), 0, (hipStream_t)stream, (float4*)g, (const float4*)x, gray,
                       prjl2_scale, state, partial, HW);
    return (int)hipGetLastError();
}

int spaa_step_and_track(float* x, const float* g, const float* partial, const int32_t* state, float adv_lr,
                        float col_lr, float* x_best, const float* cam, float* cam_best, int B, int HWp, int HWc,
                        spaa_stream_t stream) {
    return spaa_step_and_track_n(x, g, partial, (HWp + 255) / 256, state, adv_lr, col_lr, x_best, cam, cam_best, B, HWp, HWc, nullptr, stream);
}

int spaa_step_and_track_n(float* x, const float* g, const float* partial, int npartial, const int32_t* state, float adv_lr,
                          float col_lr, float* x_best, const float* cam, float* cam_best, int B, int HWp, int HWc,
                          uint8_t* clamp_bits, spaa_stream_t stream) {
    if (!x || !g || !partial || !state || !x_best || !cam || !cam_best || B < 1 || HWp < 1 || HWc < 1 || npartial < 1)
        return hipErrorInvalidValue;
    const int nblk = (HWp + 255) / 256;
    dim3 grid(nblk, B);
    hipLaunchKernelGGL(step_kernel, grid, dim3(256), 0, (hipStream_t)stream, (float4*)x, (const float4*)g, partial,
                       npartial, state, adv_lr, col_lr, (float4*)x_best, HWp, clamp_bits);
    hipLaunchKernelGGL(track_cam_kernel, dim3((int)(((int64_t)B * HWc + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)cam, (float4*)cam_best, state, B, HWc);
    return (int)hipGetLastError();
}

}  // extern "C"
