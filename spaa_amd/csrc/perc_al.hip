// perc_al.hip — device-side pieces of PerC_AL.adversary_projector (the PerC-AL + CompenNet++ baseline attacker).
//
// Replaces (perc_al/__init__.py, relative to /root/reference/src/python): CrossEntropyLoss(reduction='sum') backward
// :186-188, the masked normalised steps :193-195 / :204-209, the box clamp + 8-bit quantisation :211-212, the
// perturbation size :215-216, the adversarial tests on the quantised image :218-238 and the best bookkeeping :240-245.
// The loop state stays on the GPU (the reference pulls p / idx to the host twice per iteration).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// x = a + b on NHWC4 (inputs + delta)
__global__ void add_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 u = a[i], v = b[i];
    x[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, 0.f);
}

// d (mult * CE_sum) / d logits = mult * (softmax(l) - onehot(label)); one workgroup per sample
__global__ __launch_bounds__(256) void ce_grad_kernel(const float* __restrict__ logits, int ncls,
                                                      const int32_t* __restrict__ label, float mult,
                                                      float* __restrict__ g) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* l = logits + (size_t)b * ncls;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < ncls; i += 256) mx = fmaxf(mx, l[i]);
    mx = block_max(mx, red);
    float se = 0.f;
    for (int i = threadIdx.x; i < ncls; i += 256) se += expf(l[i] - mx);
    se = block_sum(se, red);
    const int t = label[b];
    for (int i = threadIdx.x; i < ncls; i += 256) {
        const float p = expf(l[i] - mx) / se;
        g[(size_t)b * ncls + i] = mult * (p - (i == t ? 1.f : 0.f));
    }
}

// x_b += step * g_b / ||g_b||  for samples whose state[b][col] == want; ||g_b||^2 from block partials. grid (nblk,B)
__global__ __launch_bounds__(256) void masked_step_kernel(float4* __restrict__ x, const float4* __restrict__ g,
                                                          const float* __restrict__ partial, int nblk,
                                                          const int32_t* __restrict__ state, int col, int want,
                                                          float step, int HW) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    if ((state[4 * b + col] != 0) != (want != 0)) return;  // uniform per workgroup
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += partial[(size_t)b * nblk + i];
    a = block_sum(a, red);
    const float nrm = sqrtf(a);
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix < HW) {
        const size_t idx = (size_t)b * HW + pix;
        float4 xv = x[idx];
        const float4 gv = g[idx];
        xv.x += step * (gv.x / nrm);
        xv.y += step * (gv.y / nrm);
        xv.z += step * (gv.z / nrm);
        x[idx] = xv;
    }
}

// colour distance of PerC-AL: g_px *= dE_px / ||dE_b||_2 (d ||d_map||_2 / d x), ||dE_b||^2 from block partials
// (third partial of the stealth-loss kernel); also writes color_dis_b = ||dE_b||_2.   grid (nblk, B)
__global__ __launch_bounds__(256) void scale_by_map_kernel(float4* __restrict__ g, const float* __restrict__ de_map,
                                                           const float* __restrict__ partial3, int nblk,
                                                           float* __restrict__ color_dis, int HW) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += partial3[3 * ((size_t)b * nblk + i) + 2];
    a = block_sum(a, red);
    const float nrm = sqrtf(a);
    if (blockIdx.x == 0 && threadIdx.x == 0) color_dis[b] = nrm;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix < HW) {
        const size_t idx = (size_t)b * HW + pix;
        const float k = (nrm != 0.f) ? de_map[idx] / nrm : 0.f;  // ATen norm_backward: 0 where the norm is 0
        float4 v = g[idx];
        v.x *= k;
        v.y *= k;
        v.z *= k;
        g[idx] = v;
    }
}

// delta = clamp(inputs + delta, 0, 1) - inputs ; x_round = round((inputs + delta) * 255) / 255 ; block partials of
// sum_px ||delta_px||_2.    grid (nblk, B)
__global__ __launch_bounds__(256) void clamp_quant_kernel(const float4* __restrict__ inputs, float4* __restrict__ delta,
                                                          float4* __restrict__ x_round, float* __restrict__ partial,
                                                          int HW) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    float l2 = 0.f;
    if (pix < HW) {
        const size_t idx = (size_t)b * HW + pix;
        const float4 in = inputs[idx];
        float4 d = delta[idx];
        d.x = fminf(fmaxf(in.x + d.x, 0.f), 1.f) - in.x;
        d.y = fminf(fmaxf(in.y + d.y, 0.f), 1.f) - in.y;
        d.z = fminf(fmaxf(in.z + d.z, 0.f), 1.f) - in.z;
        d.w = 0.f;
        delta[idx] = d;
        x_round[idx] = make_float4(rintf((in.x + d.x) * 255.f) / 255.f, rintf((in.y + d.y) * 255.f) / 255.f,
                                   rintf((in.z + d.z) * 255.f) / 255.f, 0.f);
        l2 = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
    }
    l2 = block_sum(l2, red);
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = l2;
}

// Decisions on the quantised image (:215-245). mode 0: targeted (argmax == label, p1 > p_thresh); 1: untargeted
// (argmax != label); 2: untargeted with Carlini margin (real - best other <= -confidence).
// state [B][4]: 0 isadv, 1 best_adv, 2 best, 3 top1.  stats [B][8]: 0 p1, 1 caml2, 2 margin, 3 color_dis,
// 5 bound_best (in/out).    one workgroup per sample
__global__ __launch_bounds__(256) void perc_decide_kernel(const float* __restrict__ logits, int ncls,
                                                          const int32_t* __restrict__ label, int mode,
                                                          float confidence, const float* __restrict__ partial, int nblk,
                                                          int HW, const float* __restrict__ color_dis, float d_thr,
                                                          float p_thresh, int32_t* __restrict__ state,
                                                          float* __restrict__ stats) {
    __shared__ float red[4];
    __shared__ float s_max[4];
    __shared__ int s_arg[4];
    const int b = blockIdx.x;
    const float* lg = logits + (size_t)b * ncls;
    const int t = label[b];
    float mx = -INFINITY, other = -INFINITY;
    int am = 0x7fffffff;
    for (int i = threadIdx.x; i < ncls; i += 256) {
        const float v = lg[i];
        if (v > mx) {
            mx = v;
            am = i;
        }
        if (i != t) other = fmaxf(other, v);
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
    for (int w = 1; w < 4; ++w)
        if (s_max[w] > mx || (s_max[w] == mx && s_arg[w] < am)) {
            mx = s_max[w];
            am = s_arg[w];
        }
    other = block_max(other, red);
    float se = 0.f;
    for (int i = threadIdx.x; i < ncls; i += 256) se += expf(lg[i] - mx);
    se = block_sum(se, red);
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += partial[(size_t)b * nblk + i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) {
        const float p1 = 1.f / se;
        const float caml2 = a / (float)HW;
        const bool high_pert = caml2 * 255.f > d_thr;
        const float margin = lg[t] - other;
        bool isadv, best_adv;
        if (mode == 0) {
            isadv = (am == t);
            best_adv = isadv && (p1 > p_thresh) && high_pert;
        } else if (mode == 1) {
            isadv = (am != t);
            best_adv = isadv && high_pert;
        } else {
            isadv = margin <= -confidence;
            best_adv = isadv && high_pert;
        }
        float* st = stats + 8 * (size_t)b;
        const float cd = color_dis[b];
        const bool best = best_adv && (cd < st[5]);
        if (best) st[5] = cd;
        st[0] = p1;
        st[1] = caml2;
        st[2] = margin;
        st[3] = cd;
        int32_t* s = state + 4 * (size_t)b;
        s[0] = isadv;
        s[1] = best_adv;
        s[2] = best;
        s[3] = am;
    }
}

// dst_b = src_b where state[b][0] (isadv) — mask_best is a subset of it (:244-245)
__global__ void track_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                             const int32_t* __restrict__ state, int B, int HW) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * HW) return;
    if (state[4 * (idx / HW)] != 0) dst[idx] = src[idx];
}

}  // namespace

extern "C" {

int spaa_add_nhwc4(const float* a, const float* b, float* x, int npix, spaa_stream_t stream) {
    if (!a || !b || !x || npix < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_kernel, dim3((npix + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float4*)a,
                       (const float4*)b, (float4*)x, npix);
    return (int)hipGetLastError();
}

int spaa_ce_grad(const float* logits, int ncls, const int32_t* label, float mult, float* g_logits, int B,
                 spaa_stream_t stream) {
    if (!logits || !label || !g_logits || B < 1 || ncls < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ce_grad_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ncls, label, mult, g_logits);
    return (int)hipGetLastError();
}

int spaa_masked_step(float* x, const float* g, const float* partial, const int32_t* state, int col, int want,
                     float step, int B, int HW, spaa_stream_t stream) {
    if (!x || !g || !partial || !state || col < 0 || col > 3 || B < 1 || HW < 1) return hipErrorInvalidValue;
    const int nblk = (HW + 255) / 256;
    hipLaunchKernelGGL(masked_step_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, (float4*)x,
                       (const float4*)g, partial, nblk, state, col, want, step, HW);
    return (int)hipGetLastError();
}

int spaa_scale_by_map(float* g, const float* de_map, const float* partial3, float* color_dis, int B, int HW,
                      spaa_stream_t stream) {
    if (!g || !de_map || !partial3 || !color_dis || B < 1 || HW < 1) return hipErrorInvalidValue;
    const int nblk = (HW + 255) / 256;
    hipLaunchKernelGGL(scale_by_map_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, (float4*)g, de_map,
                       partial3, nblk, color_dis, HW);
    return (int)hipGetLastError();
}

int spaa_perc_clamp_quant(const float* inputs, float* delta, float* x_round, float* partial, int B, int HW,
                          spaa_stream_t stream) {
    if (!inputs || !delta || !x_round || !partial || B < 1 || HW < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(clamp_quant_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)inputs, (float4*)delta, (float4*)x_round, partial, HW);
    return (int)hipGetLastError();
}

int spaa_perc_decide(const float* logits, int ncls, const int32_t* label, int mode, float confidence,
                     const float* partial, int nblk, int HW, const float* color_dis, float d_thr, float p_thresh,
                     int32_t* state, float* stats, int B, spaa_stream_t stream) {
    if (!logits || !label || !partial || !color_dis || !state || !stats || B < 1 || ncls < 1 || nblk < 1 || mode < 0 ||
        mode > 2)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(perc_decide_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, ncls, label, mode,
                       confidence, partial, nblk, HW, color_dis, d_thr, p_thresh, state, stats);
    return (int)hipGetLastError();
}

int spaa_track_where(const float* src, float* dst, const int32_t* state, int B, int HW, spaa_stream_t stream) {
    if (!src || !dst || !state || B < 1 || HW < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(track_kernel, dim3((int)(((int64_t)B * HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (float4*)dst, state, B, HW);
    return (int)hipGetLastError();
}

}  // extern "C"
