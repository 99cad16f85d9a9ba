// train_ops.hip — the parts of the PCNet training step (/root/reference/src/python/
// train_network.py:235-363) that are not convolutions: gradients of the loss w.r.t. the WarpingNet's sampling grid and its
// affine / thin-plate-spline parameters (models.py:163-185, pytorch_tps.py:54-106 under autograd), and the Adam update
// (torch.optim.Adam as the reference configures it, train_network.py:252-254).  Reductions are fixed-order (block partials,
// then one pass over the blocks): deterministic, no float atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "warp_common.hpp"

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

// d loss / d fine_grid, summed over the batch (the reference repeats one grid B times, models.py:172: the sum is what
// reaches the shared parameters).  g_xw is the gradient w.r.t. the MASKED warped image; grid_sampler_2d_backward w.r.t. the
// grid, bilinear, zeros padding, align_corners=True: d/dgx = (W-1)/2 * sum_c g_c * d v_c / d x.
__global__ void warp_bwd_grid_kernel(const float4* __restrict__ g_xw, const float4* __restrict__ x,
                                     const float4* __restrict__ grid, const float* __restrict__ mask,
                                     float4* __restrict__ g_grid, int B, int Hp, int Wp, int HWc) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HWc) return;
    const float4 gr = grid[pix];
    const float xf = (gr.x + 1.f) * (0.5f * (float)(Wp - 1)), yf = (gr.y + 1.f) * (0.5f * (float)(Hp - 1));
    const float xw = floorf(xf), yn = floorf(yf);
    const float w = xf - xw, e = 1.f - w, n = yf - yn, s = 1.f - n;
    const int x0 = (int)xw, y0 = (int)yn;
    const float m = (mask != nullptr) ? mask[pix] : 1.f;
    float gx = 0.f, gy = 0.f;
    const bool vy0 = (unsigned)y0 < (unsigned)Hp, vy1 = (unsigned)(y0 + 1) < (unsigned)Hp;
    const bool vx0 = (unsigned)x0 < (unsigned)Wp, vx1 = (unsigned)(x0 + 1) < (unsigned)Wp;
    for (int b = 0; b < B; ++b) {
        const float4* xb = x + (size_t)b * Hp * Wp;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 nw = (vy0 && vx0) ? xb[y0 * Wp + x0] : z, ne = (vy0 && vx1) ? xb[y0 * Wp + x0 + 1] : z;
        const float4 sw = (vy1 && vx0) ? xb[(y0 + 1) * Wp + x0] : z, se = (vy1 && vx1) ? xb[(y0 + 1) * Wp + x0 + 1] : z;
        const float4 g = g_xw[(size_t)b * HWc + pix];
        // v = nw e s + ne w s + sw e n + se w n
        const float dvx0 = (ne.x - nw.x) * s + (se.x - sw.x) * n, dvy0 = (sw.x - nw.x) * e + (se.x - ne.x) * w;
        const float dvx1 = (ne.y - nw.y) * s + (se.y - sw.y) * n, dvy1 = (sw.y - nw.y) * e + (se.y - ne.y) * w;
        const float dvx2 = (ne.z - nw.z) * s + (se.z - sw.z) * n, dvy2 = (sw.z - nw.z) * e + (se.z - ne.z) * w;
        gx += g.x * dvx0 + g.y * dvx1 + g.z * dvx2;
        gy += g.x * dvy0 + g.y * dvy1 + g.z * dvy2;
    }
    g_grid[pix] = make_float4(gx * m * (0.5f * (float)(Wp - 1)), gy * m * (0.5f * (float)(Hp - 1)), 0.f, 0.f);
}

// fine = clamp(refine + coarse, -1, 1) (models.py:176): the clamp passes the gradient where -1 <= v <= 1; the refine
// net's last activation is LeakyReLU(0.1).  g_sum -> gradient w.r.t. (refine + coarse); g_r6 -> w.r.t. the last layer's
// pre-activation.
__global__ void finish_grid_bwd_kernel(const float4* __restrict__ g_fine, const float4* __restrict__ coarse,
                                       const float4* __restrict__ refine, float4* __restrict__ g_sum,
                                       float4* __restrict__ g_r6, int npix) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npix) return;
    const float4 g = g_fine[idx], c = coarse[idx];
    float vx = c.x, vy = c.y;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (refine != nullptr) {
        r = refine[idx];
        vx += r.x;
        vy += r.y;
    }
    const float gx = (vx >= -1.f && vx <= 1.f) ? g.x : 0.f, gy = (vy >= -1.f && vy <= 1.f) ? g.y : 0.f;
    g_sum[idx] = make_float4(gx, gy, 0.f, 0.f);
    if (g_r6 != nullptr) g_r6[idx] = make_float4(r.x > 0.f ? gx : 0.1f * gx, r.y > 0.f ? gy : 0.1f * gy, 0.f, 0.f);
}

// Gradient of the coarse grid (coarse_grid_kernel in warp.hip) w.r.t. affine_mat [6] and theta [(T+2) x 2]:
// partial[block][6 + 2 (T+2)].  NP = number of outputs, at most 6 + 2 * (64 + 2).
constexpr int MAXT = 64;
__global__ __launch_bounds__(256) void coarse_grid_bwd_kernel(const float4* __restrict__ g_coarse, const float* __restrict__ affine6,
                                                              const float* __restrict__ theta, const float* __restrict__ ctrl,
                                                              int T, int Hin, int Win, int Hout, int Wout,
                                                              float* __restrict__ partial) {
    __shared__ float red[4];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < Hout * Wout;
    const int oy = live ? idx / Wout : 0, ox = live ? idx - oy * Wout : 0;
    const float px = linspace_at(0.f, 1.f, Wout, ox), py = linspace_at(0.f, 1.f, Hout, oy);
    // forward recomputation (same arithmetic as coarse_grid_kernel)
    float u[MAXT];
    float bx = 0.f, by = 0.f, wsx = 0.f, wsy = 0.f;
    for (int t = 0; t < T; ++t) {
        const float dx = px - ctrl[2 * t], dy = py - ctrl[2 * t + 1];
        const float d = sqrtf(dx * dx + dy * dy);
        u[t] = (d * d) * logf(d + 1e-6f);
        if (t > 0) {
            const float wx = theta[2 * (t - 1)], wy = theta[2 * (t - 1) + 1];
            bx += u[t] * wx;
            by += u[t] * wy;
            wsx += wx;
            wsy += wy;
        }
    }
    bx += u[0] * (-wsx);
    by += u[0] * (-wsy);
    const float* a = theta + 2 * (T - 1);
    const float zx = (a[0] + px * a[2] + py * a[4]) + bx, zy = (a[1] + px * a[3] + py * a[5]) + by;
    const float tx = (px + zx) * 2.f - 1.f, ty = (py + zy) * 2.f - 1.f;
    const float xf = (tx + 1.f) * (0.5f * (float)(Win - 1)), yf = (ty + 1.f) * (0.5f * (float)(Hin - 1));
    const float xw = floorf(xf), yn = floorf(yf);
    const float w = xf - xw, e = 1.f - w, n = yf - yn, s = 1.f - n;
    const int x0 = (int)xw, y0 = (int)yn;
    const float4 g = live ? g_coarse[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    // affine values at the four taps (zero outside the affine grid) and the tap-wise sums for d/d affine
    float ax[4], ay[4], sbx = 0.f, sby = 0.f, s1 = 0.f;
    const int ty4[4] = {y0, y0, y0 + 1, y0 + 1}, tx4[4] = {x0, x0 + 1, x0, x0 + 1};
    const float wg[4] = {e * s, w * s, e * n, w * n};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ax[k] = ay[k] = 0.f;
        if ((unsigned)ty4[k] < (unsigned)Hin && (unsigned)tx4[k] < (unsigned)Win) {
            const float bxn = linspace_at(-1.f, 1.f, Win, tx4[k]), byn = linspace_at(-1.f, 1.f, Hin, ty4[k]);
            ax[k] = bxn * affine6[0] + byn * affine6[1] + affine6[2];
            ay[k] = bxn * affine6[3] + byn * affine6[4] + affine6[5];
            sbx += bxn * wg[k];
            sby += byn * wg[k];
            s1 += wg[k];
        }
    }
    float out[6];
    out[0] = g.x * sbx; out[1] = g.x * sby; out[2] = g.x * s1;
    out[3] = g.y * sbx; out[4] = g.y * sby; out[5] = g.y * s1;
    // d (gx, gy) / d (tx, ty), then tx = (px + zx) * 2 - 1
    const float dgx_dx = ((ax[1] - ax[0]) * s + (ax[3] - ax[2]) * n) * (0.5f * (float)(Win - 1));
    const float dgx_dy = ((ax[2] - ax[0]) * e + (ax[3] - ax[1]) * w) * (0.5f * (float)(Hin - 1));
    const float dgy_dx = ((ay[1] - ay[0]) * s + (ay[3] - ay[2]) * n) * (0.5f * (float)(Win - 1));
    const float dgy_dy = ((ay[2] - ay[0]) * e + (ay[3] - ay[1]) * w) * (0.5f * (float)(Hin - 1));
    const float gzx = 2.f * (g.x * dgx_dx + g.y * dgy_dx), gzy = 2.f * (g.x * dgx_dy + g.y * dgy_dy);
    float* pp = partial + (size_t)blockIdx.x * (6 + 2 * (T + 2));
    for (int k = 0; k < 6; ++k) {
        const float r = block_sum(out[k], red);
        if (threadIdx.x == 0) pp[k] = r;
    }
    for (int t = 1; t < T; ++t) {   // theta rows 0 .. T-2: the free TPS weights (w_0 = -sum of the others)
        const float du = u[t] - u[0];
        const float rx = block_sum(gzx * du, red), ry = block_sum(gzy * du, red);
        if (threadIdx.x == 0) {
            pp[6 + 2 * (t - 1)] = rx;
            pp[6 + 2 * (t - 1) + 1] = ry;
        }
    }
    const float basis[3] = {1.f, px, py};   // theta rows T-1 .. T+1: the affine part of the spline
    for (int k = 0; k < 3; ++k) {
        const float rx = block_sum(gzx * basis[k], red), ry = block_sum(gzy * basis[k], red);
        if (threadIdx.x == 0) {
            pp[6 + 2 * (T - 1 + k)] = rx;
            pp[6 + 2 * (T - 1 + k) + 1] = ry;
        }
    }
}

// out = (act > 0) ? g : 0 over float4 elements (ReLU backward where no convolution epilogue is at hand)
__global__ void relu_gate_kernel(const float4* __restrict__ g, const float4* __restrict__ act, float4* __restrict__ out, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = g[i], a = act[i];
    out[i] = make_float4(a.x > 0.f ? v.x : 0.f, a.y > 0.f ? v.y : 0.f, a.z > 0.f ? v.z : 0.f, a.w > 0.f ? v.w : 0.f);
}

__global__ void sum_rows_kernel(const float* __restrict__ partial, float* __restrict__ out, int nrows, int ncols) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    float s = 0.f;
    for (int r = 0; r < nrows; ++r) s += partial[(size_t)r * ncols + c];
    out[c] = s;
}

// torch.optim.Adam (amsgrad=False, maximize=False): L2 weight decay folded into the gradient
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int64_t n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
}

}  // namespace

extern "C" {

int spaa_warp_bwd_grid(const float* g_xw, const float* x, const float* grid, const float* mask, float* g_grid, int B, int Hp,
                       int Wp, int Hc, int Wc, spaa_stream_t stream) {
    if (!g_xw || !x || !grid || !g_grid || B < 1 || Hp < 1 || Wp < 1 || Hc < 1 || Wc < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(warp_bwd_grid_kernel, dim3((Hc * Wc + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float4*)g_xw,
                       (const float4*)x, (const float4*)grid, mask, (float4*)g_grid, B, Hp, Wp, Hc * Wc);
    return (int)hipGetLastError();
}

int spaa_warp_finish_grid_bwd(const float* g_fine, const float* coarse, const float* refine, float* g_sum, float* g_r6,
                              int npix, spaa_stream_t stream) {
    if (!g_fine || !coarse || !g_sum || npix < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(finish_grid_bwd_kernel, dim3((npix + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float4*)g_fine,
                       (const float4*)coarse, (const float4*)refine, (float4*)g_sum, (float4*)g_r6, npix);
    return (int)hipGetLastError();
}

int spaa_warp_coarse_grid_bwd(const float* g_coarse, const float* affine6, const float* theta, const float* ctrl, int T, int Hin,
                              int Win, int Hout, int Wout, float* partial, float* g_params, spaa_stream_t stream) {
    if (!g_coarse || !affine6 || !theta || !ctrl || !partial || !g_params || T < 2 || T > MAXT || Hin < 1 || Win < 1 ||
        Hout < 1 || Wout < 1)
        return hipErrorInvalidValue;
    const int nblk = (Hout * Wout + 255) / 256, ncols = 6 + 2 * (T + 2);
    hipLaunchKernelGGL(coarse_grid_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float4*)g_coarse, affine6,
                       theta, ctrl, T, Hin, Win, Hout, Wout, partial);
    hipLaunchKernelGGL(sum_rows_kernel, dim3((ncols + 63) / 64), dim3(64), 0, (hipStream_t)stream, partial, g_params, nblk, ncols);
    return (int)hipGetLastError();
}

int spaa_relu_gate(const float* g, const float* act, float* out, int64_t n, spaa_stream_t stream) {
    if (!g || !act || !out || n < 4 || (n & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(relu_gate_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)g,
                       (const float4*)act, (float4*)out, n / 4);
    return (int)hipGetLastError();
}

int spaa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, double beta1,
                   double beta2, float eps, float weight_decay, int step, spaa_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) return hipErrorInvalidValue;
    // bias corrections in double from the double betas, as torch.optim.Adam forms them (1 - 0.999f^t is off by 1.3e-5 at t = 1)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, lr, (float)beta1, (float)beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2));
    return (int)hipGetLastError();
}

}  // extern "C"
