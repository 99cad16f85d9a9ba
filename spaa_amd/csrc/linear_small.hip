// linear_small.hip — nn.Linear on a handful of rows (the classifiers' last layer at batch <= 64: 512 / 2048 / 4096 -> 1000 logits,
// and its input gradient): out[m][n] = bias[n] + sum_k x[m][k] * w[n][k].
// (/root/reference/src/python/classifier.py:60 runs torchvision's `fc` through ATen addmm; backward: mm with the weight.)
// As a 1 x 1 convolution on the implicit-GEMM tiles this GEMM is one or two workgroups of work (M = 64 rows) and took 20 us
// forward (eight K ranges + a second pass) and 37 us backward on a chip that is otherwise idle; here a wave keeps NPW weight rows in
// registers and takes 16 of the M rows per pass (a row of x is a coalesced 16-byte load per lane, L2-resident), one fixed-order
// butterfly sum per output: 250-1000 waves, about 5 us.  fp32 FMAs, deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// sum over the 64 lanes in a fixed order, on the VALU's cross-lane data paths (no LDS round trips); the total lands in lane 63
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_term(const float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, BOUND));
}
__device__ __forceinline__ float wave_sum63(float v) {
    v += dpp_term<0xB1, 0xF, true>(v);     // quad_perm [1,0,3,2]
    v += dpp_term<0x4E, 0xF, true>(v);     // quad_perm [2,3,0,1]
    v += dpp_term<0x141, 0xF, true>(v);    // row_half_mirror
    v += dpp_term<0x140, 0xF, true>(v);    // row_mirror: every lane of a 16-lane row holds the row's sum
    v += dpp_term<0x142, 0xA, false>(v);   // row_bcast15 into rows 1 and 3
    v += dpp_term<0x143, 0xC, false>(v);   // row_bcast31 into rows 2 and 3
    return v;
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float* ptr, const uint32_t bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ f4 load4(const __amdgpu_buffer_rsrc_t r, const int off) {   // (out-of-range offset: zeros, no branch)
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// KV: 16-byte pieces of a row per lane (K <= 256 KV); NPW: weight rows (outputs) per wave; RP: rows of x per wave and pass
template <int KV, int NPW, int RP>
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, const int M,
                                                           const int K, const int N, const int ldx, const int ldw, const int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * NPW;
    constexpr int OOB = (int)0x80000000;
    const auto rx = rsrc_of(x, (uint32_t)M * (uint32_t)ldx * 4u), rw = rsrc_of(w, (uint32_t)N * (uint32_t)ldw * 4u);
    f4 wr[NPW][KV];
#pragma unroll
    for (int j = 0; j < NPW; ++j)
#pragma unroll
        for (int v = 0; v < KV; ++v) {
            const int k = (v * 64 + lane) * 4;
            wr[j][v] = load4(rw, (n0 + j < N && k < K) ? ((n0 + j) * ldw + k) * 4 : OOB);
        }
    // RP rows of x per wave and pass (four waves: 4 RP rows): all loads of a pass are issued before the first sum is needed, and the
    // RP x NPW butterfly sums are independent chains (a row at a time the loop was a load latency + six dependent cross-lane steps
    // per row: 25 us instead of 5)
    for (int mb = 0; mb < M; mb += 4 * RP) {
        const int m0 = mb + wave * RP;
        float sum[RP][NPW];
#pragma unroll
        for (int i = 0; i < RP; ++i) {
            f4 xv[KV];
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                const int k = (v * 64 + lane) * 4;
                xv[v] = load4(rx, (m0 + i < M && k < K) ? ((m0 + i) * ldx + k) * 4 : OOB);
            }
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                float t = 0.f;
#pragma unroll
                for (int v = 0; v < KV; ++v) {
                    t = fmaf(xv[v].x, wr[j][v].x, t);
                    t = fmaf(xv[v].y, wr[j][v].y, t);
                    t = fmaf(xv[v].z, wr[j][v].z, t);
                    t = fmaf(xv[v].w, wr[j][v].w, t);
                }
                sum[i][j] = t;
            }
        }
#pragma unroll
        for (int i = 0; i < RP; ++i)
#pragma unroll
            for (int j = 0; j < NPW; ++j) sum[i][j] = wave_sum63(sum[i][j]);
        if (lane == 63) {
#pragma unroll
            for (int i = 0; i < RP; ++i)
#pragma unroll
                for (int j = 0; j < NPW; ++j)
                    if (m0 + i < M && n0 + j < N) out[(size_t)(m0 + i) * ldo + n0 + j] = sum[i][j] + (bias != nullptr ? bias[n0 + j] : 0.f);
        }
    }
}

template <int KV, int NPW, int RP>
int launch(const float* x, const float* w, const float* bias, float* out, int M, int K, int N, int ldx, int ldw, int ldo, hipStream_t st) {
    hipLaunchKernelGGL((linear_small_kernel<KV, NPW, RP>), dim3((N + NPW - 1) / NPW), dim3(256), 0, st, x, w, bias, out, M, K, N, ldx, ldw, ldo);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int spaa_linear_small(const float* x, const float* w, const float* bias, float* out, int M, int K, int N, int ldx, int ldw,
                                 int ldo, spaa_stream_t stream) {
    if (!x || !w || !out || M < 1 || M > 256 || K < 4 || K > 4096 || N < 1 || (K & 3) || (ldx & 3) || (ldw & 3) || ldx < K || ldw < K || ldo < N || (int64_t)N * ldw * 4 >= ((int64_t)1 << 31) ||
        ((uintptr_t)x & 15) || ((uintptr_t)w & 15))
        return hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    if (K <= 512) return launch<2, 4, 16>(x, w, bias, out, M, K, N, ldx, ldw, ldo, st);
    if (K <= 1024) return launch<4, 4, 8>(x, w, bias, out, M, K, N, ldx, ldw, ldo, st);
    if (K <= 2048) return launch<8, 2, 4>(x, w, bias, out, M, K, N, ldx, ldw, ldo, st);
    return launch<16, 1, 2>(x, w, bias, out, M, K, N, ldx, ldw, ldo, st);
}
