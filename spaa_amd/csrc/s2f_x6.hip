// s2f_x6.hip — fp32 mode: the FORWARD form of a 3 x 3 / stride-2 / padding-1 convolution with few input channels as a persistent,
// barrier-free kernel on the bf16x6 arithmetic (the fp32 sibling of csrc/s2f_h16.hip):
//     conv2(x1) + res2_s, conv2_s(res1_s)       Conv2d(32, 64, 3, 2, 1)                                       models.py:224,230,286,292
// (paths relative to /root/reference/src/python).  These ran on the implicit-GEMM bf16x6 tile (x6v2_128x64g2): every K step stages a
// weight tile and a gathered pixel tile through LDS behind barriers -- 117 + 91 us at batch 64 against ~55 us of matrix-core time at the
// rate the big layers reach and 68 / 50 us of HBM time at the practical streaming rate.
//
//   * ALL weights in LDS for the whole launch, split on the host into the three bf16 planes (w == h + m + l exactly) and laid out in the
//     MFMA's per-lane operand order ([K step][tap][plane][16-row block][64 lanes][8 bf16], rows permuted so that a lane ends with eight
//     consecutive channels per pair of row blocks); no barrier after the prologue;
//   * a wave owns 32 consecutive output pixels of a row (two groups of 16: every weight operand read feeds two pixel groups) and walks
//     tasks (image, row, 32-pixel segment) on its own; its pixel operands are 2 x 16-byte-per-lane buffer loads of every second input
//     pixel (a pixel that does not exist = the out-of-range offset = the zero padding), split exactly into three bf16 fragments in
//     registers, tap t + 2 requested before the products of tap t;
//   * arithmetic = that of the other bf16x6 kernels: six of the nine partial products, small terms first, fp32 accumulation;
//   * epilogue from the accumulators: bias, residual, ReLU, byte-mask gate, 2 x 16-byte stores per lane, 2-byte gate stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "launch_util.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct s2fx_args {
    const float* in;           // [B, Hi, Wi, in_cstride], channels [0, Cin)
    const uint16_t* w_img;     // [Cin / 32][9 taps][3 planes][COUT / 16][64 lanes][8] bf16
    const float* bias;         // [COUT] or NULL
    const float* add;          // [B, Ho, Wo, COUT] or NULL
    const uint8_t* gate_bits;  // [B, Ho, Wo, COUT / 4] or NULL
    float* out;                // [B, Ho, Wo, COUT]
    uint8_t* mask_out;         // [B, Ho, Wo, COUT / 4] or NULL
    int B, Hi, Wi, in_cstride, ks1, relu, nseg;
};

__device__ __forceinline__ unsigned int cvt2(float a, float b) {
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned int p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned int p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 fp32 -> three bf16x8 with x == h + m + l exactly (as csrc/tapconv_wino.hip: split8)
__device__ __forceinline__ void split8(const u32x4 a, const u32x4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float x[8] = {__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]), __uint_as_float(a[3]),
                        __uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3])};
    u32x4 hh, mm, ll;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned int ph = cvt2(x[2 * i], x[2 * i + 1]);
        const float r0 = x[2 * i] - lo_f(ph), r1 = x[2 * i + 1] - hi_f(ph);
        const unsigned int pm = cvt2(r0, r1);
        const float s0 = r0 - lo_f(pm), s1 = r1 - hi_f(pm);
        hh[i] = ph;
        mm[i] = pm;
        ll[i] = cvt2(s0, s1);
    }
    h = __builtin_bit_cast(bf16x8, hh);
    m = __builtin_bit_cast(bf16x8, mm);
    l = __builtin_bit_cast(bf16x8, ll);
}

// six of the nine partial products of (w0 + w1 + w2) . (p0 + p1 + p2), small terms first (csrc/tapconv_x6d.hip)
__device__ __forceinline__ f32x4 mfma6(const bf16x8 w0, const bf16x8 w1, const bf16x8 w2, const bf16x8 p0, const bf16x8 p1, const bf16x8 p2,
                                       f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, p0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, p2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, p1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, p0, acc, 0, 0, 0);
    return acc;
}

template <int COUT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void s2f_x6_kernel(const s2fx_args p) {
    constexpr int NRB = COUT / 16, NP = NRB / 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    // ---- prologue: the weight image into LDS as it is (1 KB pieces, LDS-DMA)
    const int n1 = p.ks1 * 9 * 3 * NRB;
    {
        const uint64_t a1 = reinterpret_cast<uint64_t>(p.w_img);
        const auto r1 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(a1 >> 32)) << 32) |
                                                                                  (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a1)),
                                                          0, n1 * 1024, 0x00020000);
        for (int i = wave; i < n1; i += NW) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr_t)(smem + i * 1024), 16, lane * 16, i * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned char* const wl = smem + lane * 16;

    // every tensor through a buffer descriptor with 32-bit byte offsets: a pixel that does not exist is the out-of-range offset (loads give
    // zero, stores are dropped) -- no branch around any memory operation, so that the waits on them are counted, not drained
    constexpr int OOB = (int)0x80000000;
    auto mk = [](const void* ptr, const int64_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(ptr != nullptr ? (int)bytes : 0), 0x00020000);
    };
    const int Ho = p.Hi >> 1, Wo = p.Wi >> 1;
    const int ntask = p.B * Ho * p.nseg;
    const int wid = blockIdx.x * NW + wave, nwv = gridDim.x * NW;
    const int64_t npx_o = (int64_t)p.B * Ho * Wo;
    const auto r_in = mk(p.in, (int64_t)p.B * p.Hi * p.Wi * p.in_cstride * 4);
    const auto r_add = mk(p.add, npx_o * COUT * 4);
    const auto r_gate = mk(p.gate_bits, npx_o * (COUT / 4));
    const auto r_out = mk(p.out, npx_o * COUT * 4);
    const auto r_mask = mk(p.mask_out, npx_o * (COUT / 4));
    const auto r_bias = mk(p.bias, COUT * 4);
    const bool has_gate = p.gate_bits != nullptr;
    const int pxb = p.in_cstride * 4;

    for (int t = wid; t < ntask; t += nwv) {
        const int seg = t % p.nseg, y = (t / p.nseg) % Ho, b = t / (p.nseg * Ho);
        const int x0 = 32 * seg;
        f32x4 acc[2][NRB];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[gi][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // THREE raw operand buffers (tap t in I[t % 3]): the operand of tap t + 2 -- of the next K step's first taps after the eighth -- is
        // requested before the products of tap t (nine taps per trip: the buffer index is a compile-time constant, and no branch sits
        // between the loads and their waits); past the last step: nothing (out-of-range offset).  A lane's two 16-byte loads are bytes
        // [16 g, 16 g + 16) of the pixel's first and second 64 bytes: a load instruction covers whole 64-byte segments, and K index
        // 8 g + e of a step is channel 4 g + e (e < 4) / 16 + 4 g + e - 4 (the host permutes the weight columns alike: pack_s2f_x6).
        u32x4 I[3][2][2];
        auto fetch = [&](const int buf, const int tap, const int ks) {
            const bool live = ks < p.ks1;      // (uniform)
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                // byte offset of input pixel (2 y - 1 + ky, 2 x - 1 + kx), or OOB (zero padding / past the row)
                const int x = x0 + 16 * gi + j;
                const int iy = 2 * y - 1 + ky, ix = 2 * x - 1 + kx;
                const bool ok = live && x < Wo && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                const int off = ok ? ((b * p.Hi + iy) * p.Wi + ix) * pxb + 16 * g : OOB;
                I[buf][gi][0] = __builtin_amdgcn_raw_buffer_load_b128(r_in, off, 128 * ks, 0);
                I[buf][gi][1] = __builtin_amdgcn_raw_buffer_load_b128(r_in, off + 64, 128 * ks, 0);
            }
        };
        fetch(0, 0, 0);
        fetch(1, 1, 0);
#pragma unroll 1
        for (int ks = 0; ks < p.ks1; ++ks) {
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int nu = u + 2;
                fetch(nu % 3, nu % 9, ks + nu / 9);
                __builtin_amdgcn_sched_barrier(0);      // (the requests stay ahead of the products: two taps of latency cover)
                const unsigned char* wk = wl + (ks * 9 + u) * (3 * NRB * 1024);
                bf16x8 P[2][3];
#pragma unroll
                for (int gi = 0; gi < 2; ++gi) split8(I[u % 3][gi][0], I[u % 3][gi][1], P[gi][0], P[gi][1], P[gi][2]);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wk + rb * 1024);
                    const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wk + (NRB + rb) * 1024);
                    const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wk + (2 * NRB + rb) * 1024);
                    acc[0][rb] = mfma6(w0, w1, w2, P[0][0], P[0][1], P[0][2], acc[0][rb]);
                    acc[1][rb] = mfma6(w0, w1, w2, P[1][0], P[1][1], P[1][2], acc[1][rb]);
                }
            }
        }
        // ---- epilogue.  D: column = output pixel j, rows 16 rb + 4 g + e = output channel 32 (rb >> 1) + 8 g + 4 (rb & 1) + e (host-permuted
        // weight rows): eight consecutive channels per pair of row blocks.  All operands requested before any value is finished.
        f32x4 bq[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r_bias, (32 * (rb >> 1) + 8 * g + 4 * (rb & 1)) * 4, 0, 0);
            bq[rb] = f32x4{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int x = x0 + 16 * gi + j;
            const bool xok = x < Wo;
            const int o = (b * Ho + y) * Wo + x;
            u32x4 av[NRB];
            unsigned int gb[NP];
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const int n0 = 32 * pp + 8 * g;
                av[2 * pp] = __builtin_amdgcn_raw_buffer_load_b128(r_add, xok ? (o * COUT + n0) * 4 : OOB, 0, 0);
                av[2 * pp + 1] = __builtin_amdgcn_raw_buffer_load_b128(r_add, xok ? (o * COUT + n0 + 4) * 4 : OOB, 0, 0);
                gb[pp] = __builtin_amdgcn_raw_buffer_load_b16(r_gate, xok ? o * (COUT / 4) + (n0 >> 2) : OOB, 0, 0);
            }
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const int n0 = 32 * pp + 8 * g;
                const unsigned int gq = has_gate ? gb[pp] : 0xffffu;
                unsigned int mb = 0;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4 v = acc[gi][2 * pp + half] + bq[2 * pp + half];
                    u32x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tv = v[e] + __uint_as_float(av[2 * pp + half][e]);
                        tv = p.relu ? fmaxf(tv, 0.f) : tv;
                        tv = ((gq >> (8 * half + e)) & 1u) ? tv : 0.f;
                        ov[e] = __float_as_uint(tv);
                        mb |= (tv > 0.f ? 1u : 0u) << (8 * half + e);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(ov, r_out, xok ? (o * COUT + n0 + 4 * half) * 4 : OOB, 0, 0);
                }
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)mb, r_mask, xok ? o * (COUT / 4) + (n0 >> 2) : OOB, 0, 0);
            }
        }
    }
}

}  // namespace

// Conv2d(Cin, Cout, 3, stride 2, padding 1) forward in fp32 (exact fp32 operands: bf16x6) with all weights resident in LDS:
// out[b][y][x][n] = epilogue(sum_{ky,kx,k} W[ky][kx][n][k] in[b][2 y - 1 + ky][2 x - 1 + kx][k]).  Cin % 32 == 0, Cout == 64, Hi, Wi even,
// (Cin / 32) x 9 x 3 x (Cout / 16) KB of weights <= 160 KB.
extern "C" int spaa_s2f_x6(const float* in, int in_cstride, int Cin, const void* w_img, const float* bias, const float* add,
                           const uint8_t* gate_bits, int relu, float* out, uint8_t* mask_out, int Cout, int B, int Hi, int Wi,
                           spaa_stream_t stream) {
    if (!in || !w_img || !out || B < 1 || Hi < 2 || Wi < 2 || (Hi & 1) || (Wi & 1) || Cin < 32 || (Cin & 31) || in_cstride < Cin ||
        (in_cstride & 3) || Cout != 64)
        return hipErrorInvalidValue;
    const size_t smem = (size_t)(Cin / 32) * 9 * 3 * (Cout / 16) * 1024;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if ((int64_t)B * Hi * Wi * in_cstride * 4 >= ((int64_t)1 << 31) || (int64_t)B * (Hi / 2) * (Wi / 2) * Cout * 4 >= ((int64_t)1 << 31))
        return hipErrorInvalidValue;      // 32-bit buffer offsets
    s2fx_args a;
    a.in = in, a.w_img = reinterpret_cast<const uint16_t*>(w_img), a.bias = bias, a.add = add, a.gate_bits = gate_bits;
    a.out = out, a.mask_out = mask_out;
    a.B = B, a.Hi = Hi, a.Wi = Wi, a.in_cstride = in_cstride, a.ks1 = Cin / 32, a.relu = relu;
    a.nseg = (Wi / 2 + 31) / 32;
    const int64_t ntask = (int64_t)B * (Hi / 2) * a.nseg;
    if (ntask > 0x7fffffff) return hipErrorInvalidValue;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    // persistent: one workgroup of eight waves per compute unit (its weights fill the LDS)
    int64_t nwg = (ntask + 7) / 8;
    if (nwg > ncu) nwg = ncu;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&s2f_x6_kernel<64, 8>), 160 * 1024, attr_set);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((s2f_x6_kernel<64, 8>), dim3((unsigned)nwg), dim3(512), smem, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}
