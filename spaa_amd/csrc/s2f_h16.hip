// s2f_h16.hip — fp16-storage mode: the FORWARD form of a 3 x 3 / stride-2 / padding-1 convolution as a persistent, barrier-free kernel
// (the sibling of csrc/fs2_h16.hip):
//     conv2(x1) + res2_s, conv2_s(res1_s)       Conv2d(32, 64, 3, 2, 1)                                       models.py:224,230,286,292
//     transConv1^T(g6)                          aten::convolution_backward(input) of ConvTranspose2d(128, 64, 3, 2, 1, 1)   models.py:237
// (paths relative to /root/reference/src/python).  These ran on the patch-staged fp16 kernel's stride-2 form: one workgroup per compute
// unit with a 70 KB patch reloaded per 32-channel block behind a barrier -- load, nine steps, epilogue strictly one after the other: 59 +
// 49 + 103 us at batch 64 against byte / FLOP bounds of 27 + 27 + 40.
//
//   * ALL weights in LDS for the whole launch in the MFMA's per-lane operand layout (host-packed [K step][tap][16-row block][64 lanes][8
//     fp16], rows permuted so that a lane ends with eight consecutive channels per pair of row blocks); no barrier after the prologue;
//   * a wave owns 32 consecutive output pixels of a row (two groups of 16: every weight operand read feeds two MFMAs) and walks tasks
//     (image, row, 32-pixel segment) on its own; its pixel operands are 16-byte-per-lane buffer loads of every second input pixel
//     (a pixel that does not exist = the out-of-range offset = the zero padding), tap t + 2 requested before the products of tap t;
//   * epilogue from the accumulators: bias, residual, ReLU, byte-mask gate, 16-byte stores, 2-byte gate stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "launch_util.hpp"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct s2f_args {
    const _Float16* in;        // [B, Hi, Wi, in_cstride], channels [0, Cin)
    const _Float16* w_img;     // [Cin / 32][9 taps][COUT / 16][64 lanes][8]
    const float* bias;         // [COUT] or NULL
    const _Float16* add;       // [B, Ho, Wo, COUT] or NULL
    const uint8_t* gate_bits;  // [B, Ho, Wo, COUT / 4] or NULL
    _Float16* out;             // [B, Ho, Wo, COUT]
    uint8_t* mask_out;         // [B, Ho, Wo, COUT / 4] or NULL
    int B, Hi, Wi, in_cstride, ks1, relu, nseg;
};

template <int COUT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : 1) void s2f_h16_kernel(const s2f_args p) {
    constexpr int NRB = COUT / 16, NP = NRB / 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    constexpr int OOB = (int)0x80000000;
    auto mk = [](const void* ptr, const int64_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(ptr != nullptr ? (int)bytes : 0), 0x00020000);
    };
    // ---- prologue: the weight image into LDS as it is (1 KB pieces, LDS-DMA)
    const int n1 = p.ks1 * 9 * NRB;
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

    const int Ho = p.Hi >> 1, Wo = p.Wi >> 1;
    const int ntask = p.B * Ho * p.nseg;
    const int wid = blockIdx.x * NW + wave, nwv = gridDim.x * NW;
    const int64_t npx_o = (int64_t)p.B * Ho * Wo;
    const auto r_in = mk(p.in, (int64_t)p.B * p.Hi * p.Wi * p.in_cstride * 2);
    const auto r_add = mk(p.add, npx_o * COUT * 2);
    const auto r_gate = mk(p.gate_bits, npx_o * (COUT / 4));
    const auto r_out = mk(p.out, npx_o * COUT * 2);
    const auto r_mask = mk(p.mask_out, npx_o * (COUT / 4));
    const auto r_bias = mk(p.bias, COUT * 4);
    const bool has_gate = p.gate_bits != nullptr;
    const int pxb = p.in_cstride * 2;

    for (int t = wid; t < ntask; t += nwv) {
        const int seg = t % p.nseg, y = (t / p.nseg) % Ho, b = t / (p.nseg * Ho);
        const int x0 = 32 * seg;
        f32x4 acc[2][NRB];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) acc[gi][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // byte offset of input pixel (2 y - 1 + ky, 2 x - 1 + kx), chunk g, or OOB (zero padding / a pixel past the image / the row)
        auto tap_off = [&](const int gi, const int tap) -> int {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int x = x0 + 16 * gi + j;
            const int iy = 2 * y - 1 + ky, ix = 2 * x - 1 + kx;
            const bool ok = x < Wo && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            return ok ? ((b * p.Hi + iy) * p.Wi + ix) * pxb + 16 * g : OOB;
        };
        // THREE operand buffers (tap t in I[t % 3]): the operand of tap t + 2 -- of the next K step's first taps after the eighth -- is
        // requested before the products of tap t (nine taps per trip: the buffer index is a compile-time constant and no branch sits
        // between the loads and their waits); past the last step: nothing (out-of-range offset).
        h8 I[3][2];
        auto fetch = [&](const int buf, const int tap, const int ks) {
            const bool live = ks < p.ks1;      // (uniform)
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int off = tap_off(gi, tap);
                I[buf][gi] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r_in, live ? off : OOB, 64 * ks, 0));
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
                const unsigned char* wk = wl + (ks * 9 + u) * (NRB * 1024);
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const h8 A = *reinterpret_cast<const h8*>(wk + rb * 1024);
                    acc[0][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, I[u % 3][0], acc[0][rb], 0, 0, 0);
                    acc[1][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, I[u % 3][1], acc[1][rb], 0, 0, 0);
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
            u32x4 av[NP];
            unsigned int gb[NP];
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const int n0 = 32 * pp + 8 * g;
                av[pp] = __builtin_amdgcn_raw_buffer_load_b128(r_add, xok ? (o * COUT + n0) * 2 : OOB, 0, 0);
                gb[pp] = __builtin_amdgcn_raw_buffer_load_b16(r_gate, xok ? o * (COUT / 4) + (n0 >> 2) : OOB, 0, 0);
            }
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                const int n0 = 32 * pp + 8 * g;
                const h8 ah = __builtin_bit_cast(h8, av[pp]);
                const unsigned int gq = has_gate ? gb[pp] : 0xffffu;
                h8 hv;
                unsigned int mb = 0;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4 v = acc[gi][2 * pp + half] + bq[2 * pp + half];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float tv = v[e] + (float)ah[4 * half + e];
                        tv = p.relu ? fmaxf(tv, 0.f) : tv;
                        tv = ((gq >> (8 * half + e)) & 1u) ? tv : 0.f;
                        hv[4 * half + e] = (_Float16)tv;
                        mb |= (hv[4 * half + e] > (_Float16)0 ? 1u : 0u) << (8 * half + e);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), r_out, xok ? (o * COUT + n0) * 2 : OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)mb, r_mask, xok ? o * (COUT / 4) + (n0 >> 2) : OOB, 0, 0);
            }
        }
    }
}

}  // namespace

extern "C" int spaa_s2f_h16(const void* in, int in_cstride, int Cin, const void* w_img, const float* bias, const void* add,
                            const uint8_t* gate_bits, int relu, void* out, uint8_t* mask_out, int Cout, int B, int Hi, int Wi,
                            spaa_stream_t stream) {
    if (!in || !w_img || !out || B < 1 || Hi < 2 || Wi < 2 || (Hi & 1) || (Wi & 1) || Cin < 32 || (Cin & 31) || in_cstride < Cin ||
        (in_cstride & 7) || (Cout != 64 && Cout != 128))
        return hipErrorInvalidValue;
    const size_t smem = (size_t)(Cin / 32) * 9 * (Cout / 16) * 1024;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if ((int64_t)B * Hi * Wi * in_cstride * 2 >= ((int64_t)1 << 31) || (int64_t)B * (Hi / 2) * (Wi / 2) * Cout * 2 >= ((int64_t)1 << 31))
        return hipErrorInvalidValue;      // 32-bit buffer offsets
    s2f_args a;
    a.in = reinterpret_cast<const _Float16*>(in), a.w_img = reinterpret_cast<const _Float16*>(w_img);
    a.bias = bias, a.add = reinterpret_cast<const _Float16*>(add), a.gate_bits = gate_bits;
    a.out = reinterpret_cast<_Float16*>(out), a.mask_out = mask_out;
    a.B = B, a.Hi = Hi, a.Wi = Wi, a.in_cstride = in_cstride, a.ks1 = Cin / 32, a.relu = relu;
    a.nseg = (Wi / 2 + 31) / 32;
    const int64_t ntask = (int64_t)B * (Hi / 2) * a.nseg;
    if (ntask > 0x7fffffff) return hipErrorInvalidValue;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set[3][SPAA_MAX_DEVICES] = {};
    if (smem > 52 * 1024) {      // persistent: one workgroup of eight waves per compute unit (its weights fill the LDS)
        int64_t nwg = (ntask + 7) / 8;
        if (nwg > ncu) nwg = ncu;
        if (Cout == 128) {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&s2f_h16_kernel<128, 8>), 160 * 1024, attr_set[0]);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((s2f_h16_kernel<128, 8>), dim3((unsigned)nwg), dim3(512), smem, st, a);
        } else {
            hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&s2f_h16_kernel<64, 8>), 160 * 1024, attr_set[1]);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((s2f_h16_kernel<64, 8>), dim3((unsigned)nwg), dim3(512), smem, st, a);
        }
    } else {                     // three workgroups of four waves per compute unit
        if (Cout != 64) return hipErrorInvalidValue;
        int64_t nwg = (ntask + 3) / 4;
        if (nwg > 3 * (int64_t)ncu) nwg = 3 * (int64_t)ncu;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&s2f_h16_kernel<64, 4>), 52 * 1024, attr_set[2]);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((s2f_h16_kernel<64, 4>), dim3((unsigned)nwg), dim3(256), smem, st, a);
    }
    return (int)hipGetLastError();
}
