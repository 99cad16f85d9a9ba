// fs2_h16.hip — fp16-storage mode: the FRACTIONAL-STRIDE 3 x 3 layers of ShadingNetSPAA as one persistent, barrier-free kernel:
//     transConv1(x5) + skipConv2(x1)                      ConvTranspose2d(128, 64, 3, 2, 1, 1) + Conv2d(32, 64, 1)   models.py:237,293,299
//     conv2^T(g2) + skipConv2^T(g6),  conv2_s^T(gs2)      aten::convolution_backward(input) of Conv2d(32, 64, 3, 2, 1)  models.py:224,230
// (paths relative to /root/reference/src/python).  Until round 6 these ran on the patch-staged fp16 kernel with the four output-parity
// classes FOLDED into the GEMM columns: every class multiplied all four pixels of its 2 x 2 input window, 16 (class, tap) products for
// the 9 real ones, one workgroup per compute unit with its patch load, products and epilogue one after the other: 150 + 95 + 55 us at
// batch 64 against byte / FLOP bounds of 54 + 48 + 34.
//
// Here the layer is written per INPUT pixel.  Input pixel (y, x) and its three neighbours -- I00 = in[y][x], I01 = in[y][x + 1],
// I10 = in[y + 1][x], I11 = in[y + 1][x + 1] -- are all that the 2 x 2 output pixels (2 y + cy, 2 x + cx) read:
//     class (cy, cx) takes operand I_rq iff r <= cy and q <= cx, through tap ky = (cy == 0 ? 1 : r == 0 ? 2 : 0), kx likewise
// = nine (operand, class) pairs = the nine taps.  One v_mfma_f32_16x16x32_f16 per (pair, 16 output channels, 32 input channels): rows =
// output channels, columns = 16 consecutive input pixels of a row.  Exactly the layer's FLOPs.
//   * ALL weights live in LDS for the whole launch, already in the MFMA's per-lane operand layout (host-packed: [K step][pair][16-channel
//     block][64 lanes][8 fp16]; 147 KB for 128 -> 64 channels): no weight streaming, no barrier after the prologue;
//   * the pixel operands are 16-byte-per-lane GLOBAL loads of whole row segments (a lane = (pixel, 8-channel chunk)): no patch staging;
//   * a wave owns 32 input pixels of a row (two column groups share every weight operand read: one LDS read per two MFMAs) and walks
//     tasks (image, row, 32-pixel segment) on its own: 8 waves per workgroup, one workgroup per compute unit, nobody waits for anybody;
//   * optional second source at OUTPUT resolution (a 1 x 1 convolution added before bias / residual / activation: models.py:293,299 and
//     its mirror image in the backward pass): per class one more operand load and Cin2 / 32 products;
//   * epilogue from the accumulators: a lane holds 4 consecutive channels of one output pixel: bias, residual, ReLU, byte-mask gate, one
//     8-byte store, one gate byte of the stored value.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "launch_util.hpp"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct fs2_args {
    const _Float16* in;      // [B, Hi, Wi, in_cstride], channels [0, Cin)
    const _Float16* in2;     // [B, 2 Hi, 2 Wi, in2_cstride], channels [0, Cin2), or NULL
    const _Float16* w_img;   // [Cin / 32][9 pairs][COUT / 16][64 lanes][8]
    const _Float16* w2_img;  // [Cin2 / 32][COUT / 16][64 lanes][8], or NULL
    const float* bias;       // [COUT] or NULL
    const _Float16* add;     // residual [B, 2 Hi, 2 Wi, COUT] or NULL
    const uint8_t* gate_bits;  // [B, 2 Hi, 2 Wi, COUT / 4] or NULL: out = bit ? v : 0
    _Float16* out;           // [B, 2 Hi, 2 Wi, COUT]
    uint8_t* mask_out;       // [B, 2 Hi, 2 Wi, COUT / 4] or NULL
    int B, Hi, Wi, in_cstride, in2_cstride, ks1, ks2, relu, nseg;   // ks1 = Cin / 32, ks2 = Cin2 / 32, nseg = 32-pixel segments per row
};

// the nine (operand rq = 2 r + q, class cl = 2 cy + cx) pairs, in the order of the weight image
__device__ constexpr int PAIR_RQ[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
__device__ constexpr int PAIR_CL[9] = {0, 1, 2, 3, 1, 3, 2, 3, 3};

// NW waves per workgroup: 8 with one workgroup per compute unit (COUT = 64: up to 152 KB of weights, 256 registers), 4 with three
// (COUT = 32: at most 40 KB of weights, 168 registers: twelve waves per compute unit hide the loads of these byte-bound layers)
template <int COUT, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : 1) void fs2_h16_kernel(const fs2_args p) {
    constexpr int NRB = COUT / 16;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    // ---- prologue: the weight images into LDS as they are (1 KB pieces, LDS-DMA)
    const int n1 = p.ks1 * 9 * NRB, n2 = p.ks2 * NRB;      // pieces
    {
        const uint64_t a1 = reinterpret_cast<uint64_t>(p.w_img);
        const auto r1 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(a1 >> 32)) << 32) |
                                                                                  (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a1)),
                                                          0, n1 * 1024, 0x00020000);
        for (int i = wave; i < n1; i += NW) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr_t)(smem + i * 1024), 16, lane * 16, i * 1024, 0, 0);
        if (n2 > 0) {
            const uint64_t a2 = reinterpret_cast<uint64_t>(p.w2_img);
            const auto r2 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(a2 >> 32)) << 32) |
                                                                                      (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a2)),
                                                              0, n2 * 1024, 0x00020000);
            for (int i = wave; i < n2; i += NW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr_t)(smem + (n1 + i) * 1024), 16, lane * 16, i * 1024, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned char* const wl = smem + lane * 16;
    const unsigned char* const wl2 = smem + n1 * 1024 + lane * 16;

    const int Ho = 2 * p.Hi, Wo = 2 * p.Wi;
    const int ntask = p.B * p.Hi * p.nseg;
    const int wid = blockIdx.x * NW + wave, nwv = gridDim.x * NW;
    // every tensor through a buffer descriptor with 32-bit byte offsets: a pixel that does not exist is the out-of-range offset (loads give
    // zero, stores are dropped) -- no branch around any memory operation, so that the waits on them are counted, not drained
    constexpr int OOB = (int)0x80000000;
    auto mk = [](const void* ptr, const int64_t bytes) {
        const uint64_t a = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(ptr != nullptr ? (int)bytes : 0), 0x00020000);
    };
    const int64_t npx_o = (int64_t)p.B * Ho * Wo;
    const auto r_in = mk(p.in, (int64_t)p.B * p.Hi * p.Wi * p.in_cstride * 2);
    const auto r_in2 = mk(p.in2, npx_o * p.in2_cstride * 2);
    const auto r_add = mk(p.add, npx_o * COUT * 2);
    const auto r_gate = mk(p.gate_bits, npx_o * (COUT / 4));
    const auto r_out = mk(p.out, npx_o * COUT * 2);
    const auto r_mask = mk(p.mask_out, npx_o * (COUT / 4));
    const auto r_bias = mk(p.bias, COUT * 4);
    const bool has_gate = p.gate_bits != nullptr;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const int pxb = p.in_cstride * 2, pxb2 = p.in2_cstride * 2;     // bytes per pixel

    for (int t = wid; t < ntask; t += nwv) {
        const int seg = t % p.nseg, y = (t / p.nseg) % p.Hi, b = t / (p.nseg * p.Hi);
        const int x0 = 32 * seg;
        f32x4 acc[2][4][NRB];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl)
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) acc[gi][cl][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // byte offsets of this lane's four operands (pixel (y + r, x0 + 16 gi + j + q), chunk g) at K step 0, or OOB
        const bool y1ok = y + 1 < p.Hi;
        int voff[2][4];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int x = x0 + 16 * gi + j + (rq & 1);
                const bool ok = x < p.Wi && (!(rq & 2) || y1ok);
                voff[gi][rq] = ok ? ((b * p.Hi + y + (rq >> 1)) * p.Wi + x) * pxb + 16 * g : OOB;
            }
        // One operand register set I[group][slot], four slots.  Main K step ks: slot rq holds in[y + r][x + q] (channels 32 ks ..); second-source
        // K step k2: slot cl holds in2[2 y + cy][2 x + cx] (channels 32 k2 ..).  The operand of slot s for step + 1 is requested right after the
        // LAST use of slot s in the current step (main steps use the slots in the order 0 0 0 0 1 1 2 2 3, second-source steps 0 1 2 3): a rolling
        // reload into the same registers, five to eight groups of products of cover per load and no second buffer (the accumulators leave no
        // room for one: two waves per SIMD, 256 registers each).  Steps are numbered 0 .. ks1 + ks2 - 1; a step past the end fetches nothing
        // (out-of-range offset).  No branch depends on the step: descriptor and offsets are selected, so the waits stay counted.
        h8 I[2][4];
        int o2[2];      // byte offset of output pixel (2 y, 2 x) in in2, chunk g
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int x = x0 + 16 * gi + j;
            o2[gi] = x < p.Wi ? ((b * Ho + 2 * y) * Wo + 2 * x) * pxb2 + 16 * g : OOB;
        }
        const int nstep = p.ks1 + p.ks2;
        auto fetch = [&](const int slot, const int step) {
            const bool is_main = step < p.ks1, is_sec = !is_main && step < nstep;     // (uniform)
            const auto rs = is_main ? r_in : r_in2;
            const int soff = is_main ? 64 * step : 64 * (step - p.ks1);
            const int add2 = ((slot >> 1) * Wo + (slot & 1)) * pxb2;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int off = is_main ? voff[gi][slot] : ((is_sec && o2[gi] != OOB) ? o2[gi] + add2 : OOB);
                I[gi][slot] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0));
            }
        };
#pragma unroll
        for (int slot = 0; slot < 4; ++slot) fetch(slot, 0);
#pragma unroll 1
        for (int ks = 0; ks < p.ks1; ++ks) {
            const unsigned char* wk = wl + ks * (9 * NRB * 1024);
#pragma unroll
            for (int pr = 0; pr < 9; ++pr) {
                const int rq = PAIR_RQ[pr], cl = PAIR_CL[pr];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const h8 A = *reinterpret_cast<const h8*>(wk + (pr * NRB + rb) * 1024);
                    acc[0][cl][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, I[0][rq], acc[0][cl][rb], 0, 0, 0);
                    acc[1][cl][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, I[1][rq], acc[1][cl][rb], 0, 0, 0);
                }
                if (pr == 8 || PAIR_RQ[pr + 1 > 8 ? 8 : pr + 1] != rq) fetch(rq, ks + 1);     // (compile time: the last pair that reads slot rq)
            }
        }
        // ---- second source at output resolution: class (cy, cx) of input pixel (y, x) = output pixel (2 y + cy, 2 x + cx)
#pragma unroll 1
        for (int k2 = 0; k2 < p.ks2; ++k2) {
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) {
                    const h8 A = *reinterpret_cast<const h8*>(wl2 + (k2 * NRB + rb) * 1024);
                    acc[0][cl][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, I[0][cl], acc[0][cl][rb], 0, 0, 0);
                    acc[1][cl][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, I[1][cl], acc[1][cl][rb], 0, 0, 0);
                }
                fetch(cl, p.ks1 + k2 + 1);
            }
        }
        // ---- epilogue.  D: column = input pixel j, rows 16 rb + 4 g + e.  The host packs the weight ROWS so that row 16 rb + 4 g + e is output
        // channel 32 (rb >> 1) + 8 g + 4 (rb & 1) + e: a lane then holds, per pair of row blocks, EIGHT consecutive channels of its output
        // pixel -- one 16-byte store (the four lanes of a pixel write 64 contiguous bytes) and one 2-byte gate store per pair, where the
        // natural row order gave 8-byte stores 32 bytes apart (measured: 55 us of a 91 us launch were its stores).
        // Every operand (bias, residual, gate bytes) is requested BEFORE any value is finished.
        constexpr int NP = NRB / 2;        // row-block pairs: 16-byte channel groups of a lane
        f32x4 bq[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r_bias, (32 * (rb >> 1) + 8 * g + 4 * (rb & 1)) * 4, 0, 0);
            bq[rb] = f32x4{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int x = x0 + 16 * gi + j;
            const bool xok = x < p.Wi;
            const int opx = (b * Ho + 2 * y) * Wo + 2 * x;        // output pixel (2 y, 2 x)
            u32x4 av[4][NP];
            unsigned int gb[4][NP];
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int o = opx + (cl >> 1) * Wo + (cl & 1);
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    const int n0 = 32 * pp + 8 * g;
                    av[cl][pp] = __builtin_amdgcn_raw_buffer_load_b128(r_add, xok ? (o * COUT + n0) * 2 : OOB, 0, 0);
                    gb[cl][pp] = __builtin_amdgcn_raw_buffer_load_b16(r_gate, xok ? o * (COUT / 4) + (n0 >> 2) : OOB, 0, 0);
                }
            }
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int o = opx + (cl >> 1) * Wo + (cl & 1);
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    const int n0 = 32 * pp + 8 * g;
                    const h8 ah = __builtin_bit_cast(h8, av[cl][pp]);
                    const unsigned int gq = has_gate ? gb[cl][pp] : 0xffffu;
                    h8 hv;
                    unsigned int mb = 0;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        f32x4 v = acc[gi][cl][2 * pp + half] + bq[2 * pp + half];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = v[e] + (float)ah[4 * half + e];
                            t = p.relu ? fmaxf(t, 0.f) : t;
                            t = ((gq >> (8 * half + e)) & 1u) ? t : 0.f;
                            hv[4 * half + e] = (_Float16)t;
                            mb |= (hv[4 * half + e] > (_Float16)0 ? 1u : 0u) << (8 * half + e);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), r_out, xok ? (o * COUT + n0) * 2 : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)mb, r_mask, xok ? o * (COUT / 4) + (n0 >> 2) : OOB, 0, 0);
                }
            }
        }
    }
}

}  // namespace

extern "C" int spaa_fs2_h16(const void* in, int in_cstride, int Cin, const void* w_img, const void* in2, int in2_cstride, int Cin2,
                            const void* w2_img, const float* bias, const void* add, const uint8_t* gate_bits, int relu, void* out,
                            uint8_t* mask_out, int Cout, int B, int Hi, int Wi, spaa_stream_t stream) {
    if (!in || !w_img || !out || B < 1 || Hi < 1 || Wi < 1 || Cin < 32 || (Cin & 31) || in_cstride < Cin || (in_cstride & 7) ||
        (Cout != 32 && Cout != 64))
        return hipErrorInvalidValue;
    if ((in2 != nullptr) != (w2_img != nullptr) || (in2 != nullptr && (Cin2 < 32 || (Cin2 & 31) || in2_cstride < Cin2 || (in2_cstride & 7))))
        return hipErrorInvalidValue;
    if (in2 == nullptr) Cin2 = 0;
    const size_t smem = (size_t)(Cin / 32 * 9 + Cin2 / 32) * (Cout / 16) * 1024;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    {   // 32-bit buffer offsets: every tensor below 2 GiB
        const int64_t npx_i = (int64_t)B * Hi * Wi, lim = (int64_t)1 << 31;
        if (npx_i * in_cstride * 2 >= lim || 4 * npx_i * Cout * 2 >= lim || (in2 != nullptr && 4 * npx_i * in2_cstride * 2 >= lim)) return hipErrorInvalidValue;
    }
    fs2_args a;
    a.in = reinterpret_cast<const _Float16*>(in), a.in2 = reinterpret_cast<const _Float16*>(in2);
    a.w_img = reinterpret_cast<const _Float16*>(w_img), a.w2_img = reinterpret_cast<const _Float16*>(w2_img);
    a.bias = bias, a.add = reinterpret_cast<const _Float16*>(add), a.gate_bits = gate_bits;
    a.out = reinterpret_cast<_Float16*>(out), a.mask_out = mask_out;
    a.B = B, a.Hi = Hi, a.Wi = Wi, a.in_cstride = in_cstride, a.in2_cstride = in2_cstride, a.ks1 = Cin / 32, a.ks2 = Cin2 / 32, a.relu = relu;
    a.nseg = (Wi + 31) / 32;
    const int64_t ntask = (int64_t)B * Hi * a.nseg;
    if (ntask > 0x7fffffff) return hipErrorInvalidValue;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set[2][SPAA_MAX_DEVICES] = {};
    if (Cout == 64) {    // persistent: one workgroup of eight waves per compute unit (its weights fill the LDS)
        int64_t nwg = (ntask + 7) / 8;
        if (nwg > ncu) nwg = ncu;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&fs2_h16_kernel<64, 8>), 160 * 1024, attr_set[0]);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((fs2_h16_kernel<64, 8>), dim3((unsigned)nwg), dim3(512), smem, st, a);
    } else {             // three workgroups of four waves per compute unit
        if (smem > 52 * 1024) return hipErrorInvalidValue;
        int64_t nwg = (ntask + 3) / 4;
        if (nwg > 3 * (int64_t)ncu) nwg = 3 * (int64_t)ncu;
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&fs2_h16_kernel<32, 4>), 52 * 1024, attr_set[1]);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((fs2_h16_kernel<32, 4>), dim3((unsigned)nwg), dim3(256), smem, st, a);
    }
    return (int)hipGetLastError();
}
