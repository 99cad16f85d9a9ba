// tapconv_x6.hip — the tap-list convolution on the bf16 matrix cores with EXACT fp32 operands ("bf16x6").
//
// Every fp32 operand x is split into three bf16 values with x == h + m + l exactly (h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m): 3 x 8 significand bits cover fp32's 24; each residual is exact in fp32).  A product a*b is then
//     ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)        [+ terms below 2^-24 |a*b|, dropped]
// — six v_mfma_f32_32x32x16_bf16 (each product exact in fp32, fp32 accumulation in the matrix core) instead of eight
// v_mfma_f32_32x32x2_f32 per 16-deep K slice: 192 instead of 512 matrix-core cycles.  The three magnitude groups are
// accumulated in separate fp32 accumulators over the whole K loop and summed once at the end (small terms are not
// rounded against the large running sum), which makes the result at least as close to the exact dot product as an
// fp32 FMA chain: measured error vs fp64 is <= that of the fp32-MFMA kernel (tests/test_gpu_parity.py::
// test_tapconv_x6_is_fp32_accurate).  This is fp32 arithmetic emulated on wider hardware, not reduced precision.
//
// Data flow per K-step (32 fp32 channels of one tap):  activations: bounds-checked buffer loads (fp32) -> split in
// registers -> three bf16 planes in LDS (80-B rows: ds_write_b64 / ds_read_b128 conflict-free) -> A fragments;
// weights: pre-split on the host into three bf16 planes [Npad][Kpad]; a lane's B fragment (8 consecutive k of one
// output channel) is contiguous there, so waves load B fragments straight from L2 into registers (no LDS), one K-step
// ahead.  LDS holds only the gathered operand: 2 x 3 x BM x 80 B = 30 KB at BM = 64 -> 4-5 workgroups per CU.
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int BK = 32;
constexpr int ROWB = 80;  // LDS row pitch in bytes: 32 bf16 (64 B) + 16 B pad
constexpr int TAP_BYTES = 16 * (SPAA_MAX_TAPS + 4);

__device__ __forceinline__ unsigned int cvt2(float a, float b) {
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned int p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned int p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// x == h + m + l exactly, 4 values at a time, packed as 4 bf16 (8 bytes) per part
__device__ __forceinline__ void split4(const f4 x, u2& h, u2& m, u2& l) {
    h.x = cvt2(x.x, x.y);
    h.y = cvt2(x.z, x.w);
    const float r0 = x.x - lo_f(h.x), r1 = x.y - hi_f(h.x), r2 = x.z - lo_f(h.y), r3 = x.w - hi_f(h.y);
    m.x = cvt2(r0, r1);
    m.y = cvt2(r2, r3);
    const float s0 = r0 - lo_f(m.x), s1 = r1 - hi_f(m.x), s2 = r2 - lo_f(m.y), s3 = r3 - hi_f(m.y);
    l.x = cvt2(s0, s1);
    l.y = cvt2(s2, s3);
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void tapconv_x6_kernel(const spaa_tapconv_t p, const int m_tiles,
                                                            const int n_tiles) {
    constexpr int WAVES_N = BN / 32;
    constexpr int A_LD = BM * 8 / 256;
    static_assert((BM / 32) * (BN / 32) == 4, "4 waves, 32x32 outputs each");
    static_assert(A_LD >= 1, "BM >= 32");
    constexpr int PLANE = BM * ROWB;      // bytes of one bf16 plane of the A tile
    constexpr int STAGE = 3 * PLANE;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int4* s_taps = reinterpret_cast<int4*>(smem_raw);
    unsigned char* As = smem_raw + TAP_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const spaa_tapclass_t cl = p.cls[blockIdx.y];

    const int nwg = m_tiles * n_tiles;
    int tile;
    {
        const int orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int n_blk = (tile % n_tiles) * BN;
    const int m_blk = (tile / n_tiles) * BM;

    for (int i = tid; i <= cl.ntaps; i += 256) {
        if (i < cl.ntaps) {
            const int dy = p.taps[2 * (cl.tap_off + i)], dx = p.taps[2 * (cl.tap_off + i) + 1];
            s_taps[i] = make_int4(dy, dx, dy * p.Win + dx, 0);
        } else {
            s_taps[i] = make_int4(-(1 << 28), 0, 0, 0);
        }
    }

    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;
    const int kq = tid & 7;

    int a_iy[A_LD], a_ix[A_LD], a_pix[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m_blk + (tid >> 3) + 32 * i;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int b = mm / HWm;
        const int r = mm - b * HWm;
        const int y = r / p.Wm;
        const int x = r - y * p.Wm;
        a_iy[i] = ok ? y * p.s_in : -(1 << 28);
        a_ix[i] = x * p.s_in;
        a_pix[i] = (b * p.Hin + y * p.s_in) * p.Win + x * p.s_in;
    }
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)p.in_cstride * 4u;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    // split weights of this class: [3][Npad][Kpad] bf16
    const int npad = (p.Cout + 127) & ~127;
    const int plane_bytes = npad * cl.Kpad * 2;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_split) + (uint64_t)cl.w_off * 6u;
    const uint32_t w_lo = __builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = __builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane(3u * (uint32_t)plane_bytes),
                                                           0x00020000);

    const int wm0 = (wave / WAVES_N) * 32;
    const int wn0 = (wave % WAVES_N) * 32;
    // B fragment of this lane: output channel n, k = 8*(lane>>5) .. +7 of each 16-deep slice
    const int b_voff = ((n_blk + wn0 + (lane & 31)) * cl.Kpad + 8 * (lane >> 5)) * 2;
    // A fragment: row wm0 + (lane&31), 16 bytes at k = 8*(lane>>5)
    const int a_frag = (wm0 + (lane & 31)) * ROWB + (lane >> 5) * 16;

    const int Cin = p.Cin;
    const int adv_tap = BK / Cin, adv_c = BK - adv_tap * Cin;
    int k_tap = (4 * kq) / Cin;
    int k_c = 4 * kq - k_tap * Cin;

    f4 ra[A_LD];
    u4 rb_nxt[3][2], rb_cur[3][2];
    const int nk = cl.Kpad / BK;

    __syncthreads();

#define X6_LOAD_A()                                                                                       \
    {                                                                                                     \
        const int4 d = s_taps[min(k_tap, cl.ntaps)];                                                      \
        const int cbyte = (p.in_coff + k_c) * 4;                                                          \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                                \
            const int iy = a_iy[i] + d.x, ix = a_ix[i] + d.y;                                             \
            const bool v = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;              \
            const int off = (a_pix[i] + d.z) * (p.in_cstride * 4) + cbyte;                                \
            ra[i] = __builtin_bit_cast(                                                                   \
                f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, v ? off : (int)0x80000000, 0, 0));     \
        }                                                                                                 \
        k_tap += adv_tap;                                                                                 \
        k_c += adv_c;                                                                                     \
        if (k_c >= Cin) {                                                                                 \
            k_c -= Cin;                                                                                   \
            k_tap += 1;                                                                                   \
        }                                                                                                 \
    }
#define X6_LOAD_B(ks)                                                                                     \
    {                                                                                                     \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                  \
            _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                              \
                rb_nxt[pl][kk] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(            \
                    rsrc_w, b_voff + pl * plane_bytes, ((ks) * BK + kk * 16) * 2, 0));                    \
    }
#define X6_STORE_A(stage)                                                                                 \
    {                                                                                                     \
        unsigned char* dst = As + (stage) * STAGE + (tid >> 3) * ROWB + kq * 8;                           \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                                \
            u2 h, m, l;                                                                                   \
            split4(ra[i], h, m, l);                                                                       \
            *reinterpret_cast<u2*>(dst + 32 * i * ROWB) = h;                                              \
            *reinterpret_cast<u2*>(dst + 32 * i * ROWB + PLANE) = m;                                      \
            *reinterpret_cast<u2*>(dst + 32 * i * ROWB + 2 * PLANE) = l;                                  \
        }                                                                                                 \
    }

    f32x16 acc0, acc1, acc2;  // hh | hm + mh | hl + lh + mm
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = 0.f;

    if (nk > 0) {
        X6_LOAD_A()
        X6_LOAD_B(0)
        X6_STORE_A(0)
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) rb_cur[pl][kk] = rb_nxt[pl][kk];
    if (nk > 1) {
        X6_LOAD_A()
        X6_LOAD_B(1)
    }
    __syncthreads();

    for (int ks = 0; ks < nk; ++ks) {
        const int stage = ks & 1;
        if (ks + 1 < nk) X6_STORE_A(stage ^ 1)
        if (ks + 2 < nk) X6_LOAD_A()
        const unsigned char* as = As + stage * STAGE + a_frag;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(as + kk * 32);
            const bf16x8 am = *reinterpret_cast<const bf16x8*>(as + kk * 32 + PLANE);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(as + kk * 32 + 2 * PLANE);
            const bf16x8 bh = __builtin_bit_cast(bf16x8, rb_cur[0][kk]);
            const bf16x8 bm = __builtin_bit_cast(bf16x8, rb_cur[1][kk]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, rb_cur[2][kk]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);
        }
        if (ks + 1 < nk) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) rb_cur[pl][kk] = rb_nxt[pl][kk];
        }
        if (ks + 2 < nk) X6_LOAD_B(ks + 2)
        __syncthreads();
    }
#undef X6_LOAD_A
#undef X6_LOAD_B
#undef X6_STORE_A

    // ---- epilogue (same semantics as the fp32-MFMA kernel): col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool linear = (p.s_out == 1) && (cl.oy0 == 0) && (cl.ox0 == 0) && (p.Hm == p.Hout) && (p.Wm == p.Wout);
    const int n = n_blk + wn0 + (lane & 31);
    if (n >= p.Cout) return;
    const float bias = (p.bias != nullptr) ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m_blk + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        size_t o;
        if (linear) {
            o = (size_t)m;
        } else {
            const int b = m / HWm;
            const int rr = m - b * HWm;
            const int y = rr / p.Wm;
            const int x = rr - y * p.Wm;
            const int oy = cl.oy0 + y * p.s_out;
            const int ox = cl.ox0 + x * p.s_out;
            if (oy >= p.Hout || ox >= p.Wout) continue;
            o = ((size_t)b * p.Hout + oy) * p.Wout + ox;
        }
        float v = (acc0[r] + (acc1[r] + acc2[r])) + bias;
        if (p.add != nullptr) v += p.add[o * p.add_cstride + p.add_coff + n];
        if (p.act == SPAA_ACT_RELU) {
            v = fmaxf(v, 0.f);
        } else if (p.act == SPAA_ACT_RELU_CLAMP1) {
            v = fmaxf(v, 0.f);
            if (p.aux_out != nullptr) p.aux_out[o * p.out_cstride + p.out_coff + n] = v;
            v = fminf(v, 1.f);
        } else if (p.act == SPAA_ACT_LEAKY01) {
            v = v > 0.f ? v : 0.1f * v;
        }
        if (p.gate != nullptr) {
            const float g = p.gate[o * p.gate_cstride + p.gate_coff + n];
            const bool pass = (p.gate_mode == SPAA_GATE_POS_LE1) ? (g > 0.f && g <= 1.f) : (g > 0.f);
            v = pass ? v : 0.f;
        }
        p.out[o * p.out_cstride + p.out_coff + n] = v;
        if (p.gate2 != nullptr) {
            const float g2 = p.gate2[o * p.gate2_cstride + p.gate2_coff + n];
            p.aux_out[o * p.out_cstride + p.out_coff + n] = (g2 > 0.f) ? v : 0.f;
        }
    }
}

template <int BM, int BN>
int launch_x6(const spaa_tapconv_t& d, hipStream_t stream) {
    const int64_t M = (int64_t)d.B * d.Hm * d.Wm;
    const int m_tiles = (int)((M + BM - 1) / BM);
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const size_t smem = (size_t)TAP_BYTES + 2 * 3 * BM * ROWB;
    dim3 grid(m_tiles * n_tiles, d.nclass, 1);
    hipLaunchKernelGGL((tapconv_x6_kernel<BM, BN>), grid, dim3(256), smem, stream, d, m_tiles, n_tiles);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// v2: larger workgroup tile with BOTH operands staged through LDS.  The v1 kernel above is bound by the vector L1
// (64 B/clk/CU): per 384 matrix-core cycles a 64x64 workgroup moves 8 KB of activations + 24 KB of per-wave weight
// fragments.  Here a BM x BN workgroup (waves 2x2, each TMxTN 32x32 tiles) stages the weight planes once per
// workgroup: (128*BM + 192*BN) bytes per (BM*BN/4096)*384 cycles per SIMD -> 36 B/clk at 128x64.  One LDS stage
// (3 bf16 planes of A and B, 46 KB at 128x64) so that 2-3 workgroups share a CU; the next tile's global loads are in
// flight (registers) while the current one feeds the matrix cores; two barriers per K-step.
template <int BM, int BN, int NG>
__global__ __launch_bounds__(256, 2) void tapconv_x6v2_kernel(const spaa_tapconv_t p, const int m_tiles,
                                                              const int n_tiles) {
    constexpr int WM = BM / 2, WN = BN / 2;  // 2 x 2 waves
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM * 8 / 256;        // float4 (4 fp32 of one pixel) per thread per K-step
    constexpr int B_LD = BN * 4 / 256;        // 16-B pieces (8 bf16 of one output channel) per thread per plane
    static_assert(TM >= 1 && TN >= 1 && A_LD >= 1 && B_LD >= 1, "tile too small");
    constexpr int PLANE_A = BM * ROWB, PLANE_B = BN * ROWB;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int4* s_taps = reinterpret_cast<int4*>(smem_raw);
    unsigned char* As = smem_raw + TAP_BYTES;
    unsigned char* Bs = As + 3 * PLANE_A;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const spaa_tapclass_t cl = p.cls[blockIdx.y];

    const int nwg = m_tiles * n_tiles;
    int tile;
    {
        const int orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int n_blk = (tile % n_tiles) * BN;
    const int m_blk = (tile / n_tiles) * BM;

    for (int i = tid; i <= cl.ntaps; i += 256) {
        if (i < cl.ntaps) {
            const int dy = p.taps[2 * (cl.tap_off + i)], dx = p.taps[2 * (cl.tap_off + i) + 1];
            s_taps[i] = make_int4(dy, dx, dy * p.Win + dx, 0);
        } else {
            s_taps[i] = make_int4(-(1 << 28), 0, 0, 0);
        }
    }

    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;
    const int kq = tid & 7;

    int a_iy[A_LD], a_ix[A_LD], a_pix[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m_blk + (tid >> 3) + 32 * i;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int b = mm / HWm;
        const int r = mm - b * HWm;
        const int y = r / p.Wm;
        const int x = r - y * p.Wm;
        a_iy[i] = ok ? y * p.s_in : -(1 << 28);
        a_ix[i] = x * p.s_in;
        a_pix[i] = (b * p.Hin + y * p.s_in) * p.Win + x * p.s_in;
    }
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)p.in_cstride * 4u;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int npad = (p.Cout + 127) & ~127;
    const int plane_bytes = npad * cl.Kpad * 2;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_split) + (uint64_t)cl.w_off * 6u;
    const uint32_t w_lo = __builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = __builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane(3u * (uint32_t)plane_bytes),
                                                           0x00020000);
    // weight staging: thread -> (row = tid>>2 (+64 j), 16-byte piece q = tid&3) of each plane
    const int bq = tid & 3;
    int b_goff[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) b_goff[j] = ((n_blk + (tid >> 2) + 64 * j) * cl.Kpad + 8 * bq) * 2;

    const int wm0 = (wave >> 1) * WM;
    const int wn0 = (wave & 1) * WN;
    const int a_frag = (wm0 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    const int b_frag = (wn0 + (lane & 31)) * ROWB + (lane >> 5) * 16;

    const int Cin = p.Cin;
    const int adv_tap = BK / Cin, adv_c = BK - adv_tap * Cin;
    int k_tap = (4 * kq) / Cin;
    int k_c = 4 * kq - k_tap * Cin;

    f4 ra[A_LD];
    u4 rb[3][B_LD];
    const int nk = cl.Kpad / BK;

    f32x16 acc[NG][TM][TN];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][i][j][r] = 0.f;

#define V2_LOAD(ks)                                                                                       \
    {                                                                                                     \
        const int4 d = s_taps[min(k_tap, cl.ntaps)];                                                      \
        const int cbyte = (p.in_coff + k_c) * 4;                                                          \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                                \
            const int iy = a_iy[i] + d.x, ix = a_ix[i] + d.y;                                             \
            const bool v = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;              \
            const int off = (a_pix[i] + d.z) * (p.in_cstride * 4) + cbyte;                                \
            ra[i] = __builtin_bit_cast(                                                                   \
                f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, v ? off : (int)0x80000000, 0, 0));     \
        }                                                                                                 \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                  \
            _Pragma("unroll") for (int j = 0; j < B_LD; ++j)                                              \
                rb[pl][j] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(                 \
                    rsrc_w, b_goff[j] + pl * plane_bytes, (ks) * (BK * 2), 0));                           \
        k_tap += adv_tap;                                                                                 \
        k_c += adv_c;                                                                                     \
        if (k_c >= Cin) {                                                                                 \
            k_c -= Cin;                                                                                   \
            k_tap += 1;                                                                                   \
        }                                                                                                 \
    }

    __syncthreads();  // taps visible
    if (nk > 0) V2_LOAD(0)

    for (int ks = 0; ks < nk; ++ks) {
        __syncthreads();  // every wave has finished reading the previous tile
        {
            unsigned char* dst = As + (tid >> 3) * ROWB + kq * 8;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                u2 h, m, l;
                split4(ra[i], h, m, l);
                *reinterpret_cast<u2*>(dst + 32 * i * ROWB) = h;
                *reinterpret_cast<u2*>(dst + 32 * i * ROWB + PLANE_A) = m;
                *reinterpret_cast<u2*>(dst + 32 * i * ROWB + 2 * PLANE_A) = l;
            }
            unsigned char* dstb = Bs + (tid >> 2) * ROWB + bq * 16;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < B_LD; ++j) *reinterpret_cast<u4*>(dstb + pl * PLANE_B + 64 * j * ROWB) = rb[pl][j];
        }
        __syncthreads();
        if (ks + 1 < nk) V2_LOAD(ks + 1)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[3][TM], b[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[pl][i] = *reinterpret_cast<const bf16x8*>(As + pl * PLANE_A + a_frag + i * 32 * ROWB + kk * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[pl][j] = *reinterpret_cast<const bf16x8*>(Bs + pl * PLANE_B + b_frag + j * 32 * ROWB + kk * 32);
            }
            // weights are the A operand (rows = output channels), pixels the B operand (columns): a lane then owns ONE
            // pixel and 16 output channels in groups of 4 consecutive ones -> float4 epilogue.  Product-major order:
            // consecutive MFMAs go to different accumulators (no back-to-back dependent issue).
            constexpr int G1 = NG > 1 ? 1 : 0, G2 = NG > 2 ? 2 : G1;
            constexpr int PW[6] = {0, 1, 0, 2, 0, 1}, PX[6] = {0, 0, 1, 0, 2, 1}, PG[6] = {0, G1, G1, G2, G2, G2};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[PG[q]][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PW[q]][j], a[PX[q]][i],
                                                                                  acc[PG[q]][i][j], 0, 0, 0);
        }
    }
#undef V2_LOAD

    // ---- epilogue.  D layout of the 32x32 tile: column (lane & 31) = pixel, row (r&3) + 8*(r>>2) + 4*(lane>>5) =
    // output channel: registers 4g..4g+3 of a lane are 4 consecutive channels of its pixel.
    const bool linear = (p.s_out == 1) && (cl.oy0 == 0) && (cl.ox0 == 0) && (p.Hm == p.Hout) && (p.Wm == p.Wout);
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    if (fast_epi_ok(p, vec)) {   // branch-free operand accesses, a pixel's four channel quads in flight (epilogue.hpp: fast_epi_*)
        const fast_epi_t fe = make_fast_epi(p, 0);
#define V2_FAST(T)                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                       \
        size_t o = 0;                                                                                                      \
        const bool okp = out_pixel(p, cl, m_blk + wm0 + 32 * i + (lane & 31), M, HWm, o);                                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                                   \
            fast_pre_t<T> pre[4];                                                                                          \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
                const int n0 = n_blk + wn0 + 32 * j + 8 * g + 4 * (lane >> 5);                                             \
                pre[g] = fast_epi_load<T, true>(fe, p, (int)o, n0, okp && n0 < p.Cout);                                    \
            }                                                                                                              \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
                const int n0 = n_blk + wn0 + 32 * j + 8 * g + 4 * (lane >> 5);                                             \
                float v[4];                                                                                                \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                            \
                    const int r = 4 * g + e;                                                                               \
                    float t = acc[0][i][j][r];                                                                             \
                    if (NG == 2) t += acc[NG > 1 ? 1 : 0][i][j][r];                                                        \
                    if (NG == 3) t += acc[NG > 1 ? 1 : 0][i][j][r] + acc[NG > 2 ? 2 : 0][i][j][r];                         \
                    v[e] = t;                                                                                              \
                }                                                                                                          \
                fast_epi_store<T, float[4], true>(fe, p, (int)o, n0, okp && n0 < p.Cout, v, pre[g]);                       \
            }                                                                                                              \
        }                                                                                                                  \
    }
        if (p.io_dtype & SPAA_IO_OUT_F16) V2_FAST(_Float16) else V2_FAST(float)
#undef V2_FAST
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_blk + wm0 + 32 * i + (lane & 31);
        if (m >= M) continue;
        size_t o;
        if (linear) {
            o = (size_t)m;
        } else {
            const int b = m / HWm;
            const int rr = m - b * HWm;
            const int y = rr / p.Wm;
            const int x = rr - y * p.Wm;
            const int oy = cl.oy0 + y * p.s_out;
            const int ox = cl.ox0 + x * p.s_out;
            if (oy >= p.Hout || ox >= p.Wout) continue;
            o = ((size_t)b * p.Hout + oy) * p.Wout + ox;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = n_blk + wn0 + 32 * j + 8 * g + 4 * (lane >> 5);
                if (n0 >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    float t = acc[0][i][j][r];
                    if (NG == 2) t += acc[NG > 1 ? 1 : 0][i][j][r];
                    if (NG == 3) t += acc[NG > 1 ? 1 : 0][i][j][r] + acc[NG > 2 ? 2 : 0][i][j][r];
                    v[e] = t;
                }
                store4(p, o, n0, v, vec);  // bias + residual + activation + gates (epilogue.hpp)
            }
        }
    }
}

template <int BM, int BN, int NG>
int launch_x6v2(const spaa_tapconv_t& d, hipStream_t stream) {
    const int64_t M = (int64_t)d.B * d.Hm * d.Wm;
    const int m_tiles = (int)((M + BM - 1) / BM);
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const size_t smem = (size_t)TAP_BYTES + 3 * (BM + BN) * ROWB;
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&tapconv_x6v2_kernel<BM, BN, NG>), (int)smem, attr_set);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(m_tiles * n_tiles, d.nclass, 1);
    hipLaunchKernelGGL((tapconv_x6v2_kernel<BM, BN, NG>), grid, dim3(256), smem, stream, d, m_tiles, n_tiles);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// v3: as v2 but the LDS image is unpadded + XOR-swizzled (64-B rows), which makes room for TWO stages: one barrier per
// K-step, and the split / LDS store of the next tile overlaps the MFMAs of the current one.
template <int BM, int BN, int NG>
__global__ __launch_bounds__(256, 2) void tapconv_x6v3_kernel(const spaa_tapconv_t p, const int m_tiles,
                                                              const int n_tiles) {
    constexpr int WM = BM / 2, WN = BN / 2;  // 2 x 2 waves
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_LD = BM * 8 / 256;        // float4 (4 fp32 of one pixel) per thread per K-step
    constexpr int B_LD = BN * 4 / 256;        // 16-B pieces (8 bf16 of one output channel) per thread per plane
    static_assert(TM >= 1 && TN >= 1 && A_LD >= 1 && B_LD >= 1, "tile too small");
    constexpr int ROW = 64;  // unpadded 32 bf16; 16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3)
    constexpr int PLANE_A = BM * ROW, PLANE_B = BN * ROW;
    constexpr int STAGE = 3 * (PLANE_A + PLANE_B);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int4* s_taps = reinterpret_cast<int4*>(smem_raw);
    unsigned char* As = smem_raw + TAP_BYTES;
    unsigned char* Bs = As + 3 * PLANE_A;  // within a stage

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const spaa_tapclass_t cl = p.cls[blockIdx.y];

    const int nwg = m_tiles * n_tiles;
    int tile;
    {
        const int orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int n_blk = (tile % n_tiles) * BN;
    const int m_blk = (tile / n_tiles) * BM;

    for (int i = tid; i <= cl.ntaps; i += 256) {
        if (i < cl.ntaps) {
            const int dy = p.taps[2 * (cl.tap_off + i)], dx = p.taps[2 * (cl.tap_off + i) + 1];
            s_taps[i] = make_int4(dy, dx, dy * p.Win + dx, 0);
        } else {
            s_taps[i] = make_int4(-(1 << 28), 0, 0, 0);
        }
    }

    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;
    const int kq = tid & 7;

    int a_iy[A_LD], a_ix[A_LD], a_pix[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int m = m_blk + (tid >> 3) + 32 * i;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int b = mm / HWm;
        const int r = mm - b * HWm;
        const int y = r / p.Wm;
        const int x = r - y * p.Wm;
        a_iy[i] = ok ? y * p.s_in : -(1 << 28);
        a_ix[i] = x * p.s_in;
        a_pix[i] = (b * p.Hin + y * p.s_in) * p.Win + x * p.s_in;
    }
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)p.in_cstride * 4u;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int npad = (p.Cout + 127) & ~127;
    const int plane_bytes = npad * cl.Kpad * 2;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_split) + (uint64_t)cl.w_off * 6u;
    const uint32_t w_lo = __builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = __builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane(3u * (uint32_t)plane_bytes),
                                                           0x00020000);
    // weight staging: thread -> (row = tid>>2 (+64 j), 16-byte piece q = tid&3) of each plane
    const int bq = tid & 3;
    int b_goff[B_LD];
#pragma unroll
    for (int j = 0; j < B_LD; ++j) b_goff[j] = ((n_blk + (tid >> 2) + 64 * j) * cl.Kpad + 8 * bq) * 2;

    const int wm0 = (wave >> 1) * WM;
    const int wn0 = (wave & 1) * WN;
    // fragment byte offsets for kk = 0 / 1 (chunk index kk*2 + (lane>>5), swizzled by the row key)
    const int a_row = wm0 + (lane & 31), b_row = wn0 + (lane & 31);
    const int a_frag0 = a_row * ROW + ((((lane >> 5)) ^ ((a_row >> 2) & 3)) << 4);
    const int a_frag1 = a_row * ROW + (((2 + (lane >> 5)) ^ ((a_row >> 2) & 3)) << 4);
    const int b_frag0 = b_row * ROW + ((((lane >> 5)) ^ ((b_row >> 2) & 3)) << 4);
    const int b_frag1 = b_row * ROW + (((2 + (lane >> 5)) ^ ((b_row >> 2) & 3)) << 4);

    const int Cin = p.Cin;
    const int adv_tap = BK / Cin, adv_c = BK - adv_tap * Cin;
    int k_tap = (4 * kq) / Cin;
    int k_c = 4 * kq - k_tap * Cin;

    f4 ra[A_LD];
    u4 rb[3][B_LD];
    const int nk = cl.Kpad / BK;

    f32x16 acc[NG][TM][TN];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][i][j][r] = 0.f;

#define V2_LOAD(ks)                                                                                       \
    {                                                                                                     \
        const int4 d = s_taps[min(k_tap, cl.ntaps)];                                                      \
        const int cbyte = (p.in_coff + k_c) * 4;                                                          \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                                \
            const int iy = a_iy[i] + d.x, ix = a_ix[i] + d.y;                                             \
            const bool v = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;              \
            const int off = (a_pix[i] + d.z) * (p.in_cstride * 4) + cbyte;                                \
            ra[i] = __builtin_bit_cast(                                                                   \
                f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, v ? off : (int)0x80000000, 0, 0));     \
        }                                                                                                 \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                                  \
            _Pragma("unroll") for (int j = 0; j < B_LD; ++j)                                              \
                rb[pl][j] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(                 \
                    rsrc_w, b_goff[j] + pl * plane_bytes, (ks) * (BK * 2), 0));                           \
        k_tap += adv_tap;                                                                                 \
        k_c += adv_c;                                                                                     \
        if (k_c >= Cin) {                                                                                 \
            k_c -= Cin;                                                                                   \
            k_tap += 1;                                                                                   \
        }                                                                                                 \
    }

#define V3_STORE(stage)                                                                                   \
    {                                                                                                     \
        unsigned char* sa = As + (stage) * STAGE;                                                         \
        unsigned char* sb = sa + 3 * PLANE_A;                                                             \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                                \
            const int row = (tid >> 3) + 32 * i;                                                          \
            unsigned char* dst = sa + row * ROW + ((((kq >> 1)) ^ ((row >> 2) & 3)) << 4) + (kq & 1) * 8; \
            u2 h, m, l;                                                                                   \
            split4(ra[i], h, m, l);                                                                       \
            *reinterpret_cast<u2*>(dst) = h;                                                              \
            *reinterpret_cast<u2*>(dst + PLANE_A) = m;                                                    \
            *reinterpret_cast<u2*>(dst + 2 * PLANE_A) = l;                                                \
        }                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < B_LD; ++j) {                                                \
            const int row = (tid >> 2) + 64 * j;                                                          \
            unsigned char* dst = sb + row * ROW + ((bq ^ ((row >> 2) & 3)) << 4);                         \
            _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                              \
                *reinterpret_cast<u4*>(dst + pl * PLANE_B) = rb[pl][j];                                   \
        }                                                                                                 \
    }

    __syncthreads();  // taps visible
    if (nk > 0) {
        V2_LOAD(0)
        V3_STORE(0)
    }
    if (nk > 1) V2_LOAD(1)
    __syncthreads();

    // step ks: tile ks is in LDS stage ks&1, the registers hold tile ks+1.  Its split + LDS store (into the other
    // stage, last read in step ks-1) and the global loads of tile ks+2 are issued before the MFMAs of step ks.
    for (int ks = 0; ks < nk; ++ks) {
        const int stage = ks & 1;
        if (ks + 1 < nk) V3_STORE(stage ^ 1)
        if (ks + 2 < nk) V2_LOAD(ks + 2)
        const unsigned char* sa = As + stage * STAGE;
        const unsigned char* sb = sa + 3 * PLANE_A;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[3][TM], b[3][TN];
            const int af = kk ? a_frag1 : a_frag0, bfo = kk ? b_frag1 : b_frag0;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[pl][i] = *reinterpret_cast<const bf16x8*>(sa + pl * PLANE_A + af + i * 32 * ROW);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[pl][j] = *reinterpret_cast<const bf16x8*>(sb + pl * PLANE_B + bfo + j * 32 * ROW);
            }
            constexpr int G1 = NG > 1 ? 1 : 0, G2 = NG > 2 ? 2 : G1;
            constexpr int PW[6] = {0, 1, 0, 2, 0, 1}, PX[6] = {0, 0, 1, 0, 2, 1}, PG[6] = {0, G1, G1, G2, G2, G2};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[PG[q]][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[PW[q]][j], a[PX[q]][i],
                                                                                  acc[PG[q]][i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#undef V3_STORE
#undef V2_LOAD

    // ---- epilogue.  D layout of the 32x32 tile: column (lane & 31) = pixel, row (r&3) + 8*(r>>2) + 4*(lane>>5) =
    // output channel: registers 4g..4g+3 of a lane are 4 consecutive channels of its pixel.
    const bool linear = (p.s_out == 1) && (cl.oy0 == 0) && (cl.ox0 == 0) && (p.Hm == p.Hout) && (p.Wm == p.Wout);
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    if (fast_epi_ok(p, vec)) {   // branch-free operand accesses, a pixel's four channel quads in flight (epilogue.hpp: fast_epi_*)
        const fast_epi_t fe = make_fast_epi(p, 0);
#define V2_FAST(T)                                                                                                         \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                       \
        size_t o = 0;                                                                                                      \
        const bool okp = out_pixel(p, cl, m_blk + wm0 + 32 * i + (lane & 31), M, HWm, o);                                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                                   \
            fast_pre_t<T> pre[4];                                                                                          \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
                const int n0 = n_blk + wn0 + 32 * j + 8 * g + 4 * (lane >> 5);                                             \
                pre[g] = fast_epi_load<T, true>(fe, p, (int)o, n0, okp && n0 < p.Cout);                                    \
            }                                                                                                              \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                \
                const int n0 = n_blk + wn0 + 32 * j + 8 * g + 4 * (lane >> 5);                                             \
                float v[4];                                                                                                \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                            \
                    const int r = 4 * g + e;                                                                               \
                    float t = acc[0][i][j][r];                                                                             \
                    if (NG == 2) t += acc[NG > 1 ? 1 : 0][i][j][r];                                                        \
                    if (NG == 3) t += acc[NG > 1 ? 1 : 0][i][j][r] + acc[NG > 2 ? 2 : 0][i][j][r];                         \
                    v[e] = t;                                                                                              \
                }                                                                                                          \
                fast_epi_store<T, float[4], true>(fe, p, (int)o, n0, okp && n0 < p.Cout, v, pre[g]);                       \
            }                                                                                                              \
        }                                                                                                                  \
    }
        if (p.io_dtype & SPAA_IO_OUT_F16) V2_FAST(_Float16) else V2_FAST(float)
#undef V2_FAST
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_blk + wm0 + 32 * i + (lane & 31);
        if (m >= M) continue;
        size_t o;
        if (linear) {
            o = (size_t)m;
        } else {
            const int b = m / HWm;
            const int rr = m - b * HWm;
            const int y = rr / p.Wm;
            const int x = rr - y * p.Wm;
            const int oy = cl.oy0 + y * p.s_out;
            const int ox = cl.ox0 + x * p.s_out;
            if (oy >= p.Hout || ox >= p.Wout) continue;
            o = ((size_t)b * p.Hout + oy) * p.Wout + ox;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = n_blk + wn0 + 32 * j + 8 * g + 4 * (lane >> 5);
                if (n0 >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    float t = acc[0][i][j][r];
                    if (NG == 2) t += acc[NG > 1 ? 1 : 0][i][j][r];
                    if (NG == 3) t += acc[NG > 1 ? 1 : 0][i][j][r] + acc[NG > 2 ? 2 : 0][i][j][r];
                    v[e] = t;
                }
                store4(p, o, n0, v, vec);  // bias + residual + activation + gates (epilogue.hpp)
            }
        }
    }
}


template <int BM, int BN, int NG>
int launch_x6v3(const spaa_tapconv_t& d, hipStream_t stream) {
    const int64_t M = (int64_t)d.B * d.Hm * d.Wm;
    const int m_tiles = (int)((M + BM - 1) / BM);
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const size_t smem = (size_t)TAP_BYTES + 2 * 3 * (BM + BN) * 64;
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&tapconv_x6v3_kernel<BM, BN, NG>), (int)smem, attr_set);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(m_tiles * n_tiles, d.nclass, 1);
    hipLaunchKernelGGL((tapconv_x6v3_kernel<BM, BN, NG>), grid, dim3(256), smem, stream, d, m_tiles, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tiles 12..14 after the common shape checks
int spaa_launch_tapconv_x6(const spaa_tapconv_t& d, int tile, hipStream_t stream) {
    if (d.w_split == nullptr) return hipErrorInvalidValue;
    for (int c = 0; c < d.nclass; ++c)
        if ((int64_t)((d.Cout + 127) & ~127) * d.cls[c].Kpad * 6 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    switch (tile) {
        case 12: return launch_x6<64, 64>(d, stream);
        case 13: return launch_x6<128, 32>(d, stream);
        case 14: return launch_x6<32, 128>(d, stream);
        case 15: return launch_x6v2<128, 64, 3>(d, stream);
        case 16: return launch_x6v2<128, 64, 2>(d, stream);
        case 17: return launch_x6v2<128, 128, 1>(d, stream);
        case 18: return launch_x6v2<64, 64, 3>(d, stream);
        case 19: return launch_x6v2<64, 128, 2>(d, stream);
        case 20: return launch_x6v3<128, 64, 3>(d, stream);
        case 21: return launch_x6v3<128, 64, 2>(d, stream);
        case 22: return launch_x6v3<64, 64, 3>(d, stream);
        case 23: return launch_x6v3<128, 128, 1>(d, stream);
        case 24: return launch_x6v3<64, 128, 2>(d, stream);
        default: return hipErrorInvalidValue;
    }
}
