// tapconv.hip — generic tap-list convolution as implicit GEMM on the fp32 matrix cores of gfx950.
//
// GEMM view:  M = B*Hm*Wm output pixels of one parity class, N = Cout, K = ntaps*Cin.
//   A[m][k] = in[b, s_in*y + dy_t, s_in*x + dx_t, c]   (gathered on the fly, zero outside the image)
//   B[k][n] = W[n][k]                                   (pre-packed, K contiguous)
// One workgroup = 4 waves (one per SIMD) computes a BM x BN tile; each wave owns a WM x WN sub-tile built from
// 32x32 accumulators of v_mfma_f32_32x32x2_f32 (exact fp32: bitwise an fmaf chain, so results match the
// reference's fp32 CPU path to rounding).  K is consumed in steps of BK = 32: global -> registers (prefetch of the
// next step issued before the MFMAs of the current one) -> LDS (row stride 36 floats: ds_write_b128 /
// ds_read_b128 conflict-free) -> ds_read_b128 fragments.  fp32 MFMA is 64 cycles per instruction per SIMD, so one
// K-step is >= 2048 MFMA-cycles per wave and hides the HBM/L2 latency of the prefetch.
//
// Replaces: aten::convolution / aten::convolution_backward(input) as dispatched by the reference at
// models.py:284-301 and their autograd (projector_based_attack.py:302,310), and torchvision's convs
// (classifier.py:60).
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"

int spaa_launch_tapconv_x6(const spaa_tapconv_t& d, int tile, hipStream_t stream);   // tapconv_x6.hip
int spaa_launch_tapconv_x6d(const spaa_tapconv_t& d, int tile, hipStream_t stream);  // tapconv_x6d.hip
int spaa_launch_tapconv_h16(const spaa_tapconv_t& d, int tile, hipStream_t stream);  // tapconv_h16.hip
int spaa_launch_tapconv_h16p(const spaa_tapconv_t& d, hipStream_t stream);           // tapconv_h16p.hip
int spaa_launch_tapconv_thinmf(const spaa_tapconv_t& d, hipStream_t stream);         // tapconv_thinmf.hip
int spaa_launch_tapconv_wino(const spaa_tapconv_t& d, hipStream_t stream);          // tapconv_wino.hip
int spaa_launch_tapconv_x6p(const spaa_tapconv_t& d, hipStream_t stream);           // tapconv_x6p.hip
int spaa_launch_tapconv_c3(const spaa_tapconv_t& d, hipStream_t stream);            // tapconv_c3.hip
int spaa_launch_thinpatch(const spaa_tapconv_t& d, hipStream_t stream);               // thinpatch.hip
int spaa_launch_smallcin(const spaa_tapconv_t& d, hipStream_t stream);                // smallcin.hip

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LDK = 36;           // LDS row stride in floats (BK + 4): 144 B rows, 16-B aligned
constexpr int TAP_SMEM_FLOATS = 4 * (SPAA_MAX_TAPS + 4);

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void tapconv_kernel(const spaa_tapconv_t p, const int m_tiles,
                                                         const int n_tiles) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int A_LD = BM * 8 / 256;
    constexpr int B_LD = BN * 8 / 256;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
    static_assert(B_LD >= 1, "BN >= 32");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    int4* s_taps = reinterpret_cast<int4*>(smem);  // (dy, dx, dy*Win + dx, 0) per tap; entry ntaps = invalid
    float* As = smem + TAP_SMEM_FLOATS;
    float* Bs = As + 2 * BM * LDK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const spaa_tapclass_t cl = p.cls[blockIdx.y];

    // XCD-aware tile order: workgroups that land on one XCD (blockIdx.x % 8) walk a contiguous range of tiles,
    // N-tiles of one M-tile first, so im2col rows / halos are re-read from that XCD's L2.
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
            s_taps[i] = make_int4(-(1 << 28), 0, 0, 0);  // k >= K: always out of bounds
        }
    }

    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;
    const int kq = tid & 7;

    // per-thread im2col rows: input coordinates of the centre tap and its pixel index
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
    // hardware-bounds-checked buffer loads: an out-of-image tap gets an offset past num_records and reads 0
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)p.in_cstride * 4u;
    // descriptor inputs forced wave-uniform (readfirstlane) so hipcc keeps the SRD in SGPRs instead of wrapping
    // every buffer_load in a waterfall loop
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    float* in_uniform = reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo);
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(
        in_uniform, 0, (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int npad = (p.Cout + 127) & ~127;
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.weights + cl.w_off), 0,
                                                           npad * cl.Kpad * 4, 0x00020000);
    int w_off[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; ++i) w_off[i] = ((n_blk + (tid >> 3) + 32 * i) * cl.Kpad + 4 * kq) * 4;

    // (tap, channel) of this thread's 4-wide K slice, advanced incrementally by BK per K-step (no division)
    const int Cin = p.Cin;
    const int adv_tap = BK / Cin, adv_c = BK - adv_tap * Cin;
    int k_tap = (4 * kq) / Cin;
    int k_c = 4 * kq - k_tap * Cin;

    f4 ra[A_LD], rb[B_LD];
    const int nk = cl.Kpad / BK;

    __syncthreads();  // taps visible

#define TAPCONV_LOAD_TILE(ks)                                                                             \
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
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i) {                                                \
            rb[i] = __builtin_bit_cast(                                                                   \
                f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_off[i], (ks) * (BK * 4), 0));         \
        }                                                                                                 \
        k_tap += adv_tap;                                                                                 \
        k_c += adv_c;                                                                                     \
        if (k_c >= Cin) {                                                                                 \
            k_c -= Cin;                                                                                   \
            k_tap += 1;                                                                                   \
        }                                                                                                 \
    }
#define TAPCONV_STORE_TILE(stage)                                                                         \
    {                                                                                                     \
        float* as_ = As + (stage) * BM * LDK + (tid >> 3) * LDK + 4 * kq;                                 \
        float* bs_ = Bs + (stage) * BN * LDK + (tid >> 3) * LDK + 4 * kq;                                 \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f4*>(as_ + 32 * i * LDK) = ra[i]; \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i) *reinterpret_cast<f4*>(bs_ + 32 * i * LDK) = rb[i]; \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;
    const int frag_off = (lane & 31) * LDK + 4 * (lane >> 5);

    if (nk > 0) {
        TAPCONV_LOAD_TILE(0)
        TAPCONV_STORE_TILE(0)
    }
    if (nk > 1) TAPCONV_LOAD_TILE(1)
    __syncthreads();

    // Step ks: tile ks is in LDS stage ks&1; the registers hold tile ks+1 (loads issued one step ago).  Its LDS store
    // and the global loads of tile ks+2 are issued BEFORE the MFMAs of step ks, so their latency hides under the matrix
    // work and the barrier at the end of the step finds every wave's writes long done.  (Stage (ks+1)&1 was last read
    // in step ks-1, which every wave has left through the previous barrier.)
    for (int ks = 0; ks < nk; ++ks) {
        const int stage = ks & 1;
        if (ks + 1 < nk) TAPCONV_STORE_TILE(stage ^ 1)
        if (ks + 2 < nk) TAPCONV_LOAD_TILE(ks + 2)
        const float* as = As + stage * BM * LDK + wm0 * LDK + frag_off;
        const float* bs = Bs + stage * BN * LDK + wn0 * LDK + frag_off;
        f4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f4*>(as + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f4*>(bs + j * 32 * LDK);
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            const int cur = kb & 1, nxt = cur ^ 1;
            if (kb + 1 < BK / 8) {  // prefetch the next 8-deep fragment pair while this one feeds the matrix core
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[nxt][i] = *reinterpret_cast<const f4*>(as + i * 32 * LDK + (kb + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[nxt][j] = *reinterpret_cast<const f4*>(bs + j * 32 * LDK + (kb + 1) * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][e], bf[cur][j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#undef TAPCONV_LOAD_TILE
#undef TAPCONV_STORE_TILE

    // ---- epilogue: C/D map of the 32x32 accumulator: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool linear = (p.s_out == 1) && (cl.oy0 == 0) && (cl.ox0 == 0) && (p.Hm == p.Hout) && (p.Wm == p.Wout);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n_blk + wn0 + 32 * j + (lane & 31);
        const bool n_ok = n < p.Cout;
        const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_blk + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= M || !n_ok) continue;
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
                float v = acc[i][j][r] + bias;
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
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Direct (VALU) variant of the same tap-list convolution for THIN layers, where a 32-wide MFMA tile would be >= 90 %
// padding: few output channels (conv6 32->3, the input-gradients of conv1 / conv1_s / the ResNet stem: N = 3) or few
// input channels (input-gradient of conv6, conv1, conv1_s: K <= 72).  One lane = one output pixel with all NOUT
// accumulators in registers; the weights of a (tap, channel-quad) are wave-uniform and come through the scalar cache
// (s_load), the im2col row through bounds-checked buffer loads.  HBM/L1-bound instead of MFMA-bound.
template <int NOUT>
__global__ __launch_bounds__(256) void directconv_kernel(const spaa_tapconv_t p) {
    const spaa_tapclass_t cl = p.cls[blockIdx.y];
    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;
    const int m = blockIdx.x * 256 + threadIdx.x;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int b = mm / HWm;
    const int r = mm - b * HWm;
    const int y = r / p.Wm;
    const int x = r - y * p.Wm;

    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)p.in_cstride * 4u;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const float* __restrict__ W = p.weights + cl.w_off;  // [Npad][Kpad], K contiguous
    const int Kpad = cl.Kpad;
    const int Cin = p.Cin;

    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;

    // few output channels: K is long -> keep several channel-quads in flight per tap;
    // few input channels: a tap is one or two quads -> keep several taps in flight
#pragma unroll(NOUT <= 4 ? 1 : 3)
    for (int t = 0; t < cl.ntaps; ++t) {
        const int dy = p.taps[2 * (cl.tap_off + t)], dx = p.taps[2 * (cl.tap_off + t) + 1];
        const int iy = y * p.s_in + dy, ix = x * p.s_in + dx;
        const bool v = ok && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
        const int off = (((b * p.Hin + iy) * p.Win + ix) * p.in_cstride + p.in_coff) * 4;
        const int voff = v ? off : (int)0x80000000;
        const float* __restrict__ wt = W + t * Cin;
#pragma unroll(NOUT <= 4 ? 8 : 1)
        for (int c = 0; c < Cin; c += 4) {
            const f4 a = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, voff + c * 4, 0, 0));
#pragma unroll
            for (int n = 0; n < NOUT; ++n) {
                const f4 w = *reinterpret_cast<const f4*>(wt + (size_t)n * Kpad + c);
                acc[n] = fmaf(a.x, w.x, acc[n]);
                acc[n] = fmaf(a.y, w.y, acc[n]);
                acc[n] = fmaf(a.z, w.z, acc[n]);
                acc[n] = fmaf(a.w, w.w, acc[n]);
            }
        }
    }
    if (!ok) return;
    const int oy = cl.oy0 + y * p.s_out, ox = cl.ox0 + x * p.s_out;
    if (oy >= p.Hout || ox >= p.Wout) return;
    const size_t o = ((size_t)b * p.Hout + oy) * p.Wout + ox;
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
        if (n >= p.Cout) break;
        float v = acc[n] + (p.bias != nullptr ? p.bias[n] : 0.f);
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

template <int NOUT>
int launch_direct(const spaa_tapconv_t& d, hipStream_t stream) {
    if (d.Cout > NOUT) return hipErrorInvalidValue;
    const int64_t M = (int64_t)d.B * d.Hm * d.Wm;
    dim3 grid((unsigned)((M + 255) / 256), d.nclass, 1);
    hipLaunchKernelGGL((directconv_kernel<NOUT>), grid, dim3(256), 0, stream, d);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Coalesced thin-N variant (Cout <= 4, Cin a power of two in 4..256): conv6 (32->3), the input-gradients of conv1 /
// conv1_s (32->3) and of the ResNet stem (64->3).  The L = Cin/4 lanes of a pixel each own one channel quad, so a
// wave reads whole 128-B lines (64/L neighbouring pixels per load instruction); every lane carries P pixels so that a
// tap's weights (three ds_read_b128 from LDS, broadcast across the pixels of the wave) feed P*12 FMAs; the L partial
// sums of a pixel are combined with xor-shuffles.  HBM/L1-bound: reads the input once, writes 16 B per pixel.
template <int P>
__global__ __launch_bounds__(256) void thinconv_kernel(const spaa_tapconv_t p) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];  // [ntaps][4][Cin] weights, then int2 taps
    const spaa_tapclass_t cl = p.cls[blockIdx.y];
    const int Cin = p.Cin;
    const int L = Cin >> 2;
    const int tid = threadIdx.x;
    const float* __restrict__ W = p.weights + cl.w_off;
    for (int i = tid; i < cl.ntaps * 4 * Cin; i += 256) {
        const int c = i % Cin;
        const int r = i / Cin;
        s_w[i] = W[(size_t)(r & 3) * cl.Kpad + (r >> 2) * Cin + c];
    }
    int2* s_t = reinterpret_cast<int2*>(s_w + cl.ntaps * 4 * Cin);
    for (int i = tid; i < cl.ntaps; i += 256)
        s_t[i] = make_int2(p.taps[2 * (cl.tap_off + i)], p.taps[2 * (cl.tap_off + i) + 1]);
    __syncthreads();

    const int cq = tid & (L - 1);
    const int ps = tid / L;
    const int PB = 256 / L;
    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;

    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)p.in_cstride * 4u;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = __builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = __builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    // XCD-aware order: the workgroups of one XCD (blockIdx.x % 8) take a contiguous range of pixel blocks, so the rows
    // above / below a block (the other taps) are served by that XCD's L2 instead of being fetched 3x from HBM
    int blk;
    {
        const int nwg = gridDim.x, orig = blockIdx.x;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        blk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int iy0[P], ix0[P], pix0[P], mm[P];
    float acc[P][4];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int m = (blk * P + j) * PB + ps;
        mm[j] = m;
        const bool ok = m < M;
        const int q = ok ? m : 0;
        const int b = q / HWm;
        const int r = q - b * HWm;
        const int y = r / p.Wm;
        const int x = r - y * p.Wm;
        iy0[j] = ok ? y * p.s_in : -(1 << 28);
        ix0[j] = x * p.s_in;
        pix0[j] = (b * p.Hin + y * p.s_in) * p.Win + x * p.s_in;
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[j][n] = 0.f;
    }
    const int cbyte = (p.in_coff + 4 * cq) * 4;
    for (int t = 0; t < cl.ntaps; ++t) {
        const int2 d = s_t[t];
        const int doff = d.x * p.Win + d.y;
        const float* wp = s_w + (size_t)t * 4 * Cin + 4 * cq;
        const f4 w0 = *reinterpret_cast<const f4*>(wp);
        const f4 w1 = *reinterpret_cast<const f4*>(wp + Cin);
        const f4 w2 = *reinterpret_cast<const f4*>(wp + 2 * Cin);
        const f4 w3 = *reinterpret_cast<const f4*>(wp + 3 * Cin);
        f4 a[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int iy = iy0[j] + d.x, ix = ix0[j] + d.y;
            const bool v = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const int off = (pix0[j] + doff) * (p.in_cstride * 4) + cbyte;
            a[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, v ? off : (int)0x80000000, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
            acc[j][0] = fmaf(a[j].x, w0.x, fmaf(a[j].y, w0.y, fmaf(a[j].z, w0.z, fmaf(a[j].w, w0.w, acc[j][0]))));
            acc[j][1] = fmaf(a[j].x, w1.x, fmaf(a[j].y, w1.y, fmaf(a[j].z, w1.z, fmaf(a[j].w, w1.w, acc[j][1]))));
            acc[j][2] = fmaf(a[j].x, w2.x, fmaf(a[j].y, w2.y, fmaf(a[j].z, w2.z, fmaf(a[j].w, w2.w, acc[j][2]))));
            acc[j][3] = fmaf(a[j].x, w3.x, fmaf(a[j].y, w3.y, fmaf(a[j].z, w3.z, fmaf(a[j].w, w3.w, acc[j][3]))));
        }
    }
    // combine the L channel-quad partial sums of each pixel (lanes of one pixel are contiguous, L is a power of two)
    for (int off = L >> 1; off > 0; off >>= 1) {
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[j][n] += __shfl_xor(acc[j][n], off, 64);
    }
    // after the butterfly every lane of a pixel holds the full sums: lane cq finishes channel cq (cq + L, ...), so the
    // L lanes of a pixel write neighbouring floats instead of one lane writing them all
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int m = mm[j];
        if (m >= M) continue;
        const int b = m / HWm;
        const int r = m - b * HWm;
        const int y = r / p.Wm;
        const int x = r - y * p.Wm;
        const int oy = cl.oy0 + y * p.s_out, ox = cl.ox0 + x * p.s_out;
        if (oy >= p.Hout || ox >= p.Wout) continue;
        const size_t o = ((size_t)b * p.Hout + oy) * p.Wout + ox;
        for (int n = cq; n < p.Cout; n += L) {
            float v = (n == 0 ? acc[j][0] : n == 1 ? acc[j][1] : n == 2 ? acc[j][2] : acc[j][3]) +
                      (p.bias != nullptr ? p.bias[n] : 0.f);
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
}

int launch_thin(const spaa_tapconv_t& d, hipStream_t stream) {
    constexpr int P = 4;
    if (d.Cout > 4 || d.Cin < 4 || d.Cin > 256 || (d.Cin & (d.Cin - 1))) return hipErrorInvalidValue;
    int maxtaps = 0;
    for (int c = 0; c < d.nclass; ++c) maxtaps = d.cls[c].ntaps > maxtaps ? d.cls[c].ntaps : maxtaps;
    const size_t smem = (size_t)maxtaps * (4 * d.Cin * sizeof(float) + sizeof(int2)) + 16;
    if (smem > 64 * 1024) return hipErrorInvalidValue;
    const int L = d.Cin / 4, PB = 256 / L;
    const int64_t M = (int64_t)d.B * d.Hm * d.Wm;
    dim3 grid((unsigned)((M + (int64_t)PB * P - 1) / ((int64_t)PB * P)), d.nclass, 1);
    hipLaunchKernelGGL((thinconv_kernel<P>), grid, dim3(256), smem, stream, d);
    return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN>
int launch(const spaa_tapconv_t& d, hipStream_t stream) {
    const int64_t M = (int64_t)d.B * d.Hm * d.Wm;
    const int m_tiles = (int)((M + BM - 1) / BM);
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const size_t smem = (size_t)(TAP_SMEM_FLOATS + 2 * (BM + BN) * LDK) * sizeof(float);
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&tapconv_kernel<BM, BN, WM, WN>), (int)smem, attr_set);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(m_tiles * n_tiles, d.nclass, 1);
    hipLaunchKernelGGL((tapconv_kernel<BM, BN, WM, WN>), grid, dim3(256), smem, stream, d, m_tiles, n_tiles);
    return (int)hipGetLastError();
}

}  // namespace

// layout probes for language bindings (tests compare them with the ctypes mirror of the struct)
extern "C" int spaa_tapconv_sizeof(void) { return (int)sizeof(spaa_tapconv_t); }
extern "C" int spaa_tapconv_offsetof(int field) {
    switch (field) {
        case 0: return (int)offsetof(spaa_tapconv_t, out);
        case 1: return (int)offsetof(spaa_tapconv_t, weights);
        case 2: return (int)offsetof(spaa_tapconv_t, taps);
        case 3: return (int)offsetof(spaa_tapconv_t, gate2);
        case 4: return (int)offsetof(spaa_tapconv_t, mask_out);
        case 5: return (int)offsetof(spaa_tapconv_t, tap_range);
        case 6: return (int)offsetof(spaa_tapconv_t, splitk_ws);
        case 7: return (int)offsetof(spaa_tapconv_t, io_dtype);
        case 8: return (int)offsetof(spaa_tapconv_t, nclass);
        case 9: return (int)offsetof(spaa_tapconv_t, cls);
        case 10: return (int)offsetof(spaa_tapconv_t, in2);
        case 11: return (int)offsetof(spaa_tapconv_t, w2_split);
        default: return -1;
    }
}

extern "C" int spaa_tapconv_f32(const spaa_tapconv_t* desc, spaa_stream_t stream_) {
    const spaa_tapconv_t& d = *desc;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    // host-side shape checks: a faulting kernel can take the whole node down
    if (d.in == nullptr || d.out == nullptr || d.weights == nullptr || d.taps == nullptr) return hipErrorInvalidValue;
    // (Winograd with two sources: the last Cin2 of the Cin input channels are read from `in2`)
    const int cin_main = (d.in2 != nullptr && (d.tile == 70 || d.tile == 71 || d.tile == 73 || (d.tile == 68 && d.nfold <= 1))) ? d.Cin - d.Cin2 : d.Cin;
    if (d.Cin <= 0 || (d.Cin & 3) || (d.in_cstride & 3) || (d.in_coff & 3) || cin_main <= 0 || d.in_coff + cin_main > d.in_cstride)
        return hipErrorInvalidValue;
    if (d.Cout <= 0 || d.out_coff + d.Cout > d.out_cstride) return hipErrorInvalidValue;
    if (d.gate2 != nullptr && (d.aux_out == nullptr || d.act == SPAA_ACT_RELU_CLAMP1)) return hipErrorInvalidValue;
    if (d.mask_out != nullptr || d.gate_bits != nullptr || d.gate2_bits != nullptr) {
        // byte masks (1 byte per 4 channels): only the epilogues built on epilogue.hpp's store4 know them, and only in its
        // 4-channel-vector form
        const int t = d.tile;
        if (!((t >= 15 && t <= 27) || (t >= 30 && t <= 46) || (t >= 48 && t <= 54) || (t >= 60 && t <= 65) || t == 68 || t == 70 || t == 71 || t == 73 || t == 74 || t == 76)) return hipErrorInvalidValue;
        if ((d.Cout | d.out_cstride | d.out_coff) & 3) return hipErrorInvalidValue;
        if (d.add != nullptr && ((d.add_cstride | d.add_coff) & 3)) return hipErrorInvalidValue;
        if (d.gate_bits != nullptr && (d.gate != nullptr || ((d.gate_cstride | d.gate_coff) & 3))) return hipErrorInvalidValue;
        if (d.gate2_bits != nullptr && (d.gate2 != nullptr || d.aux_out == nullptr || d.act == SPAA_ACT_RELU_CLAMP1 ||
                                        ((d.gate2_cstride | d.gate2_coff) & 3)))
            return hipErrorInvalidValue;
        if (d.gate != nullptr && ((d.gate_cstride | d.gate_coff) & 3)) return hipErrorInvalidValue;
        if (d.gate2 != nullptr && ((d.gate2_cstride | d.gate2_coff) & 3)) return hipErrorInvalidValue;
    }
    if (d.nclass < 1 || d.nclass > SPAA_MAX_CLASSES || d.B <= 0 || d.Hm <= 0 || d.Wm <= 0) return hipErrorInvalidValue;
    if (d.s_in < 1 || d.s_out < 1) return hipErrorInvalidValue;
    for (int c = 0; c < d.nclass; ++c) {
        const spaa_tapclass_t& cl = d.cls[c];
        if (cl.ntaps < 0 || cl.ntaps > SPAA_MAX_TAPS || cl.K != cl.ntaps * d.Cin || (cl.Kpad % BK) || cl.Kpad < cl.K)
            return hipErrorInvalidValue;
    }
    // buffer loads use 32-bit byte offsets; 0x80000000 is the out-of-bounds sentinel
    if ((int64_t)d.B * d.Hin * d.Win * d.in_cstride * 4 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    for (int c = 0; c < d.nclass; ++c)
        if ((int64_t)((d.Cout + 127) & ~127) * d.cls[c].Kpad * 4 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    int tile = d.tile;
    if (d.in2 != nullptr && tile != 74 && tile != 68 && tile != 72 && tile != 70 && tile != 71 && tile != 73) return hipErrorInvalidValue;   // (second source: the patch-staged stride-2 kernels and the Winograd kernel)
    // fp16-storage mode: fp16 inputs only through the h16 kernels; fp16 outputs only through the shared epilogue
    if ((d.io_dtype & SPAA_IO_IN_F16) && !((tile >= 60 && tile <= 65) || tile == 68 || ((tile == 29 || tile == 72) && !(d.io_dtype & SPAA_IO_OUT_F16)))) return hipErrorInvalidValue;
    if (!(d.io_dtype & SPAA_IO_IN_F16) && tile >= 60 && tile <= 63) return hipErrorInvalidValue;
    if ((d.io_dtype & SPAA_IO_OUT_F16) && !((tile >= 15 && tile <= 24) || tile == 38 || (tile >= 60 && tile <= 65) || tile == 68 || tile == 76))
        return hipErrorInvalidValue;
    if ((d.io_dtype & SPAA_IO_OUT_F16) && (d.ksplit < 0 || (d.ksplit > 1 && !((tile >= 60 && tile <= 63) || tile == 68)))) return hipErrorInvalidValue;  // (fp32 partial sums: only the fp16 kernels' own second passes write fp16)
    if (d.gate != nullptr && d.gate_mode == SPAA_GATE_MUL && tile < 25) return hipErrorInvalidValue;
    if (d.nfold > 1 && !((tile >= 25 && tile <= 27) || (tile >= 30 && tile <= 37) || (tile >= 39 && tile <= 46) || (tile >= 48 && tile <= 54) || (tile >= 60 && tile <= 65) || tile == 68))
        return hipErrorInvalidValue;
    if (tile == 0) {  // heuristic: widest N tile that fits Cout; shrink M when the grid would not fill 256 CUs twice
        const int64_t M = (int64_t)d.B * d.Hm * d.Wm * d.nclass;
        if (d.Cout > 64) tile = (M / 128) * ((d.Cout + 127) / 128) >= 512 ? 1 : 6;
        else if (d.Cout > 32) tile = (M / 128) >= 512 ? 4 : 6;
        else tile = 5;
    }
    switch (tile) {
        case 1: return launch<128, 128, 64, 64>(d, stream);
        case 2: return launch<256, 64, 64, 64>(d, stream);
        case 3: return launch<256, 32, 64, 32>(d, stream);
        case 4: return launch<128, 64, 64, 32>(d, stream);
        case 5: return launch<128, 32, 32, 32>(d, stream);
        case 6: return launch<64, 64, 32, 32>(d, stream);
        case 7: return launch<64, 128, 32, 64>(d, stream);
        case 8: return launch<128, 64, 32, 64>(d, stream);
        case 9: return launch_direct<4>(d, stream);
        case 10: return launch_direct<32>(d, stream);
        case 11: return launch_thin(d, stream);
        case 12:
        case 13:
        case 14:
        case 15:
        case 16:
        case 17:
        case 18:
        case 19:
        case 20:
        case 21:
        case 22:
        case 23:
        case 24: return spaa_launch_tapconv_x6(d, tile, stream);
        case 25:
        case 26:
        case 27: return spaa_launch_tapconv_x6d(d, tile, stream);
        case 28:
        case 29: return spaa_launch_thinpatch(d, stream);
        case 30:
        case 31:
        case 32:
        case 33:
        case 34:
        case 35:
        case 36:
        case 37: return spaa_launch_tapconv_x6d(d, tile, stream);
        case 38: return spaa_launch_smallcin(d, stream);
        case 39:
        case 40:
        case 41:
        case 42:
        case 43:
        case 44:
        case 45:
        case 46: return spaa_launch_tapconv_x6d(d, tile, stream);
        case 47: return spaa_launch_thinpatch(d, stream);
        case 48:
        case 49:
        case 50:
        case 51:
        case 52:
        case 53:
        case 54: return spaa_launch_tapconv_x6d(d, tile, stream);
        case 60:
        case 61:
        case 62:
        case 63:
        case 64:
        case 65: return spaa_launch_tapconv_h16(d, tile, stream);
        case 68: return spaa_launch_tapconv_h16p(d, stream);
        case 70:
        case 71:
        case 73: return spaa_launch_tapconv_wino(d, stream);
        case 72: return spaa_launch_tapconv_thinmf(d, stream);
        case 74: return spaa_launch_tapconv_x6p(d, stream);
        case 76: return spaa_launch_tapconv_c3(d, stream);
        default: return hipErrorInvalidValue;
    }
}
