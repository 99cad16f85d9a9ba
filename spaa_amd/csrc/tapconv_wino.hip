// tapconv_wino.hip — 3x3 / stride-1 / pad-1 convolutions (and their input gradients) by Winograd F(2x2, 3x3) on the bf16x6
// matrix-core arithmetic of tapconv_x6d.hip: 16 multiplications per 2x2 output tile instead of 36.
//
// The six 64x64 x (128 <-> 256 channel) layers of ShadingNetSPAA (conv4, conv5, conv4_s and their input gradients: 69 % of
// PCNet's MACs, /root/reference/src/python/models.py:286-298) sit at the chip's power wall in the direct form (DESIGN.md
// section 3): fewer products is the only lever left there.
//     Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        g: 3x3 filter, d: 4x4 input patch (stride 2), Y: 2x2 outputs
//   * U = G g G^T is computed on the host in fp64, rounded ONCE to fp32 and split exactly into three bf16 planes, packed like
//     a 16-tap weight matrix  W[n][pos * Cin + c],  pos = 4 xi + nu  (spaa_amd/convplan.py: wino_*_plan);
//   * V = B^T d B (entries are sums / differences of four inputs: +-1 coefficients, fp32 adds) is formed in registers from a
//     patch of the input staged ONCE per 32-channel block in LDS by LDS-DMA (out-of-image pixels: the out-of-range offset,
//     zeros = the convolution's zero padding), split into three bf16 fragments like every other activation operand;
//   * per position one 32-deep product  M_pos = U_pos . V_pos  (6 bf16 MFMAs per 16x16 block, fp32 accumulation) is folded
//     into the four output accumulators with A^T . A's +-1 coefficients:  16 + 4x16 accumulator registers per 16 channels;
//   * a workgroup (8 waves) owns 8 x 16 Winograd tiles (16 x 32 output pixels) of one image x 128 output channels; wave w owns
//     tile row w (16 tiles = the 16 columns of the MFMA's B operand); the U planes of a (position, channel block) are shared
//     by the workgroup: 24 KB per step, two LDS stages, one barrier per step, DMA of step t+1 under step t's MFMAs.
// Accuracy: measured against fp64 the error is ~2x that of the direct fp32 sum (the transforms add roundings); gated by
// tests/test_gpu_parity.py::test_winograd_*.
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ unsigned int cvt2(float a, float b) {
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned int p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned int p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 fp32 -> three bf16x8 with x == h + m + l exactly
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u4 hh, mm, ll;
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

constexpr int TY = 8, TX = 16;                 // Winograd tiles per workgroup (rows x columns): 16 x 32 output pixels
constexpr int PH = 2 * TY + 2, PW = 2 * TX + 2;  // input patch 18 x 34 pixels
constexpr int NPX = PH * PW;                   // 612
constexpr int NPIECE = (NPX + 7) / 8;          // 1-KiB pieces of 8 pixels x 32 channels (fp32)
constexpr int PATCH_BYTES = NPIECE * 1024;     // 78848

// 64-byte weight rows (32 bf16), chunk swizzle as tapconv_x6d.hip swz_w<16>
__device__ __forceinline__ int swz_w16(int n) { return ((n >> 3) & 1) << 1; }

template <int BN>
__global__ __launch_bounds__(512, 1) void wino_x6_kernel(const spaa_tapconv_t p, const int wg_y, const int wg_x, const int n_tiles) {
    constexpr int TJ = BN / 16;
    constexpr int W_PLANE = BN * 64;
    constexpr int WS_BYTES = 3 * W_PLANE;
    constexpr int W_PIECES = 3 * BN / 16;
    constexpr int NW = 8;
    constexpr int WPW = (W_PIECES + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* wsm = smem + PATCH_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const spaa_tapclass_t cl = p.cls[0];
    const int Cin = p.Cin, H = p.Hin, W = p.Win;

    // XCD-aware order over (image, patch row, patch column, n tile): an XCD takes a contiguous range
    int n_blk, img, oy0, ox0;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        n_blk = (t % n_tiles) * BN;
        t /= n_tiles;
        ox0 = (t % wg_x) * (2 * TX);
        t /= wg_x;
        oy0 = (t % wg_y) * (2 * TY);
        img = t / wg_y;
    }
    const int row_bytes = p.in_cstride * 4;
    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(H * W) * (uint32_t)row_bytes;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    const uint32_t in_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int npad = (p.Cout + 127) & ~127;
    const int plane_bytes = npad * cl.Kpad * 2;
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_split);
    const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
                                                           (int)__builtin_amdgcn_readfirstlane(3u * (uint32_t)plane_bytes), 0x00020000);

    // ---- patch staging: piece i (8 consecutive pixel SLOTS) -> wave i % 8; lane -> (slot lane >> 3, physical chunk lane & 7).
    // LDS layout (bank-conflict-free for a wave's reads: 16 tiles = patch columns 2 apart, 4 channel chunks):
    //   slot of patch pixel p = row * PW + column:  p ^ ((p >> 1) & 1)   (neighbouring pixel pairs alternate their order, so
    //   pixels 2 apart alternate between the two 128-byte halves of the 64 banks);
    //   16-byte chunk L of the pixel's 32 channels sits at chunk  L ^ ((column >> 2) & 7).
    constexpr int PPW = (NPIECE + NW - 1) / NW;  // pieces per wave
    int pa_off[PPW];                             // source byte offset without the channel block (or the OOB sentinel)
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int piece = wave + NW * i;
        const int slot = piece * 8 + (lane >> 3);
        const int pix = slot ^ ((slot >> 1) & 1);
        const int pr = pix / PW, pc = pix - pr * PW;
        const int c = (lane & 7) ^ ((pc >> 2) & 7);
        const int iy = oy0 - 1 + pr, ix = ox0 - 1 + pc;
        const bool ok = piece < NPIECE && pix < NPX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        pa_off[i] = ok ? ((img * H + iy) * W + ix) * row_bytes + p.in_coff * 4 + c * 16 : (int)0x80000000;
    }
    // ---- weight staging (as tapconv_x6d.hip): piece q = wave + 8 i -> (plane, 16-row block); lane -> (row, physical chunk)
    int w_goff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int q = wave + NW * i;
        const int pl = q / (BN / 16), rb = q % (BN / 16);
        const int n = 16 * rb + (lane >> 2);
        const int c = (lane & 3) ^ swz_w16(n);
        w_goff[i] = pl * plane_bytes + (n_blk + n) * cl.Kpad * 2 + c * 16;
    }
    const int w_addr_l = (lane & 15) * 64 + (((lane >> 4) ^ swz_w16(lane & 15)) * 16);

    // ---- this lane's tile: row `wave`, column lane & 15, channels 8 (lane >> 4) .. + 7 of the block
    const int tx = lane & 15, q8 = lane >> 4;

    f32x4 Y[4][TJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) Y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkb = Cin >> 5;           // 32-channel blocks
    const int nsteps = nkb * 16;        // (block, position) steps
#define WINO_DMA_W(st, step)                                                                                      \
    {                                                                                                              \
        const int kb_ = (step) >> 4, pos_ = (step) & 15;                                                           \
        const int soff_ = (pos_ * Cin + kb_ * 32) * 2;                                                             \
        _Pragma("unroll") for (int i = 0; i < WPW; ++i) if (W_PIECES % NW == 0 || wave + NW * i < W_PIECES)        \
            dma16(rsrc_w, wsm + (st) * WS_BYTES + (wave + NW * i) * 1024, w_goff[i], soff_);                       \
    }
    WINO_DMA_W(0, 0)
    for (int kb = 0; kb < nkb; ++kb) {
        // ---- stage the patch of this channel block (everybody is past the previous block's reads)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (wave + NW * i < NPIECE)
                dma16(rsrc_in, smem + (wave + NW * i) * 1024, pa_off[i] == (int)0x80000000 ? pa_off[i] : pa_off[i] + kb * 128, 0);
#pragma unroll 1
        for (int pos = 0; pos < 16; ++pos) {
            const int step = kb * 16 + pos;
            const int xi = pos >> 2, nu = pos & 3;
            // B^T rows: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3  ->  s1 * d[a1] + s2 * d[a2]
            const int a1 = xi == 0 ? 0 : 1, a2 = xi == 3 ? 3 : 2;
            const float s1 = xi == 2 ? -1.f : 1.f, s2 = (xi == 1 || xi == 2) ? 1.f : -1.f;
            const int b1 = nu == 0 ? 0 : 1, b2 = nu == 3 ? 3 : 2;
            const float t1 = nu == 2 ? -1.f : 1.f, t2 = (nu == 1 || nu == 2) ? 1.f : -1.f;
            // A^T = [[1,1,1,0],[0,1,-1,-1]]: coefficient of M(xi, nu) in output (i, j) = At[i][xi] * At[j][nu]
            const float ci0 = xi < 3 ? 1.f : 0.f, ci1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);
            const float cj0 = nu < 3 ? 1.f : 0.f, cj1 = nu == 0 ? 0.f : (nu == 1 ? 1.f : -1.f);
            const float c00 = ci0 * cj0, c01 = ci0 * cj1, c10 = ci1 * cj0, c11 = ci1 * cj1;
            // own DMAs (patch pieces, weight pieces of this step) have landed; everybody's have after the barrier; every wave is
            // also past its reads of the other weight stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step + 1 < nsteps) WINO_DMA_W((step + 1) & 1, step + 1)
            float vv[8];
            {
                // pixel (row 2 wave + a, column 2 tx + b): p = row * PW + column, (p >> 1) & 1 = (a + tx + (b >> 1)) & 1 (PW / 2 is odd)
                const int u1 = tx + (b1 >> 1), u2 = tx + (b2 >> 1);
                const int p11 = (2 * wave + a1) * PW + 2 * tx + b1, p12 = (2 * wave + a1) * PW + 2 * tx + b2;
                const int p21 = (2 * wave + a2) * PW + 2 * tx + b1, p22 = (2 * wave + a2) * PW + 2 * tx + b2;
                const int o11 = (p11 ^ ((a1 + u1) & 1)) << 7, o12 = (p12 ^ ((a1 + u2) & 1)) << 7;
                const int o21 = (p21 ^ ((a2 + u1) & 1)) << 7, o22 = (p22 ^ ((a2 + u2) & 1)) << 7;
                const int z1 = (u1 >> 1) & 7, z2 = (u2 >> 1) & 7;
                const int h0a = ((2 * q8) ^ z1) << 4, h1a = ((2 * q8 + 1) ^ z1) << 4;
                const int h0b = ((2 * q8) ^ z2) << 4, h1b = ((2 * q8 + 1) ^ z2) << 4;
                const f32x4 d11l = *reinterpret_cast<const f32x4*>(smem + o11 + h0a);
                const f32x4 d11h = *reinterpret_cast<const f32x4*>(smem + o11 + h1a);
                const f32x4 d12l = *reinterpret_cast<const f32x4*>(smem + o12 + h0b);
                const f32x4 d12h = *reinterpret_cast<const f32x4*>(smem + o12 + h1b);
                const f32x4 d21l = *reinterpret_cast<const f32x4*>(smem + o21 + h0a);
                const f32x4 d21h = *reinterpret_cast<const f32x4*>(smem + o21 + h1a);
                const f32x4 d22l = *reinterpret_cast<const f32x4*>(smem + o22 + h0b);
                const f32x4 d22h = *reinterpret_cast<const f32x4*>(smem + o22 + h1b);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (+-1 coefficients: every product is exact, each line rounds once per addition)
                    const float ra = __builtin_fmaf(d12l[e], t2, d11l[e] * t1), rb = __builtin_fmaf(d22l[e], t2, d21l[e] * t1);
                    vv[e] = __builtin_fmaf(rb, s2, ra * s1);
                    const float rc = __builtin_fmaf(d12h[e], t2, d11h[e] * t1), rd = __builtin_fmaf(d22h[e], t2, d21h[e] * t1);
                    vv[4 + e] = __builtin_fmaf(rd, s2, rc * s1);
                }
            }
            bf16x8 pf[3];
            split8(vv, pf[0], pf[1], pf[2]);
            const unsigned char* wc = wsm + (step & 1) * WS_BYTES + w_addr_l;
#pragma unroll
            for (int jq = 0; jq < TJ; jq += 2) {
                f32x4 m[2];
                bf16x8 w0[2], w1[2], w2[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    w0[u] = *reinterpret_cast<const bf16x8*>(wc + (jq + u) * 1024);
                    w1[u] = *reinterpret_cast<const bf16x8*>(wc + (jq + u) * 1024 + W_PLANE);
                    w2[u] = *reinterpret_cast<const bf16x8*>(wc + (jq + u) * 1024 + 2 * W_PLANE);
                    m[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                // weights = A operand (rows = output channels), tiles = B operand (columns); small terms first; two independent
                // chains interleaved
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[u], pf[0], m[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[u], pf[2], m[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[u], pf[1], m[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[u], pf[0], m[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[u], pf[1], m[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[u], pf[0], m[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        Y[0][jq + u][e] = __builtin_fmaf(m[u][e], c00, Y[0][jq + u][e]);
                        Y[1][jq + u][e] = __builtin_fmaf(m[u][e], c01, Y[1][jq + u][e]);
                        Y[2][jq + u][e] = __builtin_fmaf(m[u][e], c10, Y[2][jq + u][e]);
                        Y[3][jq + u][e] = __builtin_fmaf(m[u][e], c11, Y[3][jq + u][e]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#undef WINO_DMA_W

    // ---- epilogue: D layout of a 16x16 block: column (lane & 15) = tile, rows 4 (lane >> 4) + e = 4 consecutive channels
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int oy = oy0 + 2 * wave + (i >> 1), ox = ox0 + 2 * tx + (i & 1);
        if (oy < p.Hout && ox < p.Wout) {
            const size_t o = ((size_t)img * p.Hout + oy) * p.Wout + ox;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float v[4] = {Y[i][j][0], Y[i][j][1], Y[i][j][2], Y[i][j][3]};
                store4_t<float>(p, o, n_blk + 16 * j + 4 * q8, v, vec);
            }
        }
    }
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tile 70 after the common shape checks.  The descriptor describes the layer in
// Winograd form: ONE class with 16 "taps" = the positions of U = G g G^T (tap entries unused), s_in = s_out = 1, same input and
// output size, Cin % 32 == 0.
int spaa_launch_tapconv_wino(const spaa_tapconv_t& d, hipStream_t stream) {
    constexpr int BN = 128;
    if (d.w_split == nullptr || (d.Cin % 32) != 0 || d.nclass != 1 || d.cls[0].ntaps != 16 || d.cls[0].K != 16 * d.Cin || d.cls[0].Kpad < d.cls[0].K || (d.cls[0].Kpad & 7) ||
        d.s_in != 1 || d.s_out != 1 || d.Hin != d.Hout || d.Win != d.Wout || d.Hm != d.Hout || d.Wm != d.Wout || d.nfold > 1 ||
        d.ksplit > 1 || d.ksplit < 0 || d.io_dtype != 0)
        return hipErrorInvalidValue;
    if ((int64_t)((d.Cout + 127) & ~127) * d.cls[0].Kpad * 6 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    const int wg_y = (d.Hout + 2 * TY - 1) / (2 * TY), wg_x = (d.Wout + 2 * TX - 1) / (2 * TX);
    const int n_tiles = (d.Cout + BN - 1) / BN;
    const int64_t nwg = (int64_t)d.B * wg_y * wg_x * n_tiles;
    if (nwg > 0x7fffffff) return hipErrorInvalidValue;
    const size_t smem = (size_t)PATCH_BYTES + 2 * (size_t)(3 * BN * 64);
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&wino_x6_kernel<BN>), (int)smem, attr_set);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL((wino_x6_kernel<BN>), dim3((unsigned)nwg), dim3(512), smem, stream, d, wg_y, wg_x, n_tiles);
    return (int)hipGetLastError();
}
