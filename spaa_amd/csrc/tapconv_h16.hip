// tapconv_h16.hip — the tap-list convolution in fp16-STORAGE mode (BASELINE.json configs[4]: "fp16 with fp32 dE2000").
//
// Activations and gradients live in HBM as fp16, the (frozen) weights are rounded to fp16 once per model, products are
// accumulated in fp32 on the matrix cores (v_mfma_f32_16x16x32_f16), and the epilogue (bias, residual, activation, ReLU
// gates, byte masks: epilogue.hpp) works in fp32 and rounds once on the way out.  Compared with the fp32 path
// (tapconv_x6d.hip: six bf16 MFMAs per product, fp32 in HBM) this is one MFMA per product and half the bytes, which is
// what moves the PCNet + dE2000 forward/backward from the matrix-core roofline to the HBM roofline.
//
// Structure (the x6d kernel's, with nothing to split):
//   * K advances in steps of 64 = two SUB-steps of 32 channels; a sub-step lies inside one tap (Cin % 32 == 0), so the
//     tap offset is wave-uniform per sub-step; with several taps and Cin > 32 the sub-steps run chunk-major (all taps
//     of a 32-channel slice while it is L2-resident);
//   * both operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): 64-byte rows (32 fp16), 16 rows per
//     1-KiB piece, 16-byte chunks XOR-swizzled on the SOURCE side (conflict-free ds_read_b128 of 16x16x32 fragments);
//     out-of-image taps use the out-of-range offset 0x80000000 (the DMA writes zeros);
//   * a wave owns 32 pixels x BN output channels and stages / reads back only its own pixel rows; the weight rows are
//     shared by the workgroup: two LDS stages, one barrier per K-step, the DMA of step t+1 lands during step t's MFMAs.
// Thin outputs (conv6: 32 -> 3, the dgrads of conv1 / conv1_s / the ResNet stem) use the BN = 16 instantiation: with one
// MFMA per product a 16-wide N tile costs nothing next to the activation bytes.
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"
#include "epilogue.hpp"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned char* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
}

// chunk swizzle of a 64-byte row (4 chunks): see tapconv_x6d.hip swz_w<16>
__device__ __forceinline__ int swz64(int r) { return ((r >> 3) & 1) << 1; }

// NS = LDS stages.  2: the DMA of step t + 1 lands during step t's MFMAs -- enough when two workgroups per CU (or a long
// MFMA phase) cover the load latency.  4: the small-M layers (ResNet layer3 / layer4 at batch 64: fewer workgroups than CUs,
// one wave per SIMD, 16 MFMAs per step against ~1 us of load latency) keep three steps in flight behind a counted vmcnt.
template <int NW, int BN, int NS = 2>
__global__ __launch_bounds__(64 * NW, (NW > 4 || NS > 2) ? 1 : 2) void tapconv_h16_kernel(const spaa_tapconv_t p, const int m_tiles, const int n_tiles) {
    constexpr int BM = 32 * NW;
    constexpr int TJ = BN / 16;
    constexpr int A_SUB = BM * 64;               // one sub-step's pixel rows
    constexpr int W_SUB = BN * 64;               // one sub-step's weight rows
    constexpr int STAGE = 2 * A_SUB + 2 * W_SUB;  // [A sub0][A sub1][W sub0][W sub1]
    constexpr int W_PIECES = 2 * BN / 16;         // 1-KiB weight pieces per K-step
    constexpr int WPW = (W_PIECES + NW - 1) / NW;

    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const spaa_tapclass_t cl = p.cls[blockIdx.y];

    // XCD-aware tile order: the workgroups of one XCD (blockIdx.x % 8) take a contiguous range of tiles
    int n_blk, m_blk;
    {
        const int nwg = m_tiles * n_tiles, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
        n_blk = (tile % n_tiles) * BN;
        m_blk = (tile / n_tiles) * BM;
    }
    const int HWm = p.Hm * p.Wm;
    const int M = p.B * HWm;
    const int Cin = p.Cin;
    const int row_bytes = p.in_cstride * 2;
    typedef const __attribute__((address_space(4))) int* cint_ptr;
    cint_ptr ctaps = (cint_ptr)(uintptr_t)(p.taps + 2 * cl.tap_off);

    const uint32_t in_bytes = (uint32_t)p.B * (uint32_t)(p.Hin * p.Win) * (uint32_t)row_bytes;
    const uint64_t in_addr = reinterpret_cast<uint64_t>(p.in);
    // (readfirstlane returns a SIGNED int: keep the halves in uint32_t, or a low word with bit 31 set sign-extends into
    // the high word of the base address)
    const uint32_t in_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)in_addr);
    const uint32_t in_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(in_addr >> 32));
    const auto rsrc_in = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)in_hi << 32) | in_lo), 0,
                                                            (int)__builtin_amdgcn_readfirstlane(in_bytes), 0x00020000);
    const int K64 = (cl.K + 63) & ~63;
    const int npad = (p.Cout * (p.nfold > 1 ? p.nfold : 1) + 127) & ~127;
    uint64_t wh_off = 0;  // fp16 plane of class c: [npad][K64_c], classes back to back
    for (int c = 0; c < (int)blockIdx.y; ++c) wh_off += (uint64_t)npad * (uint64_t)((p.cls[c].K + 63) & ~63);
    const uint64_t w_addr = reinterpret_cast<uint64_t>(p.w_half) + wh_off * 2u;
    const uint32_t w_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)w_addr);
    const uint32_t w_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(w_addr >> 32));
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uint64_t)w_hi << 32) | w_lo), 0,
        (int)__builtin_amdgcn_readfirstlane((uint32_t)npad * (uint32_t)K64 * 2u), 0x00020000);

    // ---- this wave's pixel rows: piece ib (16 rows) -> lane holds row 32 wave + 16 ib + (lane >> 2), physical chunk lane & 3
    int a_off[2];
    uint32_t a_mlo[2], a_mhi[2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
        const int r = 32 * wave + 16 * ib + (lane >> 2);
        const int c = (lane & 3) ^ swz64(r);  // logical 16-byte chunk (8 channels) held at this lane's LDS slot
        const int m = m_blk + r;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int b = mm / HWm;
        const int rr = mm - b * HWm;
        const int y = rr / p.Wm;
        const int x = rr - y * p.Wm;
        const int iy0 = y * p.s_in, ix0 = x * p.s_in;
        a_off[ib] = ((b * p.Hin + iy0) * p.Win + ix0) * row_bytes + p.in_coff * 2 + c * 16;
        uint32_t lo = 0, hi = 0;
        for (int t = 0; t < cl.ntaps; ++t) {
            const int dy = ctaps[2 * t], dx = ctaps[2 * t + 1];
            const bool v = ok && (unsigned)(iy0 + dy) < (unsigned)p.Hin && (unsigned)(ix0 + dx) < (unsigned)p.Win;
            if (t < 32) lo |= (v ? 1u : 0u) << t;
            else hi |= (v ? 1u : 0u) << (t - 32);
        }
        a_mlo[ib] = lo;
        a_mhi[ib] = hi;
    }
    // ---- weight pieces of this wave: piece q = wave + NW i -> (sub-step, 16-row block)
    int w_goff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int q = wave + NW * i;
        const int rb = q % (BN / 16);
        const int n = 16 * rb + (lane >> 2);
        const int c = (lane & 3) ^ swz64(n);
        w_goff[i] = (n_blk + n) * K64 * 2 + c * 16;
    }

    const int nsub = cl.K / 32;          // real sub-steps
    const int nk_all = (nsub + 1) >> 1;  // K-steps (the last one may hold a single real sub-step)
    // split-K (p.ksplit > 1: skinny GEMMs -- few pixels, long K: the fully connected layers of VGG-16 at batch 64, ResNet layer4):
    // grid.z workgroups take equal ranges of the K-steps and write raw fp32 partial sums; h16_splitk_reduce_kernel adds them in
    // fixed order and applies the epilogue
    const int ksp = p.ksplit > 1 ? p.ksplit : 1;
    const int ks0 = (int)((int64_t)nk_all * blockIdx.z / ksp), ks1 = (int)((int64_t)nk_all * (blockIdx.z + 1) / ksp);
    const int nk = ks1 - ks0;            // this workgroup's K-steps: ks0 + 0 .. ks0 + nk - 1
    const bool chunk_major = (cl.ntaps > 1) && (Cin > 32);
    const int cpt = Cin >> 5;            // 32-channel chunks per tap

    // issue the DMAs of K-step `ks` into stage `st`
#define H16_STAGE(ks, st)                                                                                          \
    {                                                                                                              \
        unsigned char* sbase = smem + (st) * STAGE;                                                                \
        int wk[2];                                                                                                 \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                            \
            const int q = 2 * (ks) + s;                                                                            \
            const bool real = q < nsub;                                                                            \
            const int qq = real ? q : 0;                                                                           \
            const int tap = chunk_major ? qq % cl.ntaps : qq / cpt;                                                \
            const int kc = (chunk_major ? qq / cl.ntaps : qq % cpt) * 32;                                          \
            const int dy = ctaps[2 * tap], dx = ctaps[2 * tap + 1];                                                \
            const int tapoff = (dy * p.Win + dx) * row_bytes + kc * 2;                                             \
            wk[s] = real ? (tap * Cin + kc) * 2 : cl.K * 2; /* a padded sub-step reads the zero columns [K, K64) */ \
            _Pragma("unroll") for (int ib = 0; ib < 2; ++ib) {                                                     \
                const uint32_t mw = tap < 32 ? a_mlo[ib] >> tap : a_mhi[ib] >> (tap - 32);                         \
                const int voff = (real && (mw & 1u)) ? a_off[ib] + tapoff : (int)0x80000000;                       \
                dma16(rsrc_in, sbase + s * A_SUB + (32 * wave + 16 * ib) * 64, voff, 0);                           \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int i = 0; i < WPW; ++i) {                                                          \
            const int q = wave + NW * i;                                                                           \
            if (W_PIECES % NW == 0 || q < W_PIECES) {                                                              \
                const int s = q / (BN / 16), rb = q % (BN / 16);                                                   \
                dma16(rsrc_w, sbase + 2 * A_SUB + s * W_SUB + rb * 1024, w_goff[i], s ? wk[1] : wk[0]);                        \
            }                                                                                                      \
        }                                                                                                          \
    }

    // fragment read addresses (bytes inside a sub-step's image): lane -> row (lane & 15), k-chunk (lane >> 4)
    int p_addr[2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
        const int r = 32 * wave + 16 * ib + (lane & 15);
        p_addr[ib] = r * 64 + (((lane >> 4) ^ swz64(r)) * 16);
    }
    const int w_addr_l = (lane & 15) * 64 + (((lane >> 4) ^ swz64(lane & 15)) * 16);

    f32x4 acc[2][TJ];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[ib][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int DPS = 4 + WPW;   // DMAs a wave issues per K-step (NS > 2: W_PIECES % NW == 0, every wave issues all of them)
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < nk) H16_STAGE(ks0 + i, i)
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks) {
        // own DMAs of step ks have landed (vmcnt) and everybody's have (barrier); every wave is also past its reads of
        // the stage that is refilled during this step (read at step ks - 1)
        if constexpr (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            if (ks + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS * (NS - 2)) : "memory");
            else if (NS > 3 && ks + NS - 3 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS * (NS > 3 ? NS - 3 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        if (ks + NS - 1 < nk) {
            const int nxt = cur == 0 ? NS - 1 : cur - 1;
            H16_STAGE(ks0 + ks + NS - 1, nxt)
        }
        const unsigned char* sb = smem + cur * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const h8 pf0 = *reinterpret_cast<const h8*>(sb + s * A_SUB + p_addr[0]);
            const h8 pf1 = *reinterpret_cast<const h8*>(sb + s * A_SUB + p_addr[1]);
            const unsigned char* wb = sb + 2 * A_SUB + s * W_SUB + w_addr_l;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const h8 wf = *reinterpret_cast<const h8*>(wb + j * 1024);
                // weights = A operand (rows = output channels), pixels = B operand (columns)
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, pf0, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, pf1, acc[1][j], 0, 0, 0);
            }
        }
        cur = cur == NS - 1 ? 0 : cur + 1;
    }
#undef H16_STAGE

    if (ksp > 1) {   // raw partial sums [split][M][npad] (npad = Cout rounded up to 128)
        float* ws = p.splitk_ws + (size_t)blockIdx.z * M * npad;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            const int m = m_blk + 32 * wave + 16 * ib + (lane & 15);
            if (m < M) {
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    if (n_blk + 16 * j < npad) *reinterpret_cast<f32x4*>(ws + (size_t)m * npad + n_blk + 16 * j + 4 * (lane >> 4)) = acc[ib][j];
            }
        }
        return;
    }

    // ---- epilogue.  D layout of a 16x16 tile: column (lane & 15) = pixel, rows 4*(lane>>4) + i = 4 consecutive channels
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
#define H16_EPI_IB(T, ib)                                                                                          \
    {                                                                                                              \
        const int m = m_blk + 32 * wave + 16 * (ib) + (lane & 15);                                                 \
        if (p.nfold > 1) {                                                                                         \
            _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                       \
                float v[4] = {acc[ib][j][0], acc[ib][j][1], acc[ib][j][2], acc[ib][j][3]};                         \
                store4_fold_t<T>(p, m, M, HWm, n_blk + 16 * j + 4 * (lane >> 4), v, vec);                          \
            }                                                                                                      \
        } else {                                                                                                   \
            size_t o;                                                                                              \
            if (out_pixel(p, cl, m, M, HWm, o)) {                                                                  \
                _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                   \
                    float v[4] = {acc[ib][j][0], acc[ib][j][1], acc[ib][j][2], acc[ib][j][3]};                     \
                    store4_t<T>(p, o, n_blk + 16 * j + 4 * (lane >> 4), v, vec);                                   \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
    }
    if (fast_epi_ok(p, vec)) {   // branch-free operand accesses, a pixel's TJ channel quads in flight (epilogue.hpp: fast_epi_*)
        const fast_epi_t fe = make_fast_epi(p, 0);
#define H16_FAST_IB(T, ib)                                                                                         \
    {                                                                                                              \
        const int m = m_blk + 32 * wave + 16 * (ib) + (lane & 15);                                                 \
        size_t o1 = 0;                                                                                             \
        const bool ok1 = p.nfold > 1 ? false : out_pixel(p, cl, m, M, HWm, o1);                                    \
        constexpr int CH = TJ < 8 ? TJ : 8;                                                                        \
        _Pragma("unroll") for (int j0 = 0; j0 < TJ; j0 += CH) {                                                    \
            fast_pre_t<T> pre[CH];                                                                                 \
            int oo[CH], nn[CH];                                                                                    \
            bool ok[CH];                                                                                           \
            _Pragma("unroll") for (int j = 0; j < CH; ++j) {                                                       \
                const int n0 = n_blk + 16 * (j0 + j) + 4 * (lane >> 4);                                            \
                if (p.nfold > 1) {                                                                                 \
                    ok[j] = fold_pixel(p, m, M, HWm, n0, oo[j], nn[j]);                                            \
                } else {                                                                                           \
                    oo[j] = (int)o1, nn[j] = n0, ok[j] = ok1 && n0 < p.Cout;                                       \
                }                                                                                                  \
                pre[j] = fast_epi_load<T, true>(fe, p, oo[j], nn[j], ok[j]);                                       \
            }                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < CH; ++j)                                                         \
                fast_epi_store<T, f32x4, true>(fe, p, oo[j], nn[j], ok[j], acc[ib][j0 + j], pre[j]);               \
        }                                                                                                          \
    }
        if (p.io_dtype & SPAA_IO_OUT_F16) {
            H16_FAST_IB(_Float16, 0)
            H16_FAST_IB(_Float16, 1)
        } else {
            H16_FAST_IB(float, 0)
            H16_FAST_IB(float, 1)
        }
#undef H16_FAST_IB
        return;
    }
    if (p.io_dtype & SPAA_IO_OUT_F16) {  // fp16 activation / gradient out
        H16_EPI_IB(_Float16, 0)
        H16_EPI_IB(_Float16, 1)
    } else {                             // fp32 image out (conv6, the input-gradients of the image-side layers)
        H16_EPI_IB(float, 0)
        H16_EPI_IB(float, 1)
    }
#undef H16_EPI_IB
}

// second pass of split-K: out = epilogue( sum over the splits, in fixed order ), 4 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void h16_splitk_reduce_kernel(const spaa_tapconv_t p, const int M, const int npad) {
    const int nq = (p.Cout + 3) >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)M * nq) return;
    const int m = (int)(idx / nq), n0 = (int)(idx - (int64_t)m * nq) * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.ksplit; ++s) sum += *reinterpret_cast<const f32x4*>(p.splitk_ws + ((size_t)s * M + m) * npad + n0);
    size_t o;
    if (!out_pixel(p, p.cls[0], m, M, p.Hm * p.Wm, o)) return;
    const bool vec = !((p.Cout | p.out_cstride | p.out_coff) & 3) &&
                     (p.add == nullptr || !((p.add_cstride | p.add_coff) & 3)) &&
                     (p.gate == nullptr || !((p.gate_cstride | p.gate_coff) & 3)) &&
                     (p.gate2 == nullptr || !((p.gate2_cstride | p.gate2_coff) & 3));
    float v[4] = {sum[0], sum[1], sum[2], sum[3]};
    store4_t<T>(p, o, n0, v, vec);
}

template <int NW, int BN, int NS = 2>
int launch_h16(const spaa_tapconv_t& d, hipStream_t stream) {
    constexpr int BM = 32 * NW;
    const int64_t M = (int64_t)d.B * d.Hm * d.Wm;
    const int m_tiles = (int)((M + BM - 1) / BM);
    const int nfold = d.nfold > 1 ? d.nfold : 1;
    if (nfold > 1 && (nfold != 4 || d.nclass != 1 || d.s_out != 2 || (d.Cout & 3))) return hipErrorInvalidValue;
    const int n_tiles = (d.Cout * nfold + BN - 1) / BN;
    // at most one workgroup per compute unit and a long K loop (ResNet layer4 at batch 64: 64 -> 61 us): the four-stage
    // instantiation; with more workgroups than CUs two resident two-stage workgroups do better (layer3: 42 against 66 us)
    if constexpr (NS == 2 && NW == 4 && BN >= 32 && BN <= 128) {
        int64_t kmin = 1 << 30;
        for (int c = 0; c < d.nclass; ++c) kmin = d.cls[c].K < kmin ? d.cls[c].K : kmin;
        if ((int64_t)m_tiles * n_tiles * d.nclass * (d.ksplit > 1 ? d.ksplit : 1) <= 256 && kmin >= 16 * 64 * (d.ksplit > 1 ? d.ksplit : 1) && !((d.reserved0 >> 25) & 1)) return launch_h16<NW, BN, 4>(d, stream);
    }
    const size_t smem = NS * (size_t)(2 * BM * 64 + 2 * BN * 64);
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&tapconv_h16_kernel<NW, BN, NS>), (int)smem, attr_set);
        if (e != hipSuccess) return (int)e;
    }
    const int ksp = d.ksplit > 1 ? d.ksplit : 1;
    if (ksp > 1 && (d.nclass != 1 || nfold > 1 || d.splitk_ws == nullptr)) return hipErrorInvalidValue;
    dim3 grid(m_tiles * n_tiles, d.nclass, ksp);
    hipLaunchKernelGGL((tapconv_h16_kernel<NW, BN, NS>), grid, dim3(64 * NW), smem, stream, d, m_tiles, n_tiles);
    if (ksp > 1) {
        const int npad = (d.Cout + 127) & ~127;
        const int64_t nthr = M * ((d.Cout + 3) >> 2);
        if (d.io_dtype & SPAA_IO_OUT_F16)
            hipLaunchKernelGGL(h16_splitk_reduce_kernel<_Float16>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, d, (int)M, npad);
        else
            hipLaunchKernelGGL(h16_splitk_reduce_kernel<float>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream, d, (int)M, npad);
    }
    return (int)hipGetLastError();
}

}  // namespace

// called by spaa_tapconv_f32 (tapconv.hip) for tiles 60..65 after the common shape checks
int spaa_launch_tapconv_h16(const spaa_tapconv_t& d, int tile, hipStream_t stream) {
    if (!(d.io_dtype & SPAA_IO_IN_F16) || d.w_half == nullptr || (d.Cin % 32) != 0 || d.ksplit < 0 || (d.ksplit > 1 && tile > 63))
        return hipErrorInvalidValue;
    const int nfold = d.nfold > 1 ? d.nfold : 1;
    for (int c = 0; c < d.nclass; ++c) {
        if (d.cls[c].Kpad != d.cls[c].K) return hipErrorInvalidValue;  // (K % 32 == 0: rows of whole sub-steps)
        if ((int64_t)((d.Cout * nfold + 127) & ~127) * ((d.cls[c].K + 63) & ~63) * 2 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    }
    // buffer offsets are 32-bit byte offsets of fp16 elements
    if ((int64_t)d.B * d.Hin * d.Win * d.in_cstride * 2 >= (int64_t)1 << 31) return hipErrorInvalidValue;
    switch (tile) {
        case 60: return launch_h16<4, 128>(d, stream);
        case 61: return launch_h16<4, 64>(d, stream);
        case 62: return launch_h16<4, 32>(d, stream);
        case 63: return launch_h16<4, 16>(d, stream);
        case 64: return launch_h16<8, 128>(d, stream);   // 256 x 128: half the weight traffic per pixel
        case 65: return launch_h16<8, 256>(d, stream);   // 256 x 256: the 256-channel layers, one N tile
        default: return hipErrorInvalidValue;
    }
}
