// shading_tail.hip — the tail of ShadingNetSPAA.forward and the head of its backward pass as ONE kernel each
// (/root/reference/src/python/models.py:296-300:  x = relu(transConv2(x));  x = clamp(relu(conv6(x) + res1), max=1)).
//
// At the benchmark size the activation between the two layers (X7: 64 x 256 x 256 x 32 fp32 = 537 MB) is the largest
// tensor of the network and exists only to be read once by the next layer: forward, transConv2 writes it and conv6 reads
// it; backward, conv6's input gradient writes its gradient (P7) and transConv2's input gradient reads it.  Here a
// workgroup keeps its tile of X7 (resp. P7) in LDS:
//   forward   X6 tile (8 x 16 pixels, 64 ch) --bf16x6 MFMA, K = 64, N = 4 parities x 32--> X7 tile (16 x 32 px, 32 ch) in LDS
//             --3x3 taps on the VALU (packed FMAs, scalar weights), + res1, ReLU, clamp--> Y, Ypre for the 14 x 30 interior;
//             HBM sees X6, res1, Y, Ypre and the ReLU gate bytes of X7 (1 byte per 4 channels) only;
//   backward  gP tile (16 x 32 px + halo, 3 ch) --3x3 taps on the VALU, gated by X7's bytes--> P7 tile in LDS as the GEMM's
//             pixel operand (K = 4 parities x 32) --bf16x6 MFMA, N = 64, gated by X6's bytes--> P6 tile (8 x 16 px, 64 ch).
// The arithmetic is that of the separate kernels: fp32 operands split exactly into three bf16 planes, six of the nine
// partial products, fp32 accumulation (tapconv_x6d.hip); fp32 FMAs for the thin 3x3 layer (thinpatch.hip).
// Persistent: one workgroup (8 waves) per CU walks the tiles; the transposed convolution's weight planes stay in LDS.
#include <hip/hip_runtime.h>
#include "launch_util.hpp"
#include <stdint.h>
#include "../../include/spaa_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) h16v* ch16_ptr;

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(4))) f16v* cf16_ptr;

__device__ __forceinline__ unsigned int cvt2(float a, float b) {
    f2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_f(unsigned int p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned int p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 fp32 -> three bf16x8 with x == h + m + l exactly
__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
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

// six of the nine partial products of (w0 + w1 + w2) . (p0 + p1 + p2), small terms first
__device__ __forceinline__ f32x4 mfma6(const bf16x8 w0, const bf16x8 w1, const bf16x8 w2, const bf16x8 p0, const bf16x8 p1,
                                       const bf16x8 p2, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, p0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, p2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, p0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, p1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, p0, acc, 0, 0, 0);
    return acc;
}

// Workgroup barrier for LDS hand-overs inside the tile loops: waits for this wave's LDS operations only.  __syncthreads() also drains
// vmcnt -- every global load requested ahead (the next tile's operands, the residual, the gate bytes) and, on gfx950, every STORE of the
// tile just finished would be waited for at each barrier: two to three exposed HBM round trips per tile (cdna_hip_programming.md section
// 5, "Pipelining across barriers").  No wave reads global memory another wave of the launch wrote: nothing else needs the drain.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Buffer descriptor of image `img` of a tensor of `img_bytes` per image (NULL tensor: no records -- loads give zero, stores are dropped).
// Every per-pixel global access of the tile loops goes through one with a 32-bit byte offset, a pixel that does not exist being the
// out-of-range offset: NO branch around a memory operation.  (With the accesses inside `if (inside)` blocks the compiler's wait-count
// insertion joined the paths conservatively and waited for the NEXT tile's operands at the first product of the current one -- one exposed
// HBM round trip per tile and wave.)
constexpr int OOB = (int)0x80000000;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t img_rsrc(const void* base, const int img, const int64_t img_bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base) + (uint64_t)(int64_t)img * (uint64_t)img_bytes;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(base != nullptr ? (int)img_bytes : 0), 0x00020000);
}
__device__ __forceinline__ f32x4 ld_f4(const __amdgpu_buffer_rsrc_t r, const int off) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return f32x4{__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3])};
}
__device__ __forceinline__ void st_f4(const f32x4 v, const __amdgpu_buffer_rsrc_t r, const int off) {
    __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, r, off, 0, 0);
}

constexpr int C6 = 64, C7 = 32;        // channels of X6 and X7 (models.py:167-168)
constexpr int RY = 8, RX = 16;         // X6 pixels of a tile
constexpr int TY = 2 * RY, TX = 2 * RX;  // X7 pixels of a tile (16 x 32), of which the interior 14 x 30 is owned
constexpr int OY = TY - 2, OX = TX - 2;
constexpr int W_BYTES = 3 * 128 * 128;   // weight planes: [plane][128 rows][64 bf16]
constexpr int PLANE = TY * TX * 16;      // X7 tile, chunk-planar: [8 chunks of 4 channels][512 pixels][16 B]
constexpr int T_BYTES = 8 * PLANE;

// stage [3][128][64] bf16 weight planes (rows of 128 B) into LDS; 16-byte chunk c of row r sits at chunk c ^ ((r >> 1) & 7)
__device__ __forceinline__ void stage_planes(const uint16_t* w, unsigned char* dst, const int wave, const int lane) {
    const uint64_t addr = reinterpret_cast<uint64_t>(w);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)addr);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, W_BYTES, 0x00020000);
    for (int piece = wave; piece < W_BYTES / 1024; piece += 8) {   // 8 rows per piece
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + piece * 1024), 16, r * 128 + c * 16, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward tail
// 16 waves per workgroup (one workgroup per CU): phase 1 gives wave w the X6 row w & 7 and the N half w >> 3 (four of the eight
// 16-channel blocks); phase 2 gives it 64 pixels and ONE half of the 32 channels (waves 0-6: channels 0-15, waves 7-13:
// 16-31; the weights stay wave-uniform = scalar loads), the halves meet through LDS.  Twice the waves of the 8-wave form hide the
// scalar-load and LDS latencies of the VALU phase.
constexpr int FWD_WAVES = 16;
constexpr int PW2 = (OY * TX + 63) / 64;            // waves per channel half in phase 2 (7: 14 rows of 32 lanes, 30 of them pixels)
constexpr int RED_BYTES = PW2 * 64 * 16;            // partial sums of the second half
constexpr int SCR_BYTES = 3 * 34 * 16;              // fp16 storage, phase 2: a wave's [dx][34 columns][4 floats]

// fp16 storage (round 5): the X7 tile lives in LDS as fp16 -- the value a separate transConv2 launch of this mode would store -- in
// chunks of 8 channels (32 KB instead of 64), conv6's weights come rounded to fp16 like every other layer's of the mode (`w6` then
// points at [3][9][32] fp16) and its taps run on v_dot2_f32_f16 (two channels per instruction, fp32 accumulation: as many
// instructions as the packed fp32 FMAs, half the LDS reads): 55 KB and 64 VGPRs -- TWO workgroups per compute unit.
template <typename T6>   // storage type of X6: float, or _Float16 in fp16-storage mode
__global__ __launch_bounds__(64 * FWD_WAVES, sizeof(T6) == 2 ? 8 : 1) void shading_tail_fwd_kernel(
    const T6* __restrict__ x6, const uint16_t* __restrict__ w2s, const float* __restrict__ bias2, const float* __restrict__ w6,
    const float* __restrict__ bias6, const float* __restrict__ r1, float* __restrict__ y, float* __restrict__ ypre,
    uint8_t* __restrict__ mask7, uint8_t* __restrict__ gate_y, const int B, const int H2, const int W2, const int tiles_y, const int tiles_x) {
    // (`ypre` may be NULL and `gate_y` set: the clamp gate 0 < pre <= 1 of the three output channels as ONE byte per pixel -- what the
    // backward head reads instead of the 16-byte pre-clamp pixel: 63 MB less written here and read there per batch-64 pass)
    // fp16 storage (configs[4]): X6's fp16 values and the weights ROUNDED TO fp16 -- `w2s` is then ONE [128][64] fp16 matrix, as the
    // `w_half` of every other layer of this mode -- on v_mfma_f32_16x16x32_f16: one MFMA per product and no operand split (the
    // bf16x6 form spent 48 MFMAs + 88 split instructions per wave and tile on operands that carry 11 bits)
    constexpr bool H16 = sizeof(T6) == 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* wl = smem;                       // weight planes
    unsigned char* tl = smem + (H16 ? W_BYTES / 3 : W_BYTES);             // X7 tile (fp16 storage: four planes of 8 channels)
    unsigned char* rl = tl + (H16 ? T_BYTES / 2 : T_BYTES);               // phase-2 partial sums (fp16 storage: a 256-byte pad)
    unsigned char* wa = rl + (H16 ? 256 : RED_BYTES);                     // fp16 storage: conv6's weights as MFMA A operands, see below
    unsigned char* scr = wa + 3 * 64 * 16;                                // fp16 storage: per-wave scratch of the shifted partial sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = 2 * H2, W = 2 * W2;
    if constexpr (H16) {
        // round 6: conv6 (3 x 3, 32 -> 3) on v_mfma_f32_16x16x32_f16 too.  K = the 32 channels, columns = 16 pixels of an X7 row, ROWS =
        // (tap column dx, output channel n): row 4 dx + n holds conv6's weights of tap (dy, dx) -- ONE MFMA per tap ROW dy gives, for
        // all three dx at once, P_dx[n][xc] = sum_c w[n][dy][dx][c] X7[c][y + dy][xc] at the UNSHIFTED pixel xc; the accumulator sums the
        // three dy, and out[n][ox] = P_0[n][ox] + P_1[n][ox + 1] + P_2[n][ox + 2] is formed through a per-wave LDS scratch.  Three
        // operand reads + three MFMAs per 16 pixels (a tap per MFMA took nine + nine: the kernel sat on the LDS pipe).  The per-lane
        // operand image [dy][lane][8 fp16] (lane = (row, 8-channel chunk)) is built once per workgroup: 3 KB
        if (tid < 3 * 64) {
            const int dy = tid >> 6, l = tid & 63, r = l & 15, gg = l >> 4;
            const int dx = r >> 2, n = r & 3;
            h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (dx < 3 && n < 3) v = *reinterpret_cast<const h8*>(reinterpret_cast<const _Float16*>(w6) + (n * 9 + 3 * dy + dx) * C7 + 8 * gg);
            *reinterpret_cast<h8*>(wa + tid * 16) = v;
        }
    }
    {   // weight planes (rows of 128 B; chunk c of row r at chunk c ^ ((r >> 1) & 7)): 48 pieces over 16 waves
        const uint64_t addr = reinterpret_cast<uint64_t>(w2s);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)addr);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, W_BYTES, 0x00020000);
        // (fp16 storage: ONE plane of fp16 weights, same rows and swizzle; the descriptor still covers three planes' worth of bytes, of
        // which the first third is read)
        for (int piece = wave; piece < (H16 ? W_BYTES / 3 : W_BYTES) / 1024; piece += FWD_WAVES) {
            const int r = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(wl + piece * 1024), 16, r * 128 + c * 16, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    const int rx = lane & 15, g = lane >> 4;
    const int row = wave & 7, nh = wave >> 3;       // phase 1: X6 row and N half of this wave
    const int ph = wave >= PW2 ? 1 : 0;             // phase 2: channel half (waves 14, 15: idle)
    // ... and pixel of the owned interior, enumerated over 14 rows of 32 (columns 30, 31 idle): a wave = two whole tile rows, so the
    // 16-lane groups of its ds_read_b128 cover 16 consecutive 16-byte slots (rows of 30 wrapped inside the groups: 35 % of the
    // kernel's LDS cycles were bank conflicts, profiles/r04_pmc_tail.txt)
    const int pix = (wave - ph * PW2) * 64 + lane;
    const int oy_r = pix / TX, ox_r = pix - oy_r * TX;
    const bool p2 = wave < 2 * PW2 && oy_r < OY && ox_r < OX;
    const int oy_l = p2 ? oy_r : 0, ox_l = p2 ? ox_r : 0;
    const unsigned char* p2base = tl + (H16 ? 2 : 4) * ph * PLANE + (oy_l * TX + ox_l) * 16;
    // per-lane constants of the phase-1 epilogue: LDS address and bias of its four 16-channel blocks
    int wofs[4];
    f32x4 bq[4];
#pragma unroll
    for (int nq = 0; nq < 4; ++nq) {
        const int nb = 4 * nh + nq, par = nb >> 1, c0 = 16 * (nb & 1) + 4 * g;
        // (fp16 tile: the quad c0 .. c0 + 3 is half (c0 >> 2) & 1 of the 8-channel chunk c0 >> 3)
        wofs[nq] = H16 ? (c0 >> 3) * PLANE + ((2 * row + (par >> 1)) * TX + 2 * rx + (par & 1)) * 16 + 8 * ((c0 >> 2) & 1)
                       : (c0 >> 2) * PLANE + ((2 * row + (par >> 1)) * TX + 2 * rx + (par & 1)) * 16;
        bq[nq] = *reinterpret_cast<const f32x4*>(bias2 + c0);
    }
    const int ntiles = B * tiles_y * tiles_x;
    // X6 operands of a tile (this wave's row; lane = (column, 8-channel chunk)), loaded one tile ahead
    f32x4 xin[4];
    h8 xh[2];      // (fp16 storage: the two 8-channel operands as they are)
    auto load_x6 = [&](const int tile_) {
        const int tx_ = tile_ % tiles_x, ty_ = (tile_ / tiles_x) % tiles_y, img = tile_ / (tiles_x * tiles_y);
        const int ay = (OY / 2) * ty_ - 1 + row, ax = (OX / 2) * tx_ - 1 + rx;
        const bool in6 = tile_ < ntiles && (unsigned)ay < (unsigned)H2 && (unsigned)ax < (unsigned)W2;
        const auto rs = img_rsrc(x6, tile_ < ntiles ? img : 0, (int64_t)H2 * W2 * C6 * (int64_t)sizeof(T6));
        const int off = in6 ? ((ay * W2 + ax) * C6 + g * 8) * (int)sizeof(T6) : OOB;   // (a pixel outside the image: zeros)
        if constexpr (H16) {
            xh[0] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            xh[1] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 64, 0, 0));
        } else {
            xin[0] = ld_f4(rs, off);
            xin[1] = ld_f4(rs, off + 16);
            xin[2] = ld_f4(rs, off + 128);
            xin[3] = ld_f4(rs, off + 144);
        }
    };
    load_x6(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx_ = tile % tiles_x, ty_ = (tile / tiles_x) % tiles_y, img = tile / (tiles_x * tiles_y);
        // owned output rows [OY ty - 1, OY ty + OY - 1), X7 rows [OY ty - 2, + TY), X6 rows [(OY/2) ty - 1, + RY)
        const int a0 = (OY / 2) * ty_ - 1, b0 = (OX / 2) * tx_ - 1;   // first X6 row / column of the tile
        const bool interior = a0 >= 0 && 2 * a0 + TY <= H && b0 >= 0 && 2 * b0 + TX <= W;
        const int64_t HW = (int64_t)H * W;     // (the image's descriptors are built where they are used: 4 SGPRs each, live for a few instructions)
        // ---- phase 1: X7 tile = relu(transConv2(X6) + bias); lane = (X6 column, 8-channel chunk)
        {
            bf16x8 pf[2][3];
            h8 ph[2];
            if constexpr (H16) {
                ph[0] = xh[0];
                ph[1] = xh[1];
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) split8(xin[2 * s], xin[2 * s + 1], pf[s][0], pf[s][1], pf[s][2]);
            }
            load_x6(tile + gridDim.x);   // the next tile's operands fly during this tile's arithmetic
#pragma unroll
            for (int nq = 0; nq < 4; ++nq) {
                const int nb = 4 * nh + nq;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // A operand: row n = 16 nb + (lane & 15), k chunk 4 s + (lane >> 4)
                    const int r = 16 * nb + rx;
                    const unsigned char* wp = wl + r * 128 + (((4 * s + g) ^ ((r >> 1) & 7)) << 4);
                    if constexpr (H16) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8*>(wp), ph[s], acc, 0, 0, 0);
                    } else {
                        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wp);
                        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wp + 128 * 128);
                        const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wp + 2 * 128 * 128);
                        acc = mfma6(w0, w1, w2, pf[s][0], pf[s][1], pf[s][2], acc);
                    }
                }
                // D: column = X6 pixel (lane & 15), rows n = 16 nb + 4 g + e: parity nb >> 1, channels 16 (nb & 1) + 4 g + e
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[e] + bq[nq][e], 0.f);
                if (!interior) {   // (uniform) a tile at the image border: pixels outside are conv6's zero padding
                    const int par = nb >> 1;
                    const int gy = 2 * a0 + 2 * row + (par >> 1), gx = 2 * b0 + 2 * rx + (par & 1);
                    if (!((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)) v = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if constexpr (H16) *reinterpret_cast<h4*>(tl + wofs[nq]) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                else *reinterpret_cast<f32x4*>(tl + wofs[nq]) = v;
            }
        }
        lds_barrier();
        // ---- phase 2: conv6 (3x3, 32 -> 3) over this wave's 16 channels; X7's gate bytes from the centre tap (relu(t) > 0 <=> t > 0)
        if constexpr (H16) {
            // wave w < 14 owns interior row w: two groups of 16 X7 columns (0..15 and 16..31); lane = (column of the group, 8-channel
            // chunk = plane of the tile) for the operands, (column, dx) for the accumulators (rows 4 dx + n).
            if (wave < OY) {
                const int oy = wave;
                f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
                const unsigned char* pb = tl + g * PLANE + (oy * TX + rx) * 16;
                const int ox = 16 * g + rx;                    // the output pixel this lane finishes (g < 2)
                const int gy = 2 * a0 + 1 + oy, gx = 2 * b0 + 1 + ox;
                const bool ok = g < 2 && ox < OX && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const int oi = gy * W + gx;                    // pixel index inside the image
                const f32x4 rv = ld_f4(img_rsrc(r1, img, HW * 16), ok ? oi * 16 : OOB);   // (requested early: used after the taps)
                // (one tap row in flight per wave: 64 registers = eight waves per SIMD hide the LDS latency between them)
#pragma unroll 1
                for (int dy = 0; dy < 3; ++dy) {
                    const h8 a = *reinterpret_cast<const h8*>(wa + (dy * 64 + lane) * 16);
                    const unsigned char* pp = pb + dy * TX * 16;
                    const h8 bA = *reinterpret_cast<const h8*>(pp), bB = *reinterpret_cast<const h8*>(pp + 16 * 16);
                    accA = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bA, accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bB, accB, 0, 0, 0);
                }
                // lane (xc, dx = g) holds P_dx[n = 0..2][xc] of both groups: through the wave's scratch [dx][34 columns][4 floats], then
                // lane ox (< 30) adds its three shifted partial sums (LDS operations of one wave execute in order: no barrier)
                unsigned char* const sc = scr + wave * SCR_BYTES;
                if (g < 3) {
                    *reinterpret_cast<f32x4*>(sc + (g * 34 + rx) * 16) = accA;
                    *reinterpret_cast<f32x4*>(sc + (g * 34 + 16 + rx) * 16) = accB;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                f32x4 av = {0.f, 0.f, 0.f, 0.f};
                if (g < 2) {
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(sc + (0 * 34 + ox) * 16);
                    const f32x4 p1 = *reinterpret_cast<const f32x4*>(sc + (1 * 34 + ox + 1) * 16);
                    const f32x4 p2 = *reinterpret_cast<const f32x4*>(sc + (2 * 34 + ox + 2) * 16);
                    av = p0 + p1 + p2;
                }
                const h8 cA = *reinterpret_cast<const h8*>(pb + (TX + 1) * 16), cB = *reinterpret_cast<const h8*>(pb + (TX + 1 + 16) * 16);
                // X7's gate bytes from the centre tap (relu(t) > 0 <=> t > 0): this lane holds 8 channels = two bytes of each group's pixel
                {
                    unsigned int bitsA = 0, bitsB = 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bitsA |= (cA[e] > (_Float16)0 ? 1u : 0u) << (8 * (e >> 2) + (e & 3));
                        bitsB |= (cB[e] > (_Float16)0 ? 1u : 0u) << (8 * (e >> 2) + (e & 3));
                    }
                    const int gxa = 2 * b0 + 1 + rx, gxb = gxa + 16;
                    const bool yok = gy >= 0 && gy < H;
                    const bool okA = yok && gxa >= 0 && gxa < W, okB = yok && rx + 16 < OX && gxb >= 0 && gxb < W;
                    const auto r_m7 = img_rsrc(mask7, img, HW * (C7 / 4));
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bitsA, r_m7, okA ? (gy * W + gxa) * (C7 / 4) + 2 * g : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)bitsB, r_m7, okB ? (gy * W + gxb) * (C7 / 4) + 2 * g : OOB, 0, 0);
                }
                {
                    f32x4 outv = {0.f, 0.f, 0.f, 0.f}, prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        const float t = fmaxf(av[n] + bias6[n] + rv[n], 0.f);
                        prev[n] = t;
                        outv[n] = fminf(t, 1.f);
                    }
                    st_f4(outv, img_rsrc(y, img, HW * 16), ok ? oi * 16 : OOB);
                    st_f4(prev, img_rsrc(ypre, img, HW * 16), ok ? oi * 16 : OOB);                 // (no `ypre`: no records, dropped)
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((prev[0] > 0.f && prev[0] <= 1.f ? 1u : 0u) | (prev[1] > 0.f && prev[1] <= 1.f ? 2u : 0u) |
                                                                         (prev[2] > 0.f && prev[2] <= 1.f ? 4u : 0u)),
                                                         img_rsrc(gate_y, img, HW), ok ? oi : OOB, 0, 0);   // (no `gate_y`: dropped)
                }
            }
            lds_barrier();   // the tile is free for the next one
            continue;
        }
        f2 acc[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        const int gy = 2 * a0 + 1 + oy_l, gx = 2 * b0 + 1 + ox_l;
        const bool ok = p2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int oi = gy * W + gx;                    // pixel index inside the image
        const f32x4 rv = ld_f4(img_rsrc(r1, img, HW * 16), (ok && ph == 0) ? oi * 16 : OOB);   // (requested early: used after the taps)
        if (p2) {
            unsigned int bits = 0;
#pragma unroll 1
            for (int t = 0; t < 9; ++t) {   // (not unrolled: 48 weight SGPRs per tap is what the scalar file holds)
                // (chunk-planar tile: the four reads of a tap are one address + immediates, neighbouring lanes 16 bytes apart)
                const unsigned char* pp = p2base + ((t / 3) * TX + t % 3) * 16;
                f16v w[3];                  // 16 channels x 3 outputs of this tap: one s_load_dwordx16 per output channel
#pragma unroll
                for (int n = 0; n < 3; ++n) w[n] = *(cf16_ptr)(uintptr_t)(w6 + (n * 9 + t) * C7 + 16 * ph);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(pp + u * PLANE);
                    const f2 a01 = {a[0], a[1]}, a23 = {a[2], a[3]};
#pragma unroll
                    for (int n = 0; n < 3; ++n) {
                        const f2 w01 = {w[n][4 * u], w[n][4 * u + 1]}, w23 = {w[n][4 * u + 2], w[n][4 * u + 3]};
                        acc[n] = __builtin_elementwise_fma(a01, w01, acc[n]);
                        acc[n] = __builtin_elementwise_fma(a23, w23, acc[n]);
                    }
                }
            }
            // X7's gate bytes of this pixel (its own four reads of the centre tap: the tap loop above stays branch-free)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p2base + (TX + 1) * 16 + u * PLANE);
                bits |= ((a[0] > 0.f ? 1u : 0u) | (a[1] > 0.f ? 2u : 0u) | (a[2] > 0.f ? 4u : 0u) | (a[3] > 0.f ? 8u : 0u)) << (8 * u);
            }
            __builtin_amdgcn_raw_buffer_store_b32(bits, img_rsrc(mask7, img, HW * (C7 / 4)), ok ? oi * (C7 / 4) + 4 * ph : OOB, 0, 0);
            if (ph == 1) *reinterpret_cast<f32x4*>(rl + pix * 16) = f32x4{acc[0][0] + acc[0][1], acc[1][0] + acc[1][1], acc[2][0] + acc[2][1], 0.f};
        }
        lds_barrier();   // the second half's sums are in LDS; the tile is free for the next one
        if (ph == 0) {     // (wave-uniform)
            const f32x4 other = *reinterpret_cast<const f32x4*>(rl + (p2 ? pix : 0) * 16);
            f32x4 outv = {0.f, 0.f, 0.f, 0.f}, prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                const float t = fmaxf((acc[n][0] + acc[n][1]) + other[n] + bias6[n] + rv[n], 0.f);
                prev[n] = t;
                outv[n] = fminf(t, 1.f);
            }
            st_f4(outv, img_rsrc(y, img, HW * 16), ok ? oi * 16 : OOB);
            st_f4(prev, img_rsrc(ypre, img, HW * 16), ok ? oi * 16 : OOB);                 // (no `ypre`: no records, dropped)
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)((prev[0] > 0.f && prev[0] <= 1.f ? 1u : 0u) | (prev[1] > 0.f && prev[1] <= 1.f ? 2u : 0u) |
                                                                 (prev[2] > 0.f && prev[2] <= 1.f ? 4u : 0u)),
                                                 img_rsrc(gate_y, img, HW), ok ? oi : OOB, 0, 0);   // (no `gate_y`: dropped)
        }
        // (the next tile's phase 1 writes `tl` only; `rl` is rewritten after its barrier: no third barrier needed, since the
        // first-half waves read `rl` before they reach that barrier)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward head:  P7 = gate7 . conv6^T(gP)  (3 -> 32, 3x3),  P6 = gate6 . transConv2^T(P7)  (K = 4 parities x 32, N = 64)
// tile = 8 x 16 pixels of X6 = 16 x 32 pixels of X7 (no halo on P7; the gP tile carries a 1-pixel halo).
// LDS: weight planes [3][128 rows: 64 used][128 bf16]... see stage below; P7 tile as the GEMM's pixel operand:
//   row = X6 pixel m = 16 ry + rx (128 rows), 128 fp32 columns k = 32 parity + c  (512 B per row)
constexpr int GP_W = TX + 2, GP_H = TY + 2;   // gP tile with halo: 18 x 34 pixels x 16 B
constexpr int WB_BYTES = 3 * 64 * 256;        // backward weight planes: [plane][64 rows n][128 bf16 k] (256 B per row)
constexpr int P7_BYTES = 128 * 512;           // 64 KB
constexpr int GP_BYTES = GP_W * GP_H * 16;    // 9792 B

// 16 waves per workgroup: phase 1 gives wave w the X7 row w of the tile (two groups of 16 pixels; until round 5 a thread owned one
// pixel and half of P7's channels and formed them with 216 packed FMAs); phase 2 gives wave w the X6 row w & 7 and the N half w >> 3
// (two of the four 16-channel blocks of P6).
// fp16 storage (round 5): P7 lives in LDS as the fp16 it is multiplied as (the rounding used to happen when a fragment was read: the same
// values, 32 KB instead of 64), the weights are one 16 KB plane: 58 KB and 64 VGPRs -- TWO workgroups per compute unit, one's VALU phase
// and barriers under the other's matrix-core phase (the fp32 form needs 122 KB: one).
template <typename T6>   // storage type of P6 (float / _Float16)
__global__ __launch_bounds__(1024, sizeof(T6) == 2 ? 8 : 1) void shading_head_bwd_kernel(const float* __restrict__ gp, const float* __restrict__ gcol,
                                                                   const int32_t* __restrict__ state, const float* __restrict__ ypre,
                                                                   const float* __restrict__ w6t,
                                                                   const uint16_t* __restrict__ w2ts,
                                                                   const uint8_t* __restrict__ mask7,
                                                                   const uint8_t* __restrict__ mask6, T6* __restrict__ p6,
                                                                   const uint8_t* __restrict__ gate_y,
                                                                   const int B, const int H2, const int W2, const int tiles_y,
                                                                   const int tiles_x) {
    constexpr bool H16 = sizeof(T6) == 2;   // fp16 storage: `w2ts` = ONE [64][128] fp16 matrix, P7 rounded to fp16 as the operand of
                                            // v_mfma_f32_16x16x32_f16 (the separate launches store P7 as fp16 in HBM: the same rounding)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* wl = smem;                       // weight planes
    unsigned char* pl = smem + (H16 ? WB_BYTES / 3 : WB_BYTES);            // P7 tile (fp16 storage: 128 rows of 256 B)
    unsigned char* gl = pl + (H16 ? P7_BYTES / 2 : P7_BYTES);               // gP tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = 2 * H2, W = 2 * W2;
    {   // stage [3][64][128] bf16 planes: rows of 256 B, 16 chunks; chunk c of row r at chunk c ^ (r & 15)
        const uint64_t addr = reinterpret_cast<uint64_t>(w2ts);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)addr);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, WB_BYTES, 0x00020000);
        for (int piece = wave; piece < (H16 ? WB_BYTES / 3 : WB_BYTES) / 1024; piece += 16) {   // 4 rows per piece
            const int r = piece * 4 + (lane >> 4);
            const int c = (lane & 15) ^ (r & 15);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(wl + piece * 1024), 16, r * 256 + c * 16, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const int rx = lane & 15, g = lane >> 4;
    const int row = wave & 7, nh = wave >> 3;        // phase 2: X6 row and N half
    // (phase 1: wave = X7 row of the 16 x 32 tile, two groups of 16 pixels; the GEMM operand row of X7 pixel (py, px) is
    // m = 16 (py >> 1) + (px >> 1), its columns k = 32 (2 (py & 1) + (px & 1)) + channel)
    const int ntiles = B * tiles_y * tiles_x;
    // round 6: conv6^T (3 x 3, 3 -> 32) on the matrix cores (fp16 storage: v_mfma_f32_16x16x32_f16): the 27 products of an X7 pixel are ONE K = 32 step --
    // k = 8 g + j: j < 3 tap 2 g, channel j; 4 <= j < 7 tap 2 g + 1, channel j - 4; the pad slots of the 4-float pixels carry tap 8
    // (k = 3, 7, 11: its channels 0, 1, 2), the rest zero.  Rows = P7's channels (two blocks of 16), columns = 16 pixels of a tile row.
    // Weights: conv6's, rounded to fp16 like the forward kernel's (`w6t` holds the fp32 values, transposed and mirrored); the cotangent
    // gP enters as hi + lo fp16 halves (two MFMAs): exact to 2^-22, the fp32 FMAs' products to rounding.
    // fp32 storage: the same K = 32 step with both operands split exactly into three bf16 planes, six of the nine partial products
    // (the arithmetic of every bf16x6 kernel of this library: products exact to 2^-24 of their operands, fp32 accumulation) -- 12 MFMAs
    // per 16 pixels where the VALU form spent 216 packed FMAs per lane.
    h8 a6[2];
    bf16x8 a6b[2][3];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int c = 16 * nb + rx;
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int tap = 2 * g + (j >> 2), o = j & 3;
            if (o == 3) {   // pad slot
                const int k = 8 * g + j;
                tap = 8;
                o = k == 3 ? 0 : (k == 7 ? 1 : (k == 11 ? 2 : 3));
            }
            wv[j] = (o < 3 && tap < 9) ? w6t[(3 * tap + o) * C7 + c] : 0.f;
        }
        if constexpr (H16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a6[nb][j] = (_Float16)wv[j];
        } else {
            split8(f32x4{wv[0], wv[1], wv[2], wv[3]}, f32x4{wv[4], wv[5], wv[6], wv[7]}, a6b[nb][0], a6b[nb][1], a6b[nb][2]);
        }
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx_ = tile % tiles_x, ty_ = (tile / tiles_x) % tiles_y, img = tile / (tiles_x * tiles_y);
        const int a0 = RY * ty_, b0 = RX * tx_;   // first X6 row / column
        // the 8 gate bytes of this lane's X6 pixel's channel half (phase 2), requested HERE -- ahead of both barriers, through the image's
        // buffer descriptor (a pixel outside = the out-of-range offset = zeros: no branch, the wait stays counted)
        const int ay = a0 + row, ax = b0 + rx;
        const bool in6 = ay < H2 && ax < W2;
        const auto g6u = __builtin_amdgcn_raw_buffer_load_b64(img_rsrc(mask6, img, (int64_t)H2 * W2 * (C6 / 4)), in6 ? (ay * W2 + ax) * (C6 / 4) + 8 * nh : OOB, 0, 0);
        // ---- phase 0: gP tile with halo (out-of-image pixels: zeros = conv6's zero padding seen from the gradient)
        if (tid < GP_W * GP_H) {
            const int qy = tid / GP_W, qx = tid - qy * GP_W;
            const int gy = 2 * a0 - 1 + qy, gx = 2 * b0 - 1 + qx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                const size_t o = (((size_t)img * H + gy) * W + gx) * 4;
                if (state != nullptr) {
                    // spaa_select_grad folded in (projector_based_attack.py:302,310: the sample's cotangent is the classifier
                    // path's gradient `gp` or the stealth loss's `gcol`; models.py:301: backward of clamp(relu(.), max = 1))
                    v = *reinterpret_cast<const f32x4*>((state[4 * img + 1] != 0 ? gcol : gp) + o);
                    if (gate_y != nullptr) {      // the clamp gate as the forward tail's byte (bit e: 0 < pre_e <= 1)
                        const unsigned int gb = gate_y[o >> 2];
#pragma unroll
                        for (int e = 0; e < 3; ++e) v[e] = ((gb >> e) & 1u) ? v[e] : 0.f;
                    } else {
                        const f32x4 y = *reinterpret_cast<const f32x4*>(ypre + o);
#pragma unroll
                        for (int e = 0; e < 3; ++e) v[e] = (y[e] > 0.f && y[e] <= 1.f) ? v[e] : 0.f;
                    }
                    v[3] = 0.f;
                } else {
                    v = *reinterpret_cast<const f32x4*>(gp + o);
                }
            }
            *reinterpret_cast<f32x4*>(gl + tid * 16) = v;
        }
        // the gate bytes of this lane's pixels (requested before the barrier)
        uint64_t mb16[2] = {0, 0};   // (wave = tile row, two groups of 16 pixels: all 8 gate bytes of this lane's pixel of each)
        {
            const int gyw = 2 * a0 + wave;
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                const int gxw = 2 * b0 + 16 * grp + rx;
                if (gyw < H && gxw < W) mb16[grp] = *reinterpret_cast<const uint64_t*>(mask7 + (((size_t)img * H + gyw) * W + gxw) * (C7 / 4));
            }
        }
        lds_barrier();
        // ---- phase 1: P7[pixel][16 nh ..] = gate7 . sum_{taps, 3 ch} gP[pixel + d] w6t[tap][ch][16 nh ..]
        {
            const int pyw = wave;
#pragma unroll
            for (int grp = 0; grp < 2; ++grp) {
                const int pxl = 16 * grp + rx;
                // this lane's 8 of the pixel's 32 K slots: taps 2 g and 2 g + 1 (g = 3: tap 6, 7), tap 8 in the pad slots of g = 0, 1
                const int t0 = 2 * g, t1 = 2 * g + 1;
                f32x4 v0 = *reinterpret_cast<const f32x4*>(gl + ((pyw + t0 / 3) * GP_W + pxl + t0 % 3) * 16);
                f32x4 v1 = *reinterpret_cast<const f32x4*>(gl + ((pyw + t1 / 3) * GP_W + pxl + t1 % 3) * 16);
                const f32x4 v8 = *reinterpret_cast<const f32x4*>(gl + ((pyw + 2) * GP_W + pxl + 2) * 16);
                v0[3] = g == 0 ? v8[0] : (g == 1 ? v8[2] : 0.f);
                v1[3] = g == 0 ? v8[1] : 0.f;
                h8 bh, bl;
                bf16x8 bp[3];
                if constexpr (H16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bh[e] = (_Float16)v0[e];
                        bh[4 + e] = (_Float16)v1[e];
                        bl[e] = (_Float16)(v0[e] - (float)bh[e]);
                        bl[4 + e] = (_Float16)(v1[e] - (float)bh[4 + e]);
                    }
                } else {
                    split8(v0, v1, bp[0], bp[1], bp[2]);
                }
                const int m1w = 16 * (pyw >> 1) + (pxl >> 1), par = 2 * (pyw & 1) + (pxl & 1);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (H16) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a6[nb], bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a6[nb], bh, acc, 0, 0, 0);
                    } else {
                        acc = mfma6(a6b[nb][0], a6b[nb][1], a6b[nb][2], bp[0], bp[1], bp[2], acc);
                    }
                    // D: column = pixel rx, rows = channels 16 nb + 4 g + e: gate byte 4 nb + g of the pixel
                    const unsigned int nib = (unsigned int)(mb16[grp] >> (8 * (4 * nb + g))) & 15u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = ((nib >> e) & 1u) ? acc[e] : 0.f;
                    if constexpr (H16) {
                        // fp16 row m1w of 16 chunks of 8 values: this quad (k = 32 par + 16 nb + 4 g ..) is half g & 1 of chunk
                        // 4 par + 2 nb + (g >> 1), stored at chunk ^ (m1w & 15)
                        *reinterpret_cast<h4*>(pl + m1w * 256 + (((4 * par + 2 * nb + (g >> 1)) ^ (m1w & 15)) << 4) + 8 * (g & 1)) =
                            h4{(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
                    } else {
                        // fp32 row m1w of 32 chunks of 4 values: chunk 8 par + 4 nb + g at chunk ^ (m1w & 31)
                        *reinterpret_cast<f32x4*>(pl + m1w * 512 + (((8 * par + 4 * nb + g) ^ (m1w & 31)) << 4)) = acc;
                    }
                }
            }
        }
        lds_barrier();
        // ---- phase 2: P6[m][n] = gate6 . sum_k P7[m][k] W[n][k];  lane = (X6 column, k chunk)
        {
            const int m = 16 * row + rx;
            bf16x8 pf[4][3];
            h8 ph[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {   // k step s: columns 32 s + 8 g .. + 7 = chunks 8 s + 2 g, + 1 (fp16 rows: chunk 4 s + g)
                if constexpr (H16) {
                    ph[s] = *reinterpret_cast<const h8*>(pl + m * 256 + (((4 * s + g) ^ (m & 15)) << 4));
                } else {
                    const f32x4 u0 = *reinterpret_cast<const f32x4*>(pl + m * 512 + (((8 * s + 2 * g) ^ (m & 31)) << 4));
                    const f32x4 u1 = *reinterpret_cast<const f32x4*>(pl + m * 512 + (((8 * s + 2 * g + 1) ^ (m & 31)) << 4));
                    split8(u0, u1, pf[s][0], pf[s][1], pf[s][2]);
                }
            }
            const uint64_t g6 = (uint64_t)g6u[0] | ((uint64_t)g6u[1] << 32);
#pragma unroll
            for (int nq = 0; nq < 2; ++nq) {
                const int nb = 2 * nh + nq;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int r = 16 * nb + rx;   // weight row n
                    const unsigned char* wp = wl + r * 256 + (((4 * s + g) ^ (r & 15)) << 4);
                    if constexpr (H16) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8*>(wp), ph[s], acc, 0, 0, 0);
                    } else {
                        const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(wp);
                        const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(wp + 64 * 256);
                        const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(wp + 2 * 64 * 256);
                        acc = mfma6(w0, w1, w2, pf[s][0], pf[s][1], pf[s][2], acc);
                    }
                }
                // D: column = X6 pixel (lane & 15), rows n = 16 nb + 4 g + e: byte 4 nq + g of this half's gate bytes
                const int n0 = 16 * nb + 4 * g;
                const unsigned int nib = (unsigned int)(g6 >> (8 * (4 * nq + g))) & 15u;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = ((nib >> e) & 1u) ? acc[e] : 0.f;
                if (in6) {
                    T6* dst = p6 + (((size_t)img * H2 + ay) * W2 + ax) * C6 + n0;
                    if constexpr (sizeof(T6) == 4) *reinterpret_cast<f32x4*>(dst) = acc;
                    else *reinterpret_cast<h4*>(dst) = h4{(_Float16)acc[0], (_Float16)acc[1], (_Float16)acc[2], (_Float16)acc[3]};
                }
            }
        }
        lds_barrier();
    }
}

}  // namespace

template <typename T6>
static int launch_tail_fwd(const T6* x6, const uint16_t* w2_split, const float* bias2, const float* w6, const float* bias6,
                           const float* res1, float* y, float* ypre, uint8_t* mask7, uint8_t* gate_y, int B, int H2, int W2, spaa_stream_t stream_) {
    if (!x6 || !w2_split || !bias2 || !w6 || !bias6 || !res1 || !y || (!ypre && !gate_y) || !mask7 || B <= 0 || H2 <= 0 || W2 <= 0)
        return hipErrorInvalidValue;
    if ((int64_t)B * H2 * W2 * 4 * C7 * 4 >= (int64_t)1 << 40) return hipErrorInvalidValue;
    if ((int64_t)H2 * W2 * C6 * 4 >= (int64_t)1 << 31) return hipErrorInvalidValue;   // (per-image buffer descriptors, 32-bit offsets inside an image)
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int H = 2 * H2, W = 2 * W2;
    const int tiles_y = (H + 1 + OY - 1) / OY, tiles_x = (W + 1 + OX - 1) / OX;   // owned rows start at -1
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    if (ntiles > 0x7fffffff) return hipErrorInvalidValue;
    constexpr bool H16 = sizeof(T6) == 2;
    const size_t smem = H16 ? (size_t)W_BYTES / 3 + T_BYTES / 2 + 256 + 3 * 64 * 16 + OY * SCR_BYTES : (size_t)W_BYTES + T_BYTES + RED_BYTES;
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&shading_tail_fwd_kernel<T6>), (int)smem, attr_set);
    if (e != hipSuccess) return (int)e;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    const int64_t slots = (int64_t)(H16 ? 2 : 1) * ncu;      // (fp16 storage: two resident workgroups per compute unit)
    const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
    hipLaunchKernelGGL(shading_tail_fwd_kernel<T6>, dim3(grid), dim3(64 * FWD_WAVES), smem, stream, x6, w2_split, bias2, w6, bias6, res1,
                       y, ypre, mask7, gate_y, B, H2, W2, tiles_y, tiles_x);
    return (int)hipGetLastError();
}

template <typename T6>
static int launch_head_bwd(const float* gp, const float* gcol, const int32_t* state, const float* ypre, const float* w6t,
                           const uint16_t* w2t_split, const uint8_t* mask7, const uint8_t* mask6,
                           T6* p6, const uint8_t* gate_y, int B, int H2, int W2, spaa_stream_t stream_) {
    if (!gp || !w6t || !w2t_split || !mask7 || !mask6 || !p6 || B <= 0 || H2 <= 0 || W2 <= 0) return hipErrorInvalidValue;
    if (state != nullptr && (!gcol || (!ypre && !gate_y))) return hipErrorInvalidValue;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int tiles_y = (H2 + RY - 1) / RY, tiles_x = (W2 + RX - 1) / RX;
    const int64_t ntiles = (int64_t)B * tiles_y * tiles_x;
    if (ntiles > 0x7fffffff) return hipErrorInvalidValue;
    constexpr bool H16 = sizeof(T6) == 2;
    const size_t smem = H16 ? (size_t)WB_BYTES / 3 + P7_BYTES / 2 + GP_BYTES : (size_t)WB_BYTES + P7_BYTES + GP_BYTES;
    static bool attr_set[SPAA_MAX_DEVICES] = {};
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&shading_head_bwd_kernel<T6>), (int)smem, attr_set);
    if (e != hipSuccess) return (int)e;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 256;
    const int64_t slots = (int64_t)(H16 ? 2 : 1) * ncu;      // (fp16 storage: two resident workgroups per compute unit)
    const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
    hipLaunchKernelGGL(shading_head_bwd_kernel<T6>, dim3(grid), dim3(1024), smem, stream, gp, gcol, state, ypre, w6t, w2t_split, mask7, mask6, p6,
                       gate_y, B, H2, W2, tiles_y, tiles_x);
    return (int)hipGetLastError();
}

extern "C" {

int spaa_shading_tail_fwd(const float* x6, const uint16_t* w2_split, const float* bias2, const float* w6, const float* bias6,
                          const float* res1, float* y, float* ypre, uint8_t* mask7, int B, int H2, int W2, spaa_stream_t stream) {
    return launch_tail_fwd<float>(x6, w2_split, bias2, w6, bias6, res1, y, ypre, mask7, nullptr, B, H2, W2, stream);
}
int spaa_shading_tail_fwd_f16(const void* x6, const void* w2_half, const float* bias2, const void* w6_half, const float* bias6,
                              const float* res1, float* y, float* ypre, uint8_t* mask7, int B, int H2, int W2, spaa_stream_t stream) {
    return launch_tail_fwd<_Float16>(reinterpret_cast<const _Float16*>(x6), reinterpret_cast<const uint16_t*>(w2_half), bias2,
                                     reinterpret_cast<const float*>(w6_half), bias6, res1, y, ypre, mask7, nullptr, B, H2, W2, stream);
}
int spaa_shading_head_bwd(const float* gp, const float* w6t, const uint16_t* w2t_split, const uint8_t* mask7, const uint8_t* mask6,
                          float* p6, int B, int H2, int W2, spaa_stream_t stream) {
    return launch_head_bwd<float>(gp, nullptr, nullptr, nullptr, w6t, w2t_split, mask7, mask6, p6, nullptr, B, H2, W2, stream);
}
int spaa_shading_head_bwd_select(const float* g_adv, const float* g_col, const int32_t* state, const float* ypre, const float* w6t,
                                 const uint16_t* w2t_split, const uint8_t* mask7, const uint8_t* mask6, float* p6, int B, int H2, int W2,
                                 spaa_stream_t stream) {
    if (!state) return hipErrorInvalidValue;
    return launch_head_bwd<float>(g_adv, g_col, state, ypre, w6t, w2t_split, mask7, mask6, p6, nullptr, B, H2, W2, stream);
}
int spaa_shading_head_bwd_select_f16(const float* g_adv, const float* g_col, const int32_t* state, const float* ypre, const float* w6t,
                                     const void* w2t_half, const uint8_t* mask7, const uint8_t* mask6, void* p6, int B, int H2, int W2,
                                     spaa_stream_t stream) {
    if (!state) return hipErrorInvalidValue;
    return launch_head_bwd<_Float16>(g_adv, g_col, state, ypre, w6t, reinterpret_cast<const uint16_t*>(w2t_half), mask7, mask6,
                                     reinterpret_cast<_Float16*>(p6), nullptr, B, H2, W2, stream);
}
int spaa_shading_head_bwd_f16(const float* gp, const float* w6t, const void* w2t_half, const uint8_t* mask7, const uint8_t* mask6,
                              void* p6, int B, int H2, int W2, spaa_stream_t stream) {
    return launch_head_bwd<_Float16>(gp, nullptr, nullptr, nullptr, w6t, reinterpret_cast<const uint16_t*>(w2t_half), mask7, mask6,
                                     reinterpret_cast<_Float16*>(p6), nullptr, B, H2, W2, stream);
}

// round 6: the clamp gate of the network output as ONE byte per pixel (`gate_y` [B,2 H2,2 W2], bit e = 0 < pre_e <= 1) instead of the
// 16-byte pre-clamp pixel: the tail writes it (ypre may then be NULL), the select head reads it (ypre is then ignored)
int spaa_shading_tail_fwd_g(const float* x6, const uint16_t* w2_split, const float* bias2, const float* w6, const float* bias6,
                            const float* res1, float* y, float* ypre, uint8_t* mask7, uint8_t* gate_y, int B, int H2, int W2, spaa_stream_t stream) {
    if (!gate_y) return hipErrorInvalidValue;
    return launch_tail_fwd<float>(x6, w2_split, bias2, w6, bias6, res1, y, ypre, mask7, gate_y, B, H2, W2, stream);
}
int spaa_shading_tail_fwd_f16_g(const void* x6, const void* w2_half, const float* bias2, const void* w6_half, const float* bias6,
                                const float* res1, float* y, float* ypre, uint8_t* mask7, uint8_t* gate_y, int B, int H2, int W2,
                                spaa_stream_t stream) {
    if (!gate_y) return hipErrorInvalidValue;
    return launch_tail_fwd<_Float16>(reinterpret_cast<const _Float16*>(x6), reinterpret_cast<const uint16_t*>(w2_half), bias2,
                                     reinterpret_cast<const float*>(w6_half), bias6, res1, y, ypre, mask7, gate_y, B, H2, W2, stream);
}
int spaa_shading_head_bwd_select_g(const float* g_adv, const float* g_col, const int32_t* state, const uint8_t* gate_y, const float* w6t,
                                   const uint16_t* w2t_split, const uint8_t* mask7, const uint8_t* mask6, float* p6, int B, int H2, int W2,
                                   spaa_stream_t stream) {
    if (!state || !gate_y) return hipErrorInvalidValue;
    return launch_head_bwd<float>(g_adv, g_col, state, nullptr, w6t, w2t_split, mask7, mask6, p6, gate_y, B, H2, W2, stream);
}
int spaa_shading_head_bwd_select_f16_g(const float* g_adv, const float* g_col, const int32_t* state, const uint8_t* gate_y, const float* w6t,
                                       const void* w2t_half, const uint8_t* mask7, const uint8_t* mask6, void* p6, int B, int H2, int W2,
                                       spaa_stream_t stream) {
    if (!state || !gate_y) return hipErrorInvalidValue;
    return launch_head_bwd<_Float16>(g_adv, g_col, state, nullptr, w6t, reinterpret_cast<const uint16_t*>(w2t_half), mask7, mask6,
                                     reinterpret_cast<_Float16*>(p6), gate_y, B, H2, W2, stream);
}

}  // extern "C"
