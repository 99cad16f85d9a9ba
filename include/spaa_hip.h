/*
 * spaa_hip.h — C-ABI of libspaa_hip.so: the MI355X (gfx950) kernels behind the SPAA attack loop.
 *
 * The reference (BingyaoHuang/SPAA) is pure Python over ATen; it has no FFI layer.  The drop-in boundary is
 * therefore the set of tensor operations its hot path dispatches (SURVEY.md §2.1 G0–G16, §8b).  Every entry point
 * takes raw device pointers, plain sizes and a hipStream_t, returns 0 on success or a hipError_t value, never
 * allocates, frees or synchronises (graph-capture safe), and names the reference call it replaces
 * (paths relative to /root/reference/src/python).
 *
 * Layout convention: activations are NHWC fp32 with an explicit channel stride (`cstride`, multiple of 4) and
 * channel offset (`coff`, multiple of 4); 3-channel images are stored NHWC with cstride 4 ("NHWC4", pad lane = 0).
 * The reference's NCHW tensors are converted at the Python boundary by spaa_nchw_to_nhwc4 / spaa_nhwc4_to_nchw.
 */
#ifndef SPAA_HIP_H
#define SPAA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* spaa_stream_t; /* hipStream_t */

#define SPAA_MAX_CLASSES 4
#define SPAA_MAX_TAPS 64
#define SPAA_SPLITK_HDR_FLOATS 4096 /* arrival counters at the head of a K-range workspace (spaa_tapconv_t.reserved1 bit 8) */

/* activation applied after bias + residual add */
enum { SPAA_ACT_NONE = 0, SPAA_ACT_RELU = 1, SPAA_ACT_RELU_CLAMP1 = 2, SPAA_ACT_LEAKY01 = 3 };
/* fp16-storage flags (spaa_tapconv_t.io_dtype) */
enum { SPAA_IO_IN_F16 = 1, SPAA_IO_OUT_F16 = 2 };
/* gate applied last (ReLU / clamp backward):  out = pass(gate) ? v : 0 */
enum { SPAA_GATE_NONE = 0, SPAA_GATE_POS = 1 /* gate > 0 */, SPAA_GATE_POS_LE1 = 2 /* 0 < gate <= 1 */,
       SPAA_GATE_MUL = 3 /* value * gate: the chain rule through `x * s` (models.py:342); tiles 25.. only */ };

typedef struct {
    int32_t oy0, ox0;  /* output offset of this parity class */
    int32_t ntaps;     /* taps of this class */
    int32_t tap_off;   /* first tap in `taps` */
    int32_t K;         /* ntaps * Cin */
    int32_t Kpad;      /* K rounded up to 32 */
    int64_t w_off;     /* float offset of this class's packed weights [Npad][Kpad] */
} spaa_tapclass_t;

/*
 * Generic "tap-list" convolution as an implicit GEMM on fp32 MFMA (v_mfma_f32_32x32x2_f32):
 *   out[b, oy0 + s_out*y, ox0 + s_out*x, n] = gate( act( bias[n] + add[...] +
 *        sum_{t<ntaps, c<Cin} in[b, s_in*y + dy_t, s_in*x + dx_t, c] * W[n][t*Cin + c] ) )
 * for y < Hm, x < Wm, zero padding outside the input.  One primitive covers
 *   - nn.Conv2d forward, stride 1/2            (models.py:223-252 conv*, skipConv*; torchvision convs)
 *   - nn.ConvTranspose2d forward, stride 2     (models.py:237-238; = 4 output-parity classes)
 *   - aten::convolution_backward w.r.t. input  (autograd of the above, projector_based_attack.py:302,310)
 * by the choice of taps / weight packing (spaa_amd/convplan.py).
 */
typedef struct {
    const float* in;
    int32_t Hin, Win, Cin, in_cstride, in_coff;
    float* out;
    int32_t Hout, Wout, Cout, out_cstride, out_coff;
    int32_t B, Hm, Wm, s_in, s_out;
    const float* weights; /* packed, see spaa_tapclass_t.w_off; rows padded to a multiple of 128 */
    const uint16_t* w_split; /* optional: the same weights as three bf16 planes per class, [3][Npad][Kpad] with
                                w == h + m + l exactly (tiles 12-14: fp32 emulated on the bf16 matrix cores); class c
                                starts at element 3 * cls[c].w_off */
    const uint16_t* w_half; /* SPAA_IO_IN_F16: the weights rounded to fp16, per class [Npad][Kpad64] (Kpad64 = K rounded up to 64,
                               zero padded); class c starts at element cls[c].w_off / Kpad * Kpad64 */
    const int32_t* taps;  /* device array of (dy, dx) pairs */
    const float* bias;    /* [Cout] or NULL */
    const float* add;     /* residual, indexed like `out`, or NULL */
    int32_t add_cstride, add_coff;
    const float* gate;    /* indexed like `out`, or NULL */
    int32_t gate_cstride, gate_coff, gate_mode;
    int32_t act;
    int32_t tile;         /* 0 = auto; 1..54, 60..65, 68, 70, 71, 72, 73, 74 = explicit kernel / workgroup tile (the dispatcher in tapconv.hip lists them;
                             spaa_amd/convplan.py: TILE_NAMES; chosen per layer shape by tools/autotune.py) */
    float* aux_out;       /* optional second output (indexed like `out`):
                             act == SPAA_ACT_RELU_CLAMP1: the value BEFORE the clamp;
                             otherwise, with gate2: (gate2 > 0) ? out_value : 0  (a second ReLU-backward gate) */
    const float* gate2;
    int32_t gate2_cstride, gate2_coff;
    uint8_t* mask_out;         /* optional ReLU-gate mask of this launch's output: one BYTE per 4 consecutive channels, bit e =
                                  (out[n0 + e] > 0), at index (o * out_cstride + out_coff + n0) >> 2 (o = output pixel).  A
                                  dgrad launch then reads 2 bits per element instead of the 32-bit activation.  Needs Cout,
                                  out_cstride, out_coff % 4 == 0 and a tile of the families 15..27, 30..46, 48..54 */
    const uint8_t* gate_bits;  /* alternative to `gate` (meaning SPAA_GATE_POS): a mask written through `mask_out` by the
                                  launch that produced the activation; indexed with gate_cstride / gate_coff */
    const uint8_t* gate2_bits; /* likewise for `gate2` (gate2_cstride / gate2_coff) */
    int32_t tap_range[4]; /* (dy_min, dy_max, dx_min, dx_max) over the taps of all classes: patch-staged kernels */
    float* splitk_ws;     /* split-K workspace, ksplit * B*Hm*Wm * Npad floats (Npad = Cout rounded up to 128), or NULL */
    int32_t ksplit;       /* 0, 1: off.
                             > 1 (tiles 25.., one class): K is cut into `ksplit` ranges computed by separate workgroups into
                             splitk_ws; a second kernel adds them in fixed order and applies the epilogue (layers with few
                             output pixels and long K, e.g. ResNet layer4: fills the chip).
                             -1 (persistent tiles 48..54, one class): stream-K — the K-steps of all tiles are cut into equal
                             ranges per workgroup; splitk_ws must hold 2 * 768 * 128 * 128 floats; cut tiles are summed in
                             segment order by a second kernel. */
    int32_t nfold;        /* <= 1: off.  4 (tiles 25.., one class, s_out == 2, Cout % 4 == 0): the four output-parity classes of
                             a kernel-2 stride-2 ConvTranspose2d share their single tap, so they are folded into the GEMM N
                             dimension: weight rows [nfold*Cout], row c*Cout + n -> output pixel (2y + c/2, 2x + c%2),
                             channel n.  The input is read once instead of once per class. */
    int32_t reserved0;    /* measurement switches (A/B runs of kernel variants: spaa_amd/convplan.py DEBUG_*); 0 in production,
                           * except bits 27-28 of a Winograd launch (tiles 70 / 71 / 73): the layer's zero padding, 0 = 1 (same-size
                           * output), 1 = 0 (unpadded: output 2 smaller), 2 = 2 (that layer's input gradient: 2 larger) */
    int32_t io_dtype;     /* fp16-STORAGE mode (BASELINE.json configs[4]: "fp16 with fp32 dE2000"), bit flags:
                             SPAA_IO_IN_F16  (tiles 60..65; tile 68: 3x3 / stride-1 layers with the input patch staged once in LDS;
                                             tile 72: thin outputs with the parity classes folded into N, fp32 out (with a
                                             4-channel pixel stride, offset 0 and Cout < 4 its 16-byte stores write ZERO into
                                             the pad channels Cout..3; every other tile leaves channels >= Cout untouched);
                                             tile 29 with an fp32 output of at most 4 channels: the image-side input
                                             gradients): `in` is fp16 NHWC (strides / offsets still in elements) and the
                                             weights come from `w_half`; fp32 accumulation on v_mfma_f32_16x16x32_f16;
                             SPAA_IO_OUT_F16 (tiles 60..65, 68, and the kernels that read fp32 IMAGES: 15..24, 38):
                                             `out`, `add`, `gate`, `aux_out`, `gate2` are fp16.
                             0 = everything fp32 (the default path; dtype "f32" in bench.py). */
    int32_t reserved1;    /* tiles 70 / 71 / 73 (and tile 68's K-range form): bit 8 = `splitk_ws` BEGINS with SPAA_SPLITK_HDR_FLOATS floats of arrival
                             counters (int32, all zero on first use; the kernels leave them zero), the K ranges' partial sums follow: the
                             last-arriving workgroup of a (region, N tile) adds the K ranges in fixed order and applies the epilogue inside the
                             kernel -- the same bits as the separate second pass, one launch less.  The workspace is then SPAA_SPLITK_HDR_FLOATS +
                             K ranges x B x H x W x Npad floats.  Bit 8 clear: the two-pass form on a workspace without header.
                             tile 76: bit 0 = fp16 operands (fp16-storage mode, fp16 output only): the image rounded to fp16 in registers, `w_split` = ONE
                             plane of fp16 weights in the layout of ConvPlan.c3_pack(half=True), products on v_mfma_f32_16x16x32_f16.
                             tile 68: bit 2 = the canvas / K-range form (small images with long K: spaa_tapconv_h16p_plan below; `ksplit` > 1 with
                             `splitk_ws` = that many K ranges), bits 0-1 = its N tile (0 chosen, 1 = 64, 2 = 128), bit 3 (tests) = canvases wherever they
                             have fewer regions, bit 4 (A/B runs) = 64-wide stride-1 layers as ONE workgroup per compute unit, bit 5 = 64-wide N tiles (two
                             workgroups per compute unit) for a wider layer, bit 6 = the layer's ReLU and the 2 x 2 / stride-2 max-pool that follows it in the epilogue
                             (stride 1, unfolded, image-aligned regions, act = ReLU, no residual / gates, even Hout / Wout): `out` is then the POOLED tensor
                             [B, Hout / 2, Wout / 2, out_cstride] and `mask_out` the pool's arg-max bytes [B, Hout / 2, Wout / 2, Cout] in spaa_maxpool_fwd's format
                             (or NULL) -- torchvision VGG-16's conv -> ReLU -> MaxPool2d(2, 2), classifier.py:21-24; bit 7 = the
                             pool-adjoint PROLOGUE of that layer's input gradient (two-workgroup form only): `in` = the gradient w.r.t. the pool's output [B, Hin / 2,
                             Win / 2, in_cstride] fp16, `in2` = the pool's arg-max bytes [B, Hin / 2, Win / 2, in2_cstride] -- spaa_maxpool_bwd with its ReLU gate applied
                             while the patch is staged.  0 otherwise. */
    int32_t nclass;
    spaa_tapclass_t cls[SPAA_MAX_CLASSES];
    /* optional SECOND SOURCE (NULL = none).
     * Tile 74: a 1 x 1 convolution of a tensor at OUTPUT resolution, added to the accumulators before bias / residual / activation
     * -- `transConv1(x) + skipConv2(x1)` (models.py:293,299) and its mirror image in the backward pass as ONE launch:
     * out[b, oy, ox, n] += sum_c in2[b, oy, ox, in2_coff + c] * W2[n][c] (weights `w2_split`, Cin2 = 32 or 64).
     * Tiles 70 / 71 (Winograd): the layer's LAST Cin2 input channels (Cin2 % 32 == 0, 0 < Cin2 < Cin) are read from `in2`
     * ([B, Hin, Win, in2_cstride]) instead of `in` -- conv(a, Wa) + conv(b, Wb) as one convolution over the concatenated channels,
     * `conv5(x4) + skipConv3(x2)` (models.py:294,298); `w2_split` unused (the weights are the layer's own, K = 16 x Cin).
     * Tile 68 (fp16 storage, FOLDED stride-2 transposed layer, nfold = 4, Cout % 16 == 0, fp16 output): the same 1 x 1 convolution
     * of an fp16 tensor at output resolution; `in2` is fp16 and `w2_split` ONE fp16 matrix [Cout][Cin2] (the weights rounded to fp16
     * like `w_half`). */
    const float* in2;         /* [B, Hout, Wout, in2_cstride] */
    int32_t in2_cstride, in2_coff, Cin2;   /* Cin2 = 32 or 64 */
    int32_t reserved2;        /* tile 74: bit 0 = the classes' tap lists have the canonical k3 / s2 order (csrc/tapconv_x6p.hip STD: the caller
                                 has compared them; the launcher re-checks what the descriptor shows: tap counts 1 / 2 / 2 / 4, window (0..1)^2) */
    const uint16_t* w2_split; /* W2 as three bf16 planes [3][Npad][Cin2] with w == h + m + l exactly */
} spaa_tapconv_t;

int spaa_tapconv_f32(const spaa_tapconv_t* desc, spaa_stream_t stream);
/* Weight and bias gradients of the layer `desc` describes in its FORWARD form (same geometry / taps / packing fields;
 * `in` = the layer's input activation; `out_cstride` / `out_coff` / Hout / Wout describe `gout`, the gradient w.r.t. the
 * layer's pre-activation [B,Hout,Wout,out_cstride]); fp32 only, nfold <= 1.
 *   dw_packed [sum_c Npad*Kpad_c]: dW in the layout of `weights` (spaa_tapclass_t.w_off), pad rows / columns zero
 *   dbias [Cout] or NULL:          sum over output pixels of gout
 *   workspace: nchunk * max(sum_c Npad*Kpad_c, Cout) floats; the pixel range is cut into `nchunk` parts that are summed in
 *   a fixed order (deterministic).  Replaces aten::convolution_backward(weight, bias) in `train_loss_batch.backward()`
 *   (train_network.py:316). */
int spaa_tapconv_wgrad(const spaa_tapconv_t* desc, const float* gout, float* dw_packed, float* dbias, float* workspace,
                       int nchunk, spaa_stream_t stream);
/* Launch plan of the Winograd F(2x2,3x3) form (tiles 70 / 71; `desc` as for spaa_tapconv_f32 with a 16-position weight matrix):
 * host-side query, nothing is launched.  plan[0..7] = { N tile (64 / 128), K ranges, canvas layout (0 / 1), images per canvas
 * (rows), (columns), workgroups, 32-channel blocks per K range, canvases }.  Small images (ResNet-18 layer3 / layer4 behind
 * classifier.py:26-28, 14 x 14 and 7 x 7) are laid out on virtual canvases so that the 16 x 32-pixel workgroup regions are full,
 * and few regions with long K are cut into K ranges summed in fixed order by a second kernel: a caller sizes `splitk_ws`
 * (plan[1] * B * H * W * Npad floats) from the plan and passes plan[1] back as `ksplit` (ksplit = 1: never split; 0 with a
 * workspace: the launcher's own choice, which this function reports). */
int spaa_tapconv_wino_plan(const spaa_tapconv_t* desc, int32_t* plan);
/* The same query for the canvas / K-range form of the patch-staged fp16 kernel (tile 68 with `reserved1` bit 2: 3 x 3 / stride-1 layers of
 * the fp16-storage classifiers on 14 x 14 and 7 x 7 maps -- ResNet-18 layer3 / layer4, VGG-16's last block; classifier.py:26-28,
 * perc_al/__init__.py:181-238): plan[0..7] as above.  `desc->ksplit` > 1 asks for that many K ranges, 1 for none, 0 leaves the choice to
 * the plan; `reserved1` bits 0-1 likewise for the N tile.  The caller passes plan[1] back as `ksplit` (with `splitk_ws` of plan[1] * B * H *
 * W * Npad floats when > 1) and plan[0] in `reserved1`. */
int spaa_tapconv_h16p_plan(const spaa_tapconv_t* desc, int32_t* plan);
/* layout probes for language bindings: sizeof(spaa_tapconv_t) and the byte offset of field # `field`
 * (0 out, 1 weights, 2 taps, 3 gate2, 4 mask_out, 5 tap_range, 6 splitk_ws, 7 io_dtype, 8 nclass, 9 cls, 10 in2, 11 w2_split; else -1) */
int spaa_tapconv_sizeof(void);
int spaa_tapconv_offsetof(int field);

/* ---- layout conversion at the NCHW boundary ---------------------------------------------------------------- */
/* src [B,3,H,W] -> dst [B,H,W,4] (lane 3 = 0); optional clamp to [0,1] (projector_based_attack.py:265) */
int spaa_nchw_to_nhwc4(const float* src, float* dst, int B, int H, int W, int clamp01, spaa_stream_t stream);
/* dst [B,3,H,W]; optional final clamp to [0,1] (projector_based_attack.py:337) */
int spaa_nhwc4_to_nchw(const float* src, float* dst, int B, int H, int W, int clamp01, spaa_stream_t stream);

/* ---- WarpingNet (models.py:163-185, pytorch_tps.py:29-106) ----------------------------------------------- */
/* Coarse grid: F.affine_grid(affine_mat) sampled at tps_grid(theta, ctrl) (models.py:168-172), batch 1.
 * out: [Hout, Wout, 4] = (gx, gy, 0, 0) in normalised [-1,1] coordinates. theta: [T+2][2], ctrl: [T][2]. */
int spaa_warp_coarse_grid(const float* affine6, const float* theta, const float* ctrl, int T, int Hin, int Win,
                          int Hout, int Wout, float* out, spaa_stream_t stream);
/* fine = clamp(refine + coarse, -1, 1) (models.py:176); all [Hout, Wout, 4] */
int spaa_warp_finish_grid(const float* coarse, const float* refine, float* fine, int npix, spaa_stream_t stream);
/* F.grid_sample(clamp(x,0,1), fine_grid, bilinear, zeros, align_corners=True) * mask (models.py:184,340), and the
 * rough input [s, xw*s] (models.py:342).  x: [B,Hp,Wp,4]; grid: [Hc,Wc,4]; mask: [Hc,Wc]; s: [B,Hc,Wc,4];
 * xw: [B,Hc,Wc,4]; cat8: [B,Hc,Wc,8] = (s.rgb, xw*s .rgb, 0, 0) or NULL. */
int spaa_warp_fwd(const float* x, const float* grid, const float* mask, const float* s, float* xw, float* cat8,
                  int B, int Hp, int Wp, int Hc, int Wc, int clamp01, spaa_stream_t stream);
/* nn.Linear on a few rows (classifier.py:60: torchvision's `fc` = ATen addmm; its input gradient = mm with the weight):
 * out[m][n] = bias[n] + sum_k x[m][k] * w[n][k], fp32, M <= 256, K <= 4096, K % 4 == 0; x / w rows 16-byte aligned (ldx, ldw % 4 == 0),
 * bias may be NULL.  The input gradient is the same call with the transposed weight. */
int spaa_linear_small(const float* x, const float* w, const float* bias, float* out, int M, int K, int N, int ldx, int ldw, int ldo,
                      spaa_stream_t stream);
/* ShadingNetSPAA's two stride-2 entry layers with use_rough in one launch (models.py:284-285,295 of the reference):
 *   S1 = relu(conv1_s(cat[s, xw * s]) + bias_s),  X1 = relu(conv1(xw) + bias1 + S1)
 * xw, s: [B,H,W,4] fp32 (channel 3 = 0), H and W even; S1, X1: [B,H/2,W/2,32] fp32 (out_f16 = 0) or fp16 (1: the residual is
 * the rounded S1); mask_S1 / mask_X1: [B,H/2,W/2,8] ReLU gate bits of the stored values (1 byte per 4 channels) or NULL.
 * w_pair: [3][32][9][4] fp32 = conv1.weight, conv1_s.weight[:, 0:3], conv1_s.weight[:, 3:6] as [n][3 ky + kx][c], c = 3 zero. */
int spaa_conv1_pair_fwd(const float* xw, const float* s, const float* w_pair, const float* bias1, const float* bias_s,
                        void* S1, void* X1, uint8_t* mask_S1, uint8_t* mask_X1, int B, int H, int W, int out_f16,
                        spaa_stream_t stream);
/* fp16-storage mode, round 6: a FRACTIONAL-STRIDE 3 x 3 layer -- nn.ConvTranspose2d(Cin, Cout, 3, 2, 1, 1) forward (models.py:237,299
 * transConv1) or aten::convolution_backward(input) of nn.Conv2d(Cout, Cin, 3, 2, 1) (conv2, conv2_s: models.py:224,230 under autograd) --
 * written per INPUT pixel: exactly the nine real (output-parity class, tap) products, all weights resident in LDS, one persistent
 * workgroup per compute unit, no barrier after the prologue (csrc/fs2_h16.hip).  in: fp16 [B,Hi,Wi,in_cstride] (channels [0,Cin), Cin % 32
 * == 0); out: fp16 [B,2 Hi,2 Wi,Cout], Cout = 32 or 64.  w_img: the weights as the kernel's matrix operands, [Cin/32][9 pairs][Cout/16][64
 * lanes][8] fp16 (spaa_amd/models.py: pack_fs2 -- pair p = (operand in[y + r][x + q], class (cy, cx)) in the order (0,0) (0,1) (0,2) (0,3)
 * (1,1) (1,3) (2,2) (2,3) (3,3) of (2 r + q, 2 cy + cx); lane = (row n & 15, chunk g): element e = W[ky][kx][16 rb + (lane & 15)][32 ks +
 * 8 g + e] with ky = (cy == 0 ? 1 : r == 0 ? 2 : 0), kx likewise).  Optional second source at OUTPUT resolution (a 1 x 1 convolution added
 * before the epilogue: models.py:293,299 `+ skipConv2(x1)` and its mirror image in the backward pass): in2 fp16 [B,2 Hi,2 Wi,in2_cstride],
 * w2_img [Cin2/32][Cout/16][64][8].  Epilogue: + bias [Cout] (fp32, may be NULL), + add (fp16 [B,2 Hi,2 Wi,Cout], may be NULL), ReLU if
 * `relu`, gate_bits (byte masks of the output's shape: out = bit ? v : 0; may be NULL), fp16 store, mask_out = gate bytes of the stored
 * values (may be NULL).  LDS: (9 Cin/32 + Cin2/32) x Cout/16 KB <= 160 KB. */
int spaa_fs2_h16(const void* in, int in_cstride, int Cin, const void* w_img, const void* in2, int in2_cstride, int Cin2,
                 const void* w2_img, const float* bias, const void* add, const uint8_t* gate_bits, int relu, void* out,
                 uint8_t* mask_out, int Cout, int B, int Hi, int Wi, spaa_stream_t stream);

/* fp16-storage mode, round 6: the FORWARD form of a 3 x 3 / stride-2 / padding-1 convolution -- nn.Conv2d(Cin, Cout, 3, 2, 1) (models.py:224,230
 * conv2 / conv2_s) or aten::convolution_backward(input) of nn.ConvTranspose2d(Cout, Cin, 3, 2, 1, 1) (transConv1, models.py:237 under autograd)
 * -- as a persistent, barrier-free kernel with all weights resident in LDS (csrc/s2f_h16.hip).  in: fp16 [B,Hi,Wi,in_cstride] (Hi, Wi even,
 * channels [0,Cin), Cin % 32 == 0); out: fp16 [B,Hi/2,Wi/2,Cout], Cout = 64 or 128.  w_img: [Cin/32][9 taps 3 ky + kx][Cout/16][64 lanes][8]
 * fp16 (spaa_amd/models.py: pack_s2f; rows permuted as for spaa_fs2_h16).  Epilogue as spaa_fs2_h16: bias, add, ReLU, gate_bits, mask_out.
 * LDS: 9 Cin/32 x Cout/16 KB <= 160 KB; every tensor below 2 GiB (32-bit buffer offsets). */
int spaa_s2f_h16(const void* in, int in_cstride, int Cin, const void* w_img, const float* bias, const void* add, const uint8_t* gate_bits,
                 int relu, void* out, uint8_t* mask_out, int Cout, int B, int Hi, int Wi, spaa_stream_t stream);

/* fp32 mode, round 6: nn.Conv2d(Cin, 64, 3, 2, 1) forward (models.py:224,230 conv2 / conv2_s) on the same persistent weights-in-LDS form
 * with the bf16x6 arithmetic of the other fp32 kernels (exact fp32 operands split into three bf16 planes, six of the nine partial products,
 * fp32 accumulation; csrc/s2f_x6.hip).  in: fp32 [B,Hi,Wi,in_cstride] (Hi, Wi even, channels [0,Cin), Cin % 32 == 0); out: fp32
 * [B,Hi/2,Wi/2,64].  w_img: [Cin/32][9 taps 3 ky + kx][3 planes][4][64 lanes][8] bf16 (spaa_amd/models.py: pack_s2f_x6).  Epilogue: bias,
 * add (fp32, the output's shape), ReLU, gate_bits, mask_out (1 byte per 4 channels), any of them NULL.  LDS: 108 Cin/32 KB <= 160 KB (Cin =
 * 32); every tensor below 2 GiB. */
int spaa_s2f_x6(const float* in, int in_cstride, int Cin, const void* w_img, const float* bias, const float* add, const uint8_t* gate_bits,
                int relu, float* out, uint8_t* mask_out, int Cout, int B, int Hi, int Wi, spaa_stream_t stream);

/* the ADJOINT of the pair in fp16-storage mode (round 6): g_xw = conv1^T(g_x1) + scene * conv1_s^T(g_s1)[rough channels 3..5]
 * (models.py:284-285,295,342 under autograd: aten::convolution_backward(input) of both layers, the product with the surface image and the
 * sum), ONE launch instead of two thin-output launches with a round trip between them.  g_x1 / g_s1: fp16 [B,H/2,W/2,32] (already
 * ReLU-gated); scene, g_xw: fp32 [B,H,W,4]; w_image: the two layers' weights rounded to fp16 as the kernel's matrix operands,
 * [2 sources][4 operands (r, q)][64 lanes][8]: lane = (row 4 (2 cy + cx) + c, 8-channel chunk g), element e = weight[n = 8 g + e][c (+ 3
 * for the rough source)][ky][kx] with ky = (cy == 0 ? 1 : r == 0 ? 2 : 0), kx likewise, zero unless c < 3, r <= cy, q <= cx
 * (spaa_amd/models.py: pack_pair1_bwd); products on v_mfma_f32_16x16x32_f16, fp32 accumulation */
int spaa_conv1_pair_bwd_f16(const void* g_x1, const void* g_s1, const float* scene, const void* w_image, float* g_xw, int B, int H, int W,
                            spaa_stream_t stream);
/* Backward of the above w.r.t. x (grid_sampler_2d_backward + clamp mask): g_x must be zeroed by the caller
 * (spaa_zero); contributions are accumulated with float atomics.
 * g_xw: [B,Hc,Wc,4] gradient w.r.t. xw (conv1 path); g_xs: [B,Hc,Wc,4] gradient w.r.t. xw*s (channels 3..5 of
 * conv1_s's input) or NULL:  g_total = (g_xw + g_xs * s) * mask. */
int spaa_warp_bwd(const float* g_xw, const float* g_xs, const float* x, const float* grid, const float* mask,
                  const float* s, float* g_x, int B, int Hp, int Wp, int Hc, int Wc, int clamp01,
                  spaa_stream_t stream);

/* Deterministic form of the same backward: the grid is constant during an attack, so the scatter is transposed once.
 * spaa_warp_taps: per (camera pixel, tap) the projector pixel index (0x7fffffff = outside) and bilinear weight,
 * src/wgt: [Hc*Wc*4].  The host sorts the entries by source pixel (stable) into `order` (entry ids) and `off`
 * ([Hp*Wp+1] list boundaries); spaa_warp_bwd_gather then sums each projector pixel's list in fixed order (no atomics,
 * no zeroing, run-to-run reproducible). */
int spaa_warp_taps(const float* grid, int Hp, int Wp, int Hc, int Wc, int32_t* src, float* wgt, spaa_stream_t stream);
int spaa_warp_bwd_gather(const float* g_xw, const float* g_xs, const float* x, const float* mask, const float* s,
                         const int32_t* off, const int32_t* order, const float* wgt, float* g_x, int B, int Hp, int Wp,
                         int Hc, int Wc, int clamp01, spaa_stream_t stream);
/* The same gather (mask folded into the weights, no second gradient) with the camera-side operand staged through LDS: a
 * workgroup owns a 16 x 16 tile of projector pixels and 4 images and reads the bounding box of the camera pixels its tap
 * lists touch once per image.  lidx / w_e: per tap-list entry (in the order of `off`) the index inside the tile's box and
 * the weight; tbox: per tile (cy0, cx0, rows, columns) of its box, at most box_cap pixels (box_cap * 64 B <= 64 KiB).
 * Built once per attack by spaa_amd/models.py: tiled_taps.  Results are bitwise those of spaa_warp_bwd_gather. */
int spaa_warp_bwd_tiled(const float* g_xw, const float* x, const int32_t* off, const int32_t* lidx, const float* w_e,
                        const int32_t* tbox, int box_cap, float* g_x, int B, int Hp, int Wp, int Hc, int Wc, int clamp01,
                        spaa_stream_t stream);
/* ... with spaa_grad_sumsq folded into its epilogue (round 6: one launch and one pass over g_x less per iteration): g_x additionally takes
 * the prjl2 term's gradient for colour-step samples (prjl2_scale != 0: `state`, `gray` as spaa_grad_sumsq), and partial_ss
 * [B][ceil(Wp/16) * ceil(Hp/16)] receives the per-(image, 16 x 16 tile) sums of ||g_x||^2 in a fixed order -- consumed by
 * spaa_step_and_track_n with npartial = that tile count.  clamp_bits (may be NULL) [B][Hp*Wp]: the clamp gate's comparisons as written by
 * spaa_step_and_track_n for THIS x (bit c: channel c inside [0, 1]): the gate then reads 1 byte per pixel instead of x's 16 (x is still
 * read when prjl2_scale != 0); the caller answers for the bytes describing the x the forward pass read */
int spaa_warp_bwd_tiled_sumsq(const float* g_xw, const float* x, const int32_t* off, const int32_t* lidx, const float* w_e,
                              const int32_t* tbox, int box_cap, float* g_x, int B, int Hp, int Wp, int Hc, int Wc, int clamp01,
                              float gray, float prjl2_scale, const int32_t* state, float* partial_ss, const uint8_t* clamp_bits,
                              spaa_stream_t stream);
/* F.grid_sample forward (models.py:184,340) from the per-attack TAP TABLE of spaa_warp_taps instead of the grid: tap_src [Hc*Wc][4]
 * projector pixel of every bilinear tap (0x7fffffff: outside, weight 0), tap_wgt [Hc*Wc][4] its weight x mask -- the table the
 * deterministic backward pass is built from, so the pair is an exact adjoint.  A workgroup owns a 32 x 8 tile of camera pixels and four
 * images; xw [B,Hc,Wc,4].  (spaa_warp_fwd stays for the 8-channel concatenation output and for callers without a table.) */
int spaa_warp_fwd_taps(const float* x, const int32_t* tap_src, const float* tap_wgt, float* xw, int B, int Hp, int Wp, int Hc, int Wc,
                       int clamp01, spaa_stream_t stream);

/* ---- PCNet training step: WarpingNet parameter gradients and the optimiser (train_network.py:235-363) ------- */
/* d loss / d fine_grid summed over the batch: grid_sampler_2d_backward w.r.t. the GRID (models.py:184 under autograd).
 * g_xw: gradient w.r.t. the masked warped image [B,Hc,Wc,4]; x: the sampled image [B,Hp,Wp,4]; g_grid: [Hc,Wc,4] */
int spaa_warp_bwd_grid(const float* g_xw, const float* x, const float* grid, const float* mask, float* g_grid, int B, int Hp,
                       int Wp, int Hc, int Wc, spaa_stream_t stream);
/* backward of spaa_warp_finish_grid: g_sum = gradient w.r.t. (refine + coarse) (clamp gate), g_r6 (optional) = gradient
 * w.r.t. the refine net's last pre-activation (LeakyReLU 0.1) */
int spaa_warp_finish_grid_bwd(const float* g_fine, const float* coarse, const float* refine, float* g_sum, float* g_r6,
                              int npix, spaa_stream_t stream);
/* backward of spaa_warp_coarse_grid: g_params [6 + 2 (T+2)] = (d/d affine_mat [2,3] row-major, d/d theta [(T+2),2]);
 * partial: ceil(Hout*Wout/256) * (6 + 2 (T+2)) floats of scratch (block sums, added in order) */
int spaa_warp_coarse_grid_bwd(const float* g_coarse, const float* affine6, const float* theta, const float* ctrl, int T, int Hin,
                              int Win, int Hout, int Wout, float* partial, float* g_params, spaa_stream_t stream);
/* ---- fused tail / head of ShadingNetSPAA (/root/reference/src/python/models.py:296-300) --------------------------
 * forward:   Ypre = relu(conv6(relu(transConv2(X6) + bias2)) + bias6 + res1),  Y = min(Ypre, 1); the activation between
 *            the two layers (X7, 32 channels at camera resolution) stays in LDS, only its ReLU gate bytes reach HBM.
 *   x6 [B,H2,W2,64] fp32; w2_split [3][128][64] bf16: the three planes (w == h + m + l) of W[n][k] =
 *   transConv2.weight[k][c][py][px], n = 32 (2 py + px) + c; bias2 [32]; w6 [3][9][32]: w6[o][3 ky + kx][c] =
 *   conv6.weight[o][c][ky][kx]; bias6 [3]; res1, y, ypre [B,2 H2,2 W2,4]; mask7 [B,2 H2,2 W2,8] gate bytes of X7
 *   (bit e of byte q = channel 4 q + e > 0, as `mask_out` of spaa_tapconv_t).
 * backward:  P6 = gate6 . transConv2^T( gate7 . conv6^T(gP) ): gp [B,2 H2,2 W2,4] (gradient w.r.t. Ypre, channel 3
 *   ignored); w6t [27][32]: w6t[3 t + o][c] = conv6.weight[o][c][2 - t / 3][2 - t % 3]; w2t_split [3][64][128] bf16 planes
 *   of W[n][k] = transConv2.weight[n][c][py][px], k = 32 (2 py + px) + c; mask7 as above, mask6 [B,H2,W2,16] gate bytes
 *   of X6; p6 [B,H2,W2,64]. */
int spaa_shading_tail_fwd(const float* x6, const uint16_t* w2_split, const float* bias2, const float* w6, const float* bias6,
                          const float* res1, float* y, float* ypre, uint8_t* mask7, int B, int H2, int W2, spaa_stream_t stream);
int spaa_shading_head_bwd(const float* gp, const float* w6t, const uint16_t* w2t_split, const uint8_t* mask7,
                          const uint8_t* mask6, float* p6, int B, int H2, int W2, spaa_stream_t stream);
/* the same two kernels in fp16-storage mode (BASELINE.json configs[4]): x6 resp. p6 are fp16 [B,H2,W2,64]; the transposed
 * convolution's weights are ROUNDED TO fp16 like every other layer's of this mode -- here `w2_split` is ONE fp16 matrix [128][64]
 * (rows n = 32 (2 py + px) + c) and `w2t_split` ONE fp16 matrix [64][128] (same index order as the bf16 planes above) -- and its
 * products run on v_mfma_f32_16x16x32_f16 with fp32 accumulation (backward: the conv6 gradient is rounded to fp16 as that MFMA's
 * operand, the rounding the separate launches apply when they store it).  Round 5: the forward kernel keeps X7 in LDS as fp16 (the value a
 * separate transConv2 launch of this mode would store) and takes conv6's weights rounded to fp16 as well: `w6` of
 * spaa_shading_tail_fwd_f16 points at [3][9][32] fp16 ([o][3 ky + kx][c]), its taps accumulate in fp32 (v_dot2_f32_f16).  Images, res1,
 * gp and conv6's transpose (`w6t`) stay fp32 */
/* (spaa_hip 0.6: the fp16 operands are typed `const void*` -- `w2_half` / `w2t_half` ONE fp16 matrix, `w6_half` fp16 [3][9][32] -- so that a
 * caller written against the bf16-plane / fp32 meaning these arguments had before round 5 no longer compiles against this header; round 6:
 * conv6 and its transpose run on v_mfma_f32_16x16x32_f16 as well (backward: conv6's weights `w6t` fp32 [27][32] are rounded to fp16 in the
 * kernel, the values the forward pass multiplies; the cotangent enters as hi + lo fp16 halves, exact to 2^-22)) */
int spaa_shading_tail_fwd_f16(const void* x6, const void* w2_half, const float* bias2, const void* w6_half, const float* bias6,
                              const float* res1, float* y, float* ypre, uint8_t* mask7, int B, int H2, int W2, spaa_stream_t stream);
int spaa_shading_head_bwd_f16(const float* gp, const float* w6t, const void* w2t_half, const uint8_t* mask7,
                              const uint8_t* mask6, void* p6, int B, int H2, int W2, spaa_stream_t stream);

/* the backward head with spaa_select_grad folded into its first phase (one launch and one [B,H,W,4] round trip less per iteration):
 * the cotangent of sample b is g_col (state[4 b + 1] != 0: the sample takes the colour step, projector_based_attack.py:310-315)
 * or g_adv (:302-307), gated by 0 < ypre <= 1 (backward of clamp(relu(.), max = 1), models.py:301); all three [B,2 H2,2 W2,4] */
int spaa_shading_head_bwd_select(const float* g_adv, const float* g_col, const int32_t* state, const float* ypre, const float* w6t,
                                 const uint16_t* w2t_split, const uint8_t* mask7, const uint8_t* mask6, float* p6, int B, int H2,
                                 int W2, spaa_stream_t stream);
int spaa_shading_head_bwd_select_f16(const float* g_adv, const float* g_col, const int32_t* state, const float* ypre, const float* w6t,
                                     const void* w2t_half, const uint8_t* mask7, const uint8_t* mask6, void* p6, int B, int H2,
                                     int W2, spaa_stream_t stream);

/* round 6: the same kernels with the clamp gate of the network output (models.py:301: backward of clamp(relu(.), max = 1) passes where
 * 0 < pre <= 1) as ONE byte per pixel -- gate_y [B,2 H2,2 W2], bit e = channel e passes -- instead of the 16-byte pre-clamp pixel: the tail
 * writes it (`ypre` may then be NULL: 63 MB less per batch-64 pass), the select head reads it instead of `ypre` (63 MB less) */
int spaa_shading_tail_fwd_g(const float* x6, const uint16_t* w2_split, const float* bias2, const float* w6, const float* bias6,
                            const float* res1, float* y, float* ypre, uint8_t* mask7, uint8_t* gate_y, int B, int H2, int W2,
                            spaa_stream_t stream);
int spaa_shading_tail_fwd_f16_g(const void* x6, const void* w2_half, const float* bias2, const void* w6_half, const float* bias6,
                                const float* res1, float* y, float* ypre, uint8_t* mask7, uint8_t* gate_y, int B, int H2, int W2,
                                spaa_stream_t stream);
int spaa_shading_head_bwd_select_g(const float* g_adv, const float* g_col, const int32_t* state, const uint8_t* gate_y, const float* w6t,
                                   const uint16_t* w2t_split, const uint8_t* mask7, const uint8_t* mask6, float* p6, int B, int H2,
                                   int W2, spaa_stream_t stream);
int spaa_shading_head_bwd_select_f16_g(const float* g_adv, const float* g_col, const int32_t* state, const uint8_t* gate_y,
                                       const float* w6t, const void* w2t_half, const uint8_t* mask7, const uint8_t* mask6, void* p6,
                                       int B, int H2, int W2, spaa_stream_t stream);

/* ReLU backward as a stand-alone op: out = (act > 0) ? g : 0, n floats (n % 4 == 0, 16-byte aligned) */
int spaa_relu_gate(const float* g, const float* act, float* out, int64_t n, spaa_stream_t stream);
/* torch.optim.Adam step on one flat parameter tensor (train_network.py:252-254: betas (0.9, 0.999), eps 1e-8, L2 weight
 * decay added to the gradient); `step` counts from 1 */
int spaa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, double beta1,
                   double beta2, float eps, float weight_decay, int step, spaa_stream_t stream);

/* ---- stealthiness losses (projector_based_attack.py:275-287; perc_al/differential_color_functions.py) ----- */
/* rgb [B,H,W,4] -> lab [B,H,W,4]  (rgb2lab_diff :39-64) */
int spaa_rgb2lab(const float* rgb, float* lab, int npix, spaa_stream_t stream);
/* per-pixel CIEDE2000 map (ciede2000_diff :109-180) of two Lab images -> de [npix] */
int spaa_ciede2000(const float* lab1, const float* lab2, float* de, int npix, spaa_stream_t stream);
/* autograd of the two functions above (the reference differentiates them with torch.autograd — the `_diff` in
 * their names): g_rgb = J_lab(rgb)^T g_lab;  g_lab1 = g_de * d dE/d lab1, g_lab2 = g_de * d dE/d lab2 (either may
 * be NULL).  All images [npix][4]. */
int spaa_rgb2lab_bwd(const float* rgb, const float* g_lab, float* g_rgb, int npix, spaa_stream_t stream);
int spaa_ciede2000_bwd(const float* lab1, const float* lab2, const float* g_de, float* g_lab1, float* g_lab2, int npix,
                       spaa_stream_t stream);
/* PCNet training loss (train_network.py:367-392 compute_loss; pytorch_ssim/__init__.py:26-58), forward and gradient:
 *   loss = l1_w * mean|infer - target| + ssim_w * (1 - mean SSIM(infer, target))      (means over B*3*H*W)
 * infer, target, g_infer: NHWC4 [B,H,W,4]; window: the 11x11 Gaussian [121]; m_mu / m_11 / m_12: [B,H,W,4] scratch maps
 * (needed when ssim_w != 0); partial [B * ceil(H/16) * ceil(W/16)][3] = block sums of (SSIM map, |d|, d^2): the host
 * adds them in order (loss value and the MSE the reference logs).  g_infer = d loss / d infer. */
int spaa_train_loss_fwd_bwd(const float* infer, const float* target, const float* window, float l1_w, float ssim_w,
                            float* m_mu, float* m_11, float* m_12, float* partial, float* g_infer, int B, int H, int W,
                            spaa_stream_t stream);
/* Fused camera-side stealth loss + gradient (one launch replaces ~600 ATen ops):
 *   caml2_px = ||scene - y||_2 over rgb ; camdE_px = dE00(lab(y), scene_lab)
 *   g_y      = gscale * (caml2_w * d caml2_px/dy + camdE_w * d camdE_px/dy)      gscale = 1/(B*H*W)
 * y, scene, scene_lab, g_y: [B,HW,4]; de_map: optional [B,HW] per-pixel dE; partial: [B][nblk][3] block partial
 * sums (caml2, camdE, camdE^2) with nblk = ceil(HW/256), reduced in fixed order by spaa_decide / spaa_scale_by_map. */
int spaa_stealth_loss_fwd_bwd(const float* y, const float* scene, const float* scene_lab, float caml2_w,
                              float camdE_w, float gscale, float* g_y, float* de_map, float* partial, int B, int HW,
                              spaa_stream_t stream);

/* calc_img_dists (utils.py:420-491), NHWC4 images x, y of `npix` = B*H*W pixels: per-pixel terms of PSNR/RMSE (sum of
 * squared differences), mean L2 (:460-471), mean L_inf (:475-486) and mean dE2000 (differential_color_functions.py:183-190)
 * reduced per 256-pixel block in fixed order: partial[(npix+255)/256][4] = (sum d^2, sum ||d||_2, sum max|d|, sum dE). */
int spaa_img_dists(const float* x, const float* y, float* partial, int npix, spaa_stream_t stream);

/* SSIM map sum (pytorch_ssim/__init__.py:26-58: 11x11 Gaussian `window` [121] as create_window builds it, replicate
 * padding, C1 = 0.01^2, C2 = 0.03^2, per channel): partial[B][ceil(H/16)][ceil(W/16)] = sum over the tile's pixels and
 * 3 channels; the mean is the sum of all partials / (B*3*H*W). */
int spaa_ssim(const float* x, const float* y, const float* window, float* partial, int B, int H, int W,
              spaa_stream_t stream);

/* ---- classifier pre/post-processing (classifier.py:55-72, img_proc.py:117-132) --------------------------- */
/* center_crop + F.interpolate(mode='area') + Normalize, NHWC4 in -> NHWC4 out; mean3/std3 are HOST pointers */
int spaa_preproc_fwd(const float* y, float* out, int B, int H, int W, int cy0, int cx0, int ch, int cw, int oh,
                     int ow, const float* mean3, const float* std3, spaa_stream_t stream);
/* adjoint: g_out [B,oh,ow,4] -> g_y [B,H,W,4] (zero outside the crop) */
int spaa_preproc_bwd(const float* g_out, float* g_y, int B, int H, int W, int cy0, int cx0, int ch, int cw,
                     int oh, int ow, const float* std3, spaa_stream_t stream);
/* max_pool2d k3 s2 p1 forward (writes per element the argmax window offset 0..8 in bits 0-6 and, in bit 7, whether the
 * maximum is positive) and backward (gather; `relu_gate` != 0: the pooled tensor's producer is a ReLU, windows whose
 * maximum is not positive pass no gradient — read from bit 7, not from the activation), NHWC, C % 4 == 0 */
int spaa_maxpool3s2_fwd(const float* in, float* out, uint8_t* argmax, int B, int Hin, int Win, int C, int Hout,
                        int Wout, spaa_stream_t stream);
int spaa_maxpool3s2_bwd(const float* g_out, const uint8_t* argmax, int relu_gate, float* g_in, int B,
                        int Hin, int Win, int C, int Hout, int Wout, spaa_stream_t stream);
/* adaptive_avg_pool2d(1): [B,HW,C] -> [B,C]; backward broadcasts g/HW and applies the ReLU gate of `act` */
int spaa_avgpool_fwd(const float* in, float* out, int B, int HW, int C, spaa_stream_t stream);
int spaa_avgpool_bwd(const float* g_out, const float* act, float* g_in, int B, int HW, int C,
                     spaa_stream_t stream);

/* Generic NHWC pooling of the VGG-16 / Inception-v3 bodies (C % 4 == 0); `*_cstride/_coff` address a channel window
 * of a concatenated buffer; backward passes are deterministic gathers with an optional ReLU gate of the input.
 * NaN contract: the stand-alone pool propagates a NaN of its input ("NaN wins", like ATen's max_pool2d).  Every ReLU epilogue of this
 * library is fmaxf(v, 0), which maps a NaN pre-activation to 0 -- in the separate conv + ReLU launch exactly as in the fused
 * conv + ReLU + 2 x 2 pool epilogue (spaa_tapconv_h16p, bit 6) -- so the two forms of a conv -> ReLU -> pool chain agree on every input,
 * NaN included: a NaN produced by a convolution never reaches a pool.  The library's contract is FINITE VALUES ONLY (inputs in range and
 * finite weights give finite activations); torch's relu would propagate a NaN where these epilogues clear it. */
int spaa_maxpool_fwd(const float* in, float* out, uint8_t* argmax, int B, int Hin, int Win, int C, int Hout, int Wout,
                     int k, int s, int p, int out_cstride, int out_coff, spaa_stream_t stream);
int spaa_maxpool_bwd(const float* g_out, const uint8_t* argmax, int relu_gate, float* g_in, int B, int Hin,
                     int Win, int C, int Hout, int Wout, int k, int s, int p, int gout_cstride, int gout_coff,
                     spaa_stream_t stream);
/* fp16-storage variants (activations and their gradients fp16; argmax bytes as above) */
int spaa_maxpool_fwd_f16(const void* in, void* out, uint8_t* argmax, int B, int Hin, int Win, int C, int Hout, int Wout,
                         int k, int s, int p, int out_cstride, int out_coff, spaa_stream_t stream);
int spaa_maxpool_bwd_f16(const void* g_out, const uint8_t* argmax, int relu_gate, void* g_in, int B, int Hin, int Win,
                         int C, int Hout, int Wout, int k, int s, int p, int gout_cstride, int gout_coff,
                         spaa_stream_t stream);
/* global average pool, fp16 activation in -> fp32 features out; backward fp32 feature gradient -> fp16 gradient, with the
 * ReLU gate of `act` (fp16, may be NULL) */
int spaa_avgpool_fwd_f16(const void* in, float* out, int B, int HW, int C, spaa_stream_t stream);
int spaa_avgpool_bwd_f16(const float* g_out, const void* act, void* g_in, int B, int HW, int C, spaa_stream_t stream);
/* ReLU-gate bytes (format of spaa_tapconv_t.mask_out) of channels [coff, coff + C) of an activation [M pixels][cstride] (fp32, or
 * fp16 when f16 != 0) into mask [M][cstride / 4]: for activations that no convolution launch wrote -- the max-pool outputs that
 * gate their consumers in torchvision's Inception-v3 (/root/reference/src/python/classifier.py:29-33; autograd's threshold_backward
 * of the ReLU that produced the pooled values) */
int spaa_gate_mask(const void* act, int f16, uint8_t* mask, int64_t M, int C, int cstride, int coff, spaa_stream_t stream);
/* avg_pool2d(k, s, p), count_include_pad=True */
int spaa_avgpool2d_fwd(const float* in, float* out, int B, int Hin, int Win, int C, int Hout, int Wout, int k, int s,
                       int p, int out_cstride, int out_coff, spaa_stream_t stream);
int spaa_avgpool2d_bwd(const float* g_out, float* g_in, int B, int Hin, int Win, int C, int Hout, int Wout, int k,
                       int s, int p, int gout_cstride, int gout_coff, spaa_stream_t stream);
/* fp16-storage variants (fp32 accumulation, one rounding on the way out): Inception-v3's 3x3 / stride-1 branch pools */
int spaa_avgpool2d_fwd_f16(const void* in, void* out, int B, int Hin, int Win, int C, int Hout, int Wout, int k, int s,
                           int p, int out_cstride, int out_coff, spaa_stream_t stream);
int spaa_avgpool2d_bwd_f16(const void* g_out, void* g_in, int B, int Hin, int Win, int C, int Hout, int Wout, int k,
                           int s, int p, int gout_cstride, int gout_coff, spaa_stream_t stream);
/* adaptive_avg_pool2d((Hout, Wout)) */
int spaa_adaptive_avgpool_fwd(const float* in, float* out, int B, int Hin, int Win, int C, int Hout, int Wout,
                              spaa_stream_t stream);
int spaa_adaptive_avgpool_bwd(const float* g_out, const float* gate_in, float* g_in, int B, int Hin, int Win, int C,
                              int Hout, int Wout, spaa_stream_t stream);

/* ---- SPAA Algorithm 1 control on device (projector_based_attack.py:269-328) ------------------------------ */
/* Per-sample decision, one workgroup per sample: softmax top-1 / argmax of logits (classifier.py:64-68), loss
 * reduction, masks (:290-299), best bookkeeping (:318-320), and the seed of the adversarial backward pass
 * g_logits[b][c] = (c == target_b) ? -/+adv_scale : 0  (adv_scale = adv_w / B).
 * state (int32 [B][4]): 0 succ, 1 best_adv, 2 best, 3 top1.   stats (float [B][8]): 0 p1, 1 caml2, 2 camdE,
 * 3 col_loss, 4 prjl2, 5 col_loss_best (in/out, init 1e6), 6 target logit, 7 reserved.  prjl2 may be NULL. */
int spaa_decide(const float* logits, int ncls, const int32_t* target, int targeted, const float* partial, int nblk,
                int HW, const float* prjl2, float prjl2_w, float caml2_w, float camdE_w, float d_thr,
                float p_thresh, float adv_scale, int32_t* state, float* stats, float* g_logits, int B,
                spaa_stream_t stream);
/* Cotangent at the PCNet output: g = best_adv_b ? g_col : g_adv (one backward pass serves both of the reference's,
 * :302,310), then the backward of clamp(relu(pre), max=1) (models.py:301): pass where 0 < ypre <= 1 (ypre NULL: none).
 * all [B,npix,4] */
int spaa_select_grad(const float* g_adv, const float* g_col, const int32_t* state, const float* ypre, float* g,
                     int B, int npix, spaa_stream_t stream);
/* prjl2_b = mean_px ||gray - x||_2 (:275) */
int spaa_prjl2_fwd(const float* x, float gray, float* prjl2, int B, int HW, spaa_stream_t stream);
/* In place g += prjl2_scale * d prjl2_px/dx for colour-step samples (prjl2_scale = prjl2_w/(B*HW), 0 = off), and
 * block partials of ||g_b||^2 -> partial [B][ceil(HW/256)] */
int spaa_grad_sumsq(float* g, const float* x, float gray, float prjl2_scale, const int32_t* state, float* partial,
                    int B, int HW, spaa_stream_t stream);
/* x_b -= lr_b * g_b/||g_b||_2 with lr = best_adv ? col_lr : adv_lr (:307,315); then where succ: x_best_b = x_b
 * (post-step, Q4) and cam_best_b = cam_b (:323-328).   x, g, x_best: [B,HWp,4]; cam, cam_best: [B,HWc,4] */
int spaa_step_and_track(float* x, const float* g, const float* partial, const int32_t* state, float adv_lr,
                        float col_lr, float* x_best, const float* cam, float* cam_best, int B, int HWp, int HWc,
                        spaa_stream_t stream);
/* the same with the number of partial sums per sample stated (partial [B][npartial]: spaa_warp_bwd_tiled_sumsq's tile sums);
 * clamp_bits (may be NULL) [B][HWp]: receives, per pixel of the UPDATED x, bit c = (0 <= x_c <= 1) -- the clamp gate of the next
 * iteration's backward pass (models.py:337 `x.clamp(0, 1)` under autograd) */
int spaa_step_and_track_n(float* x, const float* g, const float* partial, int npartial, const int32_t* state, float adv_lr,
                          float col_lr, float* x_best, const float* cam, float* cam_best, int B, int HWp, int HWc,
                          uint8_t* clamp_bits, spaa_stream_t stream);

/* ---- PerC_AL.adversary_projector (perc_al/__init__.py:133-256) ------------------------------------------ */
/* x = a + b (inputs + delta), NHWC4 */
int spaa_add_nhwc4(const float* a, const float* b, float* x, int npix, spaa_stream_t stream);
/* g_logits = mult * d CrossEntropy_sum / d logits = mult * (softmax - onehot)  (:186-187) */
int spaa_ce_grad(const float* logits, int ncls, const int32_t* label, float mult, float* g_logits, int B,
                 spaa_stream_t stream);
/* x_b += step * g_b/||g_b||_2 where (state[b][col] != 0) == want; partial from spaa_grad_sumsq  (:193-195,204-209) */
int spaa_masked_step(float* x, const float* g, const float* partial, const int32_t* state, int col, int want,
                     float step, int B, int HW, spaa_stream_t stream);
/* g_px *= dE_px / ||dE_b||_2 (gradient of color_dis = ||d_map||_2, :198-200), color_dis_b out; partial3 from
 * spaa_stealth_loss_fwd_bwd */
int spaa_scale_by_map(float* g, const float* de_map, const float* partial3, float* color_dis, int B, int HW,
                      spaa_stream_t stream);
/* delta = clamp(inputs+delta,0,1) - inputs; x_round = round((inputs+delta)*255)/255 (:211-212,15-18); partial
 * [B][nblk] = block sums of ||delta_px||_2 (:215) */
int spaa_perc_clamp_quant(const float* inputs, float* delta, float* x_round, float* partial, int B, int HW,
                          spaa_stream_t stream);
/* masks on the quantised image (:216-243). mode 0 targeted, 1 untargeted, 2 untargeted with Carlini margin.
 * state [B][4]: isadv, best_adv, best, top1; stats [B][8]: p1, caml2, margin, color_dis, -, bound_best(in/out) */
int spaa_perc_decide(const float* logits, int ncls, const int32_t* label, int mode, float confidence,
                     const float* partial, int nblk, int HW, const float* color_dis, float d_thr, float p_thresh,
                     int32_t* state, float* stats, int B, spaa_stream_t stream);
/* dst_b = src_b where state[b][0] != 0 (:244-245) */
int spaa_track_where(const float* src, float* dst, const int32_t* state, int B, int HW, spaa_stream_t stream);

/* misc */
int spaa_zero(void* p, int64_t bytes, spaa_stream_t stream);
const char* spaa_version(void);

#ifdef __cplusplus
}
#endif
#endif
