/* c2m_hip.h -- C ABI of libc2m_hip.so, the MI355X (gfx950) kernel library behind c2m_amd.
 *
 * The reference (PierfrancescoArdino/C2M) has no FFI on this path: its hot path is stock ATen operators called from
 * Python.  Each entry point below therefore names the reference *call site* (file:line under
 * /root/reference/src) whose ATen op chain it replaces; the Python-side binding is the ctypes stub in
 * c2m_amd/_lib.py (shown in INTEGRATION.md).
 *
 * Conventions (every function):
 *   - plain pointers to DEVICE memory, fp32 unless noted, contiguous NCHW / NCTHW; sizes as int / long
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and never synchronise or allocate;
 *     scratch is caller-provided (`workspace`), sized by the matching *_workspace_* query
 *   - return value: hipError_t as int (0 = success); shape errors return hipErrorInvalidValue
 *   - thread-safe and re-entrant (no global mutable state); callable from the autograd thread
 */
#ifndef C2M_HIP_H
#define C2M_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { C2M_ACT_NONE = 0, C2M_ACT_RELU = 1, C2M_ACT_LRELU = 2, C2M_ACT_SIGMOID = 3 };

/* Names of the geom[] entries of the convolution entry points (enum c2m_geom_index, enum c2m_wino_geom_index) and the ABI version
 * of this header.  A binding checks at load time that the library was built from the same header:
 *   c2m_abi_version() == C2M_ABI_VERSION, c2m_geom_len() == C2M_G_LEN, c2m_wino_geom_len() == C2M_WG_LEN.                      */
#include "c2m_geom.h"
int c2m_abi_version(void);
int c2m_geom_len(void);
int c2m_wino_geom_len(void);

/* ---- convolution: implicit GEMM on v_mfma_f32_32x32x2_f32 (conv_igemm.hip) -------------------------------------
 * Replaces nn.Conv2d / nn.Conv3d (+ ReflectionPad2d/3d, bias, LeakyReLU/ReLU/Sigmoid) at:
 *   modules/layers/down_block.py:14-23,35-47   same_block.py:14-23,36-46,55-67   up_block.py:9-13
 *   residual_block.py:13-31,42-71   spade_block.py:47-49   vgg.py:92-137   generator/generator.py:76-78
 *   and their autograd backward (aten::convolution_backward).
 *
 * D[m][pix] = act( sum_k A[m][k] * G(k,pix) + bias[m] ),  G = input gathered through `ktab`.
 * K order: (channel chunk, tap group, tap slot, channel in chunk); one 16-deep K-step = NS taps x CK channels.
 * ktab: nk groups of (1 + NS) int4: header {channel_offset, nvalid_channels (-2: ones row), 0, 0} then NS taps
 *       {dt, dy, dx, valid}; A is the weight matrix packed by the host into the same order ([M][nk*16]).
 * geom[] (int64; 93 entries -- 90 / 91 / 92 are read by every call --, 120 when parity classes are batched):
 *   0 M   1 nk (K-steps; for wgrad: J rows)   2 lda   3 Npix = N*To*Ho*Wo   4 To 5 Ho 6 Wo   7 Ti 8 Hi 9 Wi
 *   10 st 11 sh 12 sw (input coord = o*stride + tap offset)   13 in_sn 14 in_st 15 in_sh (elements; w stride 1)
 *   16 out_sn 17 out_sc 18 out_st 19 out_sh 20 out_sw 21 out_off   22 reflect (0 zeros / 1 reflect) 23 is3d
 *   24 NS (1, 2 or 4)   25 in_sc (channel stride)   26 splits (from c2m_conv_igemm_splits, or 1)
 *   27 slab_stride (elements between split-K slabs)   28 Cin 29 taps 30 tap groups per chunk 31 real groups (wgrad)
 *   32 x_bytes 33 dy_bytes (wgrad; buffer-load bounds, < 2 GiB)
 *   34 operand precision: 0 = fp32 (exact v_mfma_f32_32x32x2_f32), 1 = bf16 operands, fp32 accumulation
 *      (v_mfma_f32_32x32x16_bf16).  Element types in memory (the bf16 data path, BASELINE configs[2-4]): geom[90] X
 *      (wgrad: X and dY), geom[91] Y / Y_interior -- 0 = fp32, 1 = bf16.  The bf16 kernels gather bf16 X ALWAYS (the host
 *      casts an fp32 input once) and write bf16 or fp32 results; split-K slabs are fp32 (c2m_splitk_reduce's `dt` picks the
 *      final type).  The fp32 kernels and the <= 4-output-channel vector-ALU kernels are fp32 in, fp32 out.
 *   36..51 two-target epilogue (reflect-pad dgrad, Y_interior != NULL): padded coordinate = o*ps + po per dim
 *      (36-38 ps, 39-41 po); outputs inside [lo, lo+ext) (42-44 lo, 45-47 ext) are stored to Y_interior with strides
 *      48 sn 49 sc 50 st 51 sh, the pad ring to Y; c2m_reflect_border_add then folds the ring.
 *   52 LDS-patch kernel (3x3, stride 1, fp32): 1 = on; 53 iy0 54 ix0 input origin of an output tile relative to its
 *      first output; 55-57 / 58-60 patch row / column of tap row / column 0,1,2 ((0,1,2) forward, (2,1,0) dgrad)
 *   61 ncls: stride parity classes batched into this launch (blockIdx.z = class*splits + split); 62 floats per class
 *      weight matrix, 63 int4 entries per class tap table, 64+c out_off of class c, 96+3c.. its (po_t, po_y, po_x)
 *      (geom[] holds 120 entries; 90 / 91 / 92 are the element-type and weight-gradient-form flags)
 *   94 NC8 gather form (conv_gather_nc8_kernel, round 4; every bf16 forward / data gradient that is not a 3x3 / 4x4-stride-2 /
 *      3x3x3 patch layer -- down_block.py:14-23 and same_block.py:50-68 in 3-D, the discriminator tails, 1x1 and 7x7 layers, the
 *      reflect data gradients over (W/2 + 1)-wide class planes): 1 = X is the NC8 form ([N][ceil(C/8)][Ti*Hi*Wi][8] bf16, geom[32]
 *      its bytes) of the gathered tensor, A the c2m_pack_weights_bf16_gather image of the launch's first class, ktab a compact
 *      table [ncls][taps] of {dt, dy, dx, 1}, geom[1] = taps * ceil(C/16) K-steps in (tap, chunk) order (C = geom[28], taps =
 *      geom[29] <= 64), geom[24] = 1; classes, splits, the two-target epilogue and the output types as for the NCHW gather kernel.
 *      95: tile variant (0 = by rows: 32 / 64 / 128-row tiles x 256 pixels).
 * With splits > 1, Y must point at a slab of splits*slab_stride floats and c2m_splitk_reduce finishes the op
 * (sum over splits in a fixed order, + bias[(i / chan_stride) % M], activation).                               */
int c2m_conv_igemm_splits(int M, int nk, int Npix);
int c2m_conv_igemm(const float* A, const void* X, void* Y, void* Y_interior, const float* bias, const int* ktab,
                   const int64_t* geom, int act, float slope, void* stream);
int c2m_splitk_reduce(const float* slab, void* out, const float* bias, long total, int splits, long chan_stride,
                      int M, int act, float slope, int dt /* element type of out */, void* stream);

/* dW[m][c][tap] (+ db[m]) = sum_pix dY[m][pix] * G(row,pix): rows in the same (chunk, tap group, slot, channel)
 * order as the forward K, plus one ones-row group (bias gradient), padded to c2m_conv_wgrad_rows(M, groups+1).
 * Deterministic split-K over pixels: slab = c2m_conv_wgrad_splits(M, J, Npix) * M * J floats; the slab reduction
 * writes dW in the native [Cout][Cin][taps] layout.  geom: 0 M, 1 J, 16 dy_sn, 17 dy_sc, 24 NS, 28..31 as above. */
int c2m_conv_wgrad_splits(int M, int J, int Npix);
int c2m_conv_wgrad_rows(int M, int ngroups);
int c2m_conv_wgrad(const void* dY, const void* X, float* slab, float* dW, float* db, const int* jtab,
                   const int64_t* geom, void* stream);

/* Native conv weights ([Cout][Cin][kt][kh][kw]) -> the packed K order above, one launch; also the per-parity-class
 * transposed matrices of the data gradient (rows = input channels), all classes stacked along the rows.
 * g[]: 0 M rows per class, 1 C (K-side channels), 2 CK, 3 KT, 4 KH, 5 KW, 6 st, 7 sh, 8 sw (1,1,1 = forward),
 *      9 row stride, 10 channel stride of the source (elements).  Replaces the reshape/permute/copy chain that
 *      aten::convolution does internally on its weights (no reference line of its own).                          */
int c2m_pack_weights(const float* w, float* packed, const int64_t* g, void* stream);
/* Every pack of a model in ONE launch (after an optimizer step each trainable weight needs each of its layouts rebuilt once:
 * ~195 launches of 5-7 us in a full G + D step).  The host fills one job record per (weight, layout) -- type 0: the arguments of
 * c2m_pack_weights, type 1: those of c2m_pack_weights_bf16_patch, type 2: those of c2m_pack_weights_bf16_gather -- with the index of its first workgroup (ascending, dense),
 * keeps the table (and one (job, workgroup) pair per workgroup of the launch) in device memory and refreshes all packs in place
 * with c2m_pack_multi.  c2m_pack_job_fill returns the number of workgroups of the job (< 0: bad geometry).                                                           */
int c2m_pack_job_bytes(void);
long c2m_pack_job_fill(void* host_job, int type, const void* w, void* packed, const int64_t* g, unsigned first_block);
int c2m_pack_multi(const void* device_jobs, const void* device_blocktab /* int32 (job, workgroup in job) per workgroup */,
                   int njobs, long total_blocks, void* stream);
/* bf16 weight image of the bf16 LDS-patch kernel (3x3 stride-1 layers in bf16 mode, BASELINE configs[2-4]):
 * out[chunk][tap][row padded to a multiple of 128][16 channels] bf16.  g[] = {M, C, s_m, s_c, flip}; flip = 1 for the
 * data gradient (rotated taps).  Pass the result as A to c2m_conv_igemm with geom[2] (lda) = the padded row count. */
long c2m_pack_weights_bf16_patch_bytes(int M, int C);
int c2m_pack_weights_bf16_patch(const float* w, void* out, const int64_t* g, void* stream);
/* bf16 weight images of the NC8 gather form (geom[94] of c2m_conv_igemm): g[] as c2m_pack_weights with g[2] = 16; out = prod(stride)
 * class images ((rt, ry, rx) row-major) of [tap][16-channel chunk][half][row padded to a multiple of 128] 16-byte units (8 bf16,
 * RNE; zero rows / channels past M / C).  Job type 2 of c2m_pack_job_fill / c2m_pack_multi. */
long c2m_pack_weights_bf16_gather_bytes(const int64_t* g);
int c2m_pack_weights_bf16_gather(const float* w, void* out, const int64_t* g, void* stream);

/* Channel-blocked ("NC8") bf16 convolutions (conv_nc8.hip, round 4; the same 3x3 stride-1 call sites -- vgg.py:92-137,
 * residual_block.py:13-31,42-71, spade_block.py:47-49, up_block.py:9-13, same_block.py:14-23 -- in bf16 mode).
 * c2m_nchw_to_nc8: [N][C][HW] bf16 -> [N][ceil(C/8)][HW][8] bf16 (zero channels past C; HW % 8 == 0): the 8 channels one
 * MFMA lane consumes become ONE 16-byte unit.  c2m_conv_patch_nc8: D = act(W * patches(X) + bias) for a 2-D 3x3 stride-1
 * layer with X in NC8, A = the c2m_pack_weights_bf16_patch image, operands fetched global -> LDS by 16-byte LDS-DMA;
 * geom[] as for c2m_conv_igemm's LDS-patch path (0 M, 1 nk = 9 * chunks, 2 padded rows of A, 3 Npix, 5 Ho, 6 Wo, 8 Hi, 9 Wi,
 * 16-21 output strides / offset, 22 reflect, 26 splits, 27 slab stride, 28 Cin, 36-51 two-target block, 52 = 1, 53-60 patch
 * origin and tap order, 91 output type); Y / Y_interior are NCHW (bf16 or fp32), split-K slabs fp32.                        */
int c2m_nchw_to_nc8(const void* x, void* y, long N, int C, long HW, void* stream);
int c2m_conv_patch_nc8(const void* A, const void* X_nc8, void* Y, void* Y_interior, const float* bias, const int64_t* geom,
                       int act, float slope, void* stream);
/* 4x4 stride-2 pad-1 2-D layers (down_block.py:14-23, discriminator.py:59-89) on the parity-plane form of the same kernel: four
 * (16 channels, input parity) chunks with 2x2 taps each -- the stride-2 gather is the LDS-DMA's per-lane address.  A =
 * c2m_pack_weights_bf16_patch(g[4] = 2) (c2m_pack_weights_bf16_s2_bytes bytes); Y contiguous [N][M][Hi/2][Wi/2], bf16 (yh) or fp32. */
long c2m_pack_weights_bf16_s2_bytes(int M, int C);
int c2m_conv_s2_nc8(const void* A, const void* X_nc8, void* Y, const float* bias, int M, int C, long N, int Hi, int Wi, int reflect,
                    int yh, int act, float slope, void* stream);
/* Data gradient of those layers: the four output parity classes as 2x2 stride-1 correlations of dY over one shared patch, one
 * launch.  A = c2m_pack_weights_bf16_patch with g = {M = input channels, C = output channels, s_m = 16, s_c = M * 16, mode 3 (zeros) /
 * 4 (reflect)}; T = dX [N][M][2Ho][2Wo] (zeros) or the padded gradient [N][M][2Ho+2][2Wo+2] (reflect; c2m_reflect_fold finishes). */
int c2m_conv_s2_dgrad_nc8(const void* A, const void* dY_nc8, void* T, int M, int K, long N, int Ho, int Wo, int reflect, int th,
                          void* stream);
/* 3x3x3 stride-1 pad-1 layers (same_block.py:50-68; motion_autoencoder.py:107-149 fuse_convs / final_fuse) on the same patch
 * kernel: the chunk index runs over (time tap, 16 channels), the workgroup's image is a (sample, frame) pair and the input frame of a
 * chunk is t + kt - 1 (reflected / zero).  X NC8 = [N][ceil(C/8)][T][H][W][8] (c2m_nchw_to_nc8 with HW = T*H*W); A = three
 * c2m_pack_weights_bf16_patch images back to back (kt = 0, 1, 2: w + 9 kt, s_m = 27 C, s_c = 27); Y contiguous [N][M][T][H][W].
 * c2m_conv_wgrad3d_nc8: its weight gradient, one transposed-read launch per time tap (slab: c2m_conv_wgrad_nc8_slab_floats with N*T). */
int c2m_conv3d_nc8(const void* A, const void* X_nc8, void* Y, const float* bias, int M, int C, long N, int T, int H, int W, int reflect,
                   int yh, int act, float slope, void* stream);
/* Its data gradient: launched over the T real frames, frame t summing its (dY frame, time tap) pairs from the device table ptab
 * (int32 [T][11] = {npairs, (frame, kt) x 5}: reflect -- the pad frames folded onto the frames they mirror; zeros -- the in-range
 * pairs); target [N][M][T][H+2][W+2] (reflect: spatially padded gradient, c2m_reflect_fold(pt 0, ph 1, pw 1) finishes) or
 * [N][M][T][H][W] (zeros); the target may have Ctot >= M channels (M = the leading input channels that need a gradient).  A: three
 * pack images with rows = input channels (kt: w + 9 kt, s_m = 27, s_c = 27 Ctot).                                                */
int c2m_conv3d_dgrad_nc8(const void* A, const void* dY_nc8, void* target, const int* ptab, int M, int Ctot, int K, long N, int T, int H,
                         int W, int reflect, int th, void* stream);
int c2m_conv_wgrad3d_nc8(const void* dY_nc8, const void* X_nc8, float* slab, float* dW, float* db, int M, int C, long N, int T, int H,
                         int W, int reflect, void* stream);
/* Weight (+ bias) gradient from NC8 operands: s2 = 0: 2-D 3x3 stride-1 pad-1 layer, dW[m][c][ky][kx] = sum dY[n][m][y][x] *
 * X[n][c][y+ky-1][x+kx-1]; s2 = 1: 4x4 stride-2 pad-1 layer, ... * X[n][c][2y+ky-1][2x+kx-1] (X is the [2H][2W] map; one workgroup per
 * input-row parity, 8 taps each).  zeros or reflect padding; fragments by ds_read_b64_tr_b16 out of plain NC8 images in LDS; slab holds
 * c2m_conv_wgrad_nc8_slab_floats(...) floats of scratch (per-split partial sums, reduced in a fixed order); db may be NULL.  */
int c2m_conv_wgrad_nc8_splits(int M, int C, long N, int H, int W, int s2);
long c2m_conv_wgrad_nc8_slab_floats(int M, int C, long N, int H, int W, int s2);
int c2m_conv_wgrad_nc8(const void* dY_nc8, const void* X_nc8, float* slab, float* dW, float* db, int M, int C, long N, int H,
                       int W, int reflect, int s2, void* stream);

/* Winograd F(2x2,3x3) form of the 3x3 stride-1 layers (conv_wino.hip; same reference call sites as above: vgg.py:92-137,
 * spade_block.py:47-49, residual_block.py:13-71, up_block.py:9-13): 2.25x fewer MFMA FLOPs, fp32, bias/activation fused.
 * c2m_wino_filter_transform packs U = G g G^T in the kernel's fragment order (dgrad = 1: transposed + rotated filter of
 * the data gradient); upack holds c2m_wino_upack_floats(M, K) floats.
 * geom[]: 0 M, 1 K, 2 images, 3 Hi, 4 Wi, 5 Ho, 6 Wo, 7 iy0, 8 ix0 (input origin of output (0,0): -pad forward),
 *         9 reflect, 10 in_sn, 11 in_sc, 12 in_sh, 13 out_sn, 14 out_sc, 15 out_sh, 16 out_off, 17 x_bytes;
 *         with Y_interior (two-target data gradient of a reflect-padded conv, as in c2m_conv_igemm): 18 y2_sn, 19 y2_sc,
 *         20 y2_sh, 21 lo_y, 22 lo_x, 23 ext_y, 24 ext_x;
 *         3x3x3 layers (fuse_convs / the 3-D blocks of the motion decoder, modules/layers/common.py Conv3d call sites) run
 *         as a 2-D Winograd over virtual channels (time tap kt, channel ci), image = (sample, frame): 25 To (frames per
 *         sample of the output), 26 in_st, 27 out_st (frame strides), 28 cin, 29 nkt (0 = 2-D layer, else 3: K = nkt*cin),
 *         30 toff (source frame = t + kt + toff), 31 Ti (source frames), 32 treflect (reflect the source frame index;
 *         otherwise frames outside [0, Ti) are zeros);
 *         33 temporal pair table (device pointer as an integer, or 0): the data gradient of a 3x3x3 layer with reflect
 *         padding in time, launched over the UNPADDED frames -- int32 ptab[To][11] = {npairs, (source frame of dY, U block
 *         = flipped time tap) x 5}: output frame t sums the pairs (to, kt) with reflect(to + kt - 1) == t (needs cin % 8
 *         == 0; toff / treflect are ignored).  With the pair table every launched frame is a real one, so Y_interior may be
 *         given for a 3x3x3 layer too (frames of Y_interior are dense [ext_y][y2_sh] planes inside each channel).
 *         geom[] always holds 34 entries.                                                                           */
long c2m_wino_upack_floats(int M, int K);
int c2m_wino_filter_transform(const float* w, float* upack, int Cout, int Cin, int dgrad, void* stream);
/* Regions (workgroup tiles of <= 32 Winograd tiles) per image c2m_conv_wino uses for an Ho x Wo output domain: 8 x 16
 * outputs, or another th x tw tile shape when that covers a badly fitting domain (18 x 34: 9 -> 6) with <= 0.9x as many. */
int c2m_wino_regions(int Ho, int Wo);
int c2m_conv_wino(const float* upack, const float* X, float* Y, float* Y_interior, const float* bias,
                  const int64_t* geom, int act, float slope, void* stream);

/* Winograd F(4x4, 3x3) form of the same 2-D launches (conv_wino4.hip: 36 multiplies per 16 outputs, 4x fewer MFMA FLOPs than
 * the direct form; points 0, +-3/4, +-3/2, inf: fp32 error 2e-6 ... 4e-6 of the tensor scale): one workgroup = 64 rows x a
 * 16 x 32 output region.  upack: c2m_wino4_upack_floats(M, K) floats written by c2m_wino4_filter_transform (same arguments as
 * the F(2x2,3x3) transform); c2m_conv_wino4 takes c2m_conv_wino's geom[] and refuses its 3x3x3 entries (geom[29], geom[33] != 0).
 * Replaces the same call sites as c2m_conv_wino (ATen conv2d forward / data gradient of the 3x3 stride-1 layers,
 * layers/vgg.py:92-137, residual_block.py:13-31, spade_block.py:47-49).                                                 */
long c2m_wino4_upack_floats(int M, int K);
int c2m_wino4_filter_transform(const float* w, float* upack, int Cout, int Cin, int dgrad, void* stream);
int c2m_wino4_regions(int Ho, int Wo);
int c2m_conv_wino4(const float* upack, const float* X, float* Y, float* Y_interior, const float* bias,
                   const int64_t* geom, int act, float slope, void* stream);

/* Winograd weight gradient of the same layers: dg = G^T [ sum_tiles (A dY A^T) (.) (B^T d B) ] G (3x3, stride 1, pad 1,
 * H % 2 == 0, W % 8 == 0).  slab: c2m_wino_wgrad_splits(...) * 16 * M * K floats, dbslab: splits * M floats; dW in the
 * native [Cout][Cin][3][3] layout, db [Cout] (may be NULL).  Deterministic (fixed-order slab reduction).           */
int c2m_wino_wgrad_splits(int M, int K, int nimg, int H, int W);
int c2m_conv_wino_wgrad(const float* dY, const float* X, float* slab, float* dbslab, float* dW, float* db, int M, int K,
                        int nimg, int H, int W, int reflect, void* stream);
/* The same for the 3x3x3 stride-1 pad-1 layers (Conv3d call sites of modules/layers/common.py): image = (sample, frame),
 * virtual input channels (time tap, channel).  dY [N][M][T][H][W], X [N][Cin][T][H][W]; dW is written as [M][3][Cin][3][3]
 * (the caller permutes to the native [M][Cin][3][3][3]); slab / dbslab sized for K = 3*Cin, nimg = N*T
 * (c2m_wino_wgrad_splits(M, 3*Cin, N*T, H, W)). */
int c2m_conv_wino_wgrad3d(const float* dY, const float* X, float* slab, float* dbslab, float* dW, float* db,
                          int M, int Cin, int N, int T, int H, int W, int reflect, void* stream);

/* Pad-ring part of the data gradient of a reflect-padded (pad 1) 3x3 stride-1 convolution, folded straight into dX
 * (conv_ring.hip).  Replaces, together with a c2m_conv_wino / c2m_conv_wino4 launch over the EXACT H x W domain, the ATen chain
 * reflection_pad2d_backward(conv2d_backward_input(..)) of the `nn.Conv2d(.., padding=1, padding_mode='reflect')` call sites
 * (layers/residual_block.py:13-31,42-71, same_block.py:50-68, spade_block.py:47-49): dX [N][M][H][W] must already hold the
 * zero-padded "same" data gradient (the interior of the padded gradient); this launch adds what rows 0, H+1 and columns 0, W+1 of
 * the padded gradient mirror onto rows 1, H-2 and columns 1, W-2 -- four three-tap GEMMs M x 3C x (N * line) on the exact fp32
 * MFMA + one thread per (image, row, corner) for the four targets that receive three ring terms.  Every element of dX has one
 * writer: no atomics, bit-repeatable.  w: native [C][M][3][3] (C = the layer's output channels, M = its input channels);
 * apack: c2m_ring_pack_floats(C, M) floats written by c2m_ring_pack; dY [N][C][H][W]; H, W >= 4; tensors < 2 GiB.            */
long c2m_ring_pack_floats(int C, int M);
int c2m_ring_pack(const float* w, float* apack, int C, int M, void* stream);
int c2m_reflect_ring_dgrad(const float* apack, const float* w, const float* dY, float* dX, int N, int C, int M, int H, int W,
                           void* stream);
/* Buffer form (the one the product runs): the ring terms are WRITTEN -- coalesced, no read-modify-write of dX, corners carried by
 * the row terms -- into R [N][M][4][r_l] (sides 0 / 1: added to row 1 / H-2 at column x; 2 / 3: to column 1 / W-2 at row y;
 * r_l >= max(H, W), r_l % 4 == 0, 16-byte aligned), and the c2m_conv_wino / c2m_conv_wino4 launch that FOLLOWS on the same stream
 * over the exact domain adds them in its epilogue: geom[C2M_WG_RING] = R, geom[C2M_WG_RING_L] = r_l (2-D, single target,
 * out_off 0).  The in-place form spent 19 of its ~50 us per launch on one-float-per-128-byte-line accesses to dX's two columns. */
int c2m_reflect_ring_buffer(const float* apack, const float* dY, float* R, int N, int C, int M, int H, int W, int r_l,
                            void* stream);

/* Adjoint of reflection padding: folds a gradient over the padded domain back (ReflectionPad2d/3d backward);
 * _border_add is the in-place form used after a two-target dgrad (dX already holds the direct term).          */
int c2m_reflect_border_add(const void* dXpad, void* dX, long NC, int T, int H, int W, int pt, int ph, int pw,
                           int dt, void* stream);
int c2m_reflect_fold(const void* dXpad, void* dX, long NC, int T, int H, int W, int pt, int ph, int pw,
                     int dt, void* stream);

/* ---- normalisation + activation (norm.hip) -----------------------------------------------------------------
 * BatchNorm2d/3d(train) / InstanceNorm2d / SPADE + LeakyReLU/ReLU:
 *   down_block.py:19-22,44-47  same_block.py:19-22,41-44,64-67  up_block.py:11-12  residual_block.py:20-28,56-70
 *   spade_block.py:68-77.   mode 0 = per (n,c) plane, 1 = per channel over N*S.                              */
/* `dt` = element type of the ACTIVATION tensors (x, y, gy, dx, the SPADE maps gb / ggb): 0 fp32, 1 bf16 (csrc/dtype.h);
 * statistics, affine parameters and their gradients are fp32 in either case, all arithmetic is fp32 in registers.      */
long c2m_norm_workspace_floats(int N, int C, long S);
int c2m_norm_stats(const void* x, float* mean, float* invstd, float* running_mean, float* running_var,
                   float* workspace, int N, int C, long S, int mode, float eps, float momentum, int dt, void* stream);
/* c2m_norm_stats followed by c2m_norm_apply (without the NC8 side output) as ONE call: instance-norm planes (mode 0) of 1024 ...
 * 32768 elements run as one launch that keeps the plane in registers between the statistics and the apply pass (same arithmetic and
 * summation order as the three-launch path; bit-identical for planes of <= 8192 elements); everything else runs the three launches.
 * c2m_norm_bwd makes the same choice by itself for mode 0 without affine-parameter gradients and planes of <= 8192 elements.      */
int c2m_norm_set_fused(int on);       /* tests / A/B: 0 = always the three-launch path; returns the previous setting */
int c2m_norm_fwd(const void* x, float* mean, float* invstd, float* running_mean, float* running_var, float* workspace,
                 const float* gamma, const float* beta, const void* gb, void* y, int N, int C, long S, int mode, float eps,
                 float momentum, int act, float slope, int dt, void* stream);
/* y_nc8 / dx_nc8 (optional; bf16 tensors with S % 8 == 0): the result ALSO in the channel-blocked layout [N][ceil(C/8)][S][8] the
 * NC8 convolutions consume (c2m_nchw_to_nc8's), written by the same pass -- the activation feeds a convolution, the gradient is the
 * dY of the convolution in front.                                                                                              */
int c2m_norm_apply(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                   const void* gb, void* y, void* y_nc8, int N, int C, long S, int mode, int act, float slope, int dt, void* stream);
int c2m_norm_bwd(const void* x, const void* gy, const float* mean, const float* invstd, const float* gamma,
                 const float* beta, const void* gb, void* ggb, float* dgamma, float* dbeta, void* dx, void* dx_nc8,
                 float* workspace, int N, int C, long S, int mode, int act, float slope, int dt, void* stream);
int c2m_act_bwd(const void* y, const void* gy, void* gx, long total, int act, float slope, int dt, void* stream);
/* The same gradient (mode 0), or the perceptual-loss tap backward of c2m_relu_tap_bwd (mode 1: t, gl, count = the tap's element
 * count, gy may be NULL), written in the NC8 layout of the bf16 convolutions INSTEAD of NCHW ([N][ceil(C/8)][S][8] bf16; y / t / gy
 * bf16 [N][C][S], S % 8 == 0): for gradients whose only readers are NC8 convolution kernels (round 4; the layout pass and the NCHW
 * write disappear).  Same fp32 arithmetic per element as the NCHW kernels. */
int c2m_grad_to_nc8(int mode, const void* y, const void* t, const void* gy, const float* gl, void* g_nc8, long N, int C, long S,
                    long count, int act, float slope, void* stream);

/* ---- optical-flow warping / resampling (warp.hip) -----------------------------------------------------------
 * utils/ops.py:183-202 resample()/get_grid()/grid_sample() and its backward; callers generator.py:86,
 * motion_autoencoder.py:125 (fused "* occlusion"), losses.py:219, model.py:204,208.                          */
/* `dt` = element type of the image-side tensors (img, out, gout, gimg / in, out): 0 fp32, 1 bf16; flow, occlusion and the
 * flow gradient are fp32 (coordinates are never rounded to bf16).                                                  */
int c2m_flow_warp_fwd(const void* img, const float* flow, const float* occ, void* out, int N, int C, int H, int W,
                      int dt, void* stream);
/* backward of the above (ATen grid_sampler_2d_backward: a float-atomic scatter on GPUs).  Deterministic here: d(image) is
 * gathered through an inverted tap list built in `workspace`, d(flow) is summed in a fixed channel order; neither
 * output needs zero-initialisation and either may be NULL.                                                      */
long c2m_flow_warp_bwd_workspace_bytes(int N, int C, int H, int W, int want_gimg, int want_gflow);
int c2m_flow_warp_bwd(const void* img, const float* flow, const float* occ, const void* gout, void* gimg,
                      float* gflow, int N, int C, int H, int W, void* workspace, int dt, void* stream);
/* F.interpolate(bilinear) (utils/utils.py:349 align_corners=True; motion_autoencoder.py:123, up_block.py:10).  */
int c2m_resize_bilinear(const void* in, void* out, long NC, int Hi, int Wi, int Ho, int Wo, int align,
                        double scale_factor, int dt, void* stream);
int c2m_upsample2x_fwd(const void* in, void* out, long NC, int Hi, int Wi, int dt, void* stream);
/* The same up-sampling (up_block.py:10) from an NC8 tensor to an NC8 tensor ([N*ceil(C/8)][H][W][8] bf16 -> [..][2H][2W][8]), for the
 * up block whose convolution reads the channel-blocked form only (round 4); per channel the arithmetic of c2m_upsample2x_fwd. */
int c2m_upsample2x_nc8(const void* in_nc8, void* out_nc8, long NCB, int Hi, int Wi, void* stream);
int c2m_upsample2x_bwd(const void* gout, void* gin, long NC, int Hi, int Wi, int dt, void* stream);
/* torchvision.ops.roi_align(aligned=False, sampling_ratio=-1), appearance_encoder/appearance_encoder.py:67-69.
 * boxes [K,5] = (batch, x1, y1, x2, y2) stay on the device; the backward gathers per feature pixel in a fixed order (no
 * float atomics, no zero-initialisation of gfeat [N,C,H,W]).                                                       */
int c2m_roi_align_fwd(const float* feat, const float* boxes, float* out, int K, int C, int H, int W, int PH, int PW,
                      float spatial_scale, void* stream);
int c2m_roi_align_bwd(const float* boxes, const float* gout, float* gfeat, int N, int K, int C, int H, int W, int PH,
                      int PW, float spatial_scale, void* stream);
/* VGG-19 max pools (layers/vgg.py, torchvision features 4/9/18/27).                                           */
int c2m_maxpool2x2_fwd(const void* in, void* out, long NC, int Hi, int Wi, int dt, void* stream);
int c2m_maxpool2x2_bwd(const void* in, const void* gout, void* gin, long NC, int Hi, int Wi, int dt, void* stream);
/* The same for a window input that is the output of a ReLU (vgg.py: conv -> ReLU -> MaxPool2d): gin is the gradient of the
 * ReLU's INPUT (pool backward x (in > 0)), which spares the activation-backward pass over the full-resolution tensor. */
int c2m_maxpool2x2_relu_bwd(const void* in, const void* gout, void* gin, long NC, int Hi, int Wi, int dt, void* stream);

/* ---- FlowNet2 operators of the online target-flow path (flownet_ops.hip) -----------------------------
 * Replace the reference's CUDA extensions, forward only (the flow net runs frozen under no_grad, flow_net.py:32,66-67):
 * resample2d/src/resample2d_kernel.cu:16-75 (bilinear warp, pixel-unit flow, border clamp; caller flownet2/models.py:127,
 * 146,166,177), channelnorm/src/channelnorm_kernel.cu:19-62 (L2 norm over channels -> [N,1,H,W]; models.py:130,148,165,
 * 175), correlation/src/correlation_cuda_kernel.cu:47-147 (+ correlation_cuda.cc:25-38 output size; FlowNetC cost volume,
 * networks/flownet_c.py:44-46).  c2m_bias_act: bias + activation in place, the epilogue of the transposed convolutions
 * (networks/submodules.py:75-80), whose matrix part runs on c2m_conv_igemm's data-gradient form.                          */
int c2m_resample2d_fwd(const float* img, const float* flow, float* out, int N, int C, int H, int W, void* stream);
int c2m_channelnorm_fwd(const float* x, float* out, int N, int C, int H, int W, void* stream);
int c2m_correlation_out_size(int H, int pad, int kernel_size, int max_displacement, int stride1);
int c2m_correlation_fwd(const float* in1, const float* in2, float* out, int N, int C, int H, int W, int pad,
                        int kernel_size, int max_displacement, int stride1, int stride2, void* stream);
int c2m_bias_act(float* x, const float* bias, long N, int C, long HW, int act, float slope, void* stream);

/* ---- sparse-motion raster + occlusion splat (motion_raster.hip): bit-exact index/mask path ------------------
 * motion_estimator/dense_motion.py:94-168 generate_sparse_motion/warp/clip_mask;
 * utils/ops.py:205-275 get_occlusion_map/get_corresponding_map.                                              */
int c2m_sparse_raster(const float* instance, const int* obj_id, const int* obj_batch, const float* thetas, float* bw,
                      float* fw, float* bin, int B, int K, int T, int H, int W, void* stream);
long c2m_occlusion_splat_workspace_bytes(long nimg, int H, int W);
int c2m_occlusion_splat(const float* flow, long sb, long sc, long st, int B, int T, int H, int W, float* occ,
                        float* clip, void* workspace, void* stream);

/* ---- loss reductions (losses.hip) ---------------------------------------------------------------------------
 * losses/losses.py:180-189 L1MaskedLoss (also :60-65 VGG L1, model.py:118-121 feature matching); :152-177 SSIM. */
int c2m_l1_mean_fwd(const void* a, const void* b, const float* mask, float* out, long total, int C, long inner,
                    void* workspace /* 8 KiB */, int dt /* a, b */, void* stream);
int c2m_l1_mean_bwd(const void* a, const void* b, const float* mask, const float* gscale, void* ga, void* gb,
                    long total, int C, long inner, int dt /* a, b, ga, gb */, void* stream);
/* Backward of a perceptual-loss tap y = relu(conv(x)) that feeds the next VGG conv and mean|y - t| (losses.py:60-65 on the
 * relu{1..5}_1 slices of layers/vgg.py:92-137): out = (gy + gl[0]/total * sign(y - t)) * (y > 0) in one pass; gy may be NULL
 * (last tap), gl is a device scalar (the gradient of the L1 mean).  out is what the conv's data gradient consumes.       */
int c2m_relu_tap_bwd(const void* y, const void* t, const void* gy, const float* gl, void* out, long total,
                     int dt /* y, t, gy, out */, void* stream);
int c2m_ssim_fwd(const float* x, const float* y, float* out, long NC, int H, int W, void* workspace /* 8 KiB */,
                 void* stream);
int c2m_ssim_bwd(const float* x, const float* y, const float* gscale, float* gx, float* coef, long NC, int H, int W,
                 void* stream);

/* ---- object GNN (gnn.hip, round 5): the attention of torch_geometric's GATv2Conv(heads, concat=False, add_self_loops=False)
 * on a dense graph of <= 64 nodes, src/modules/motion_estimator/sparse_motion_estimator.py:104-112 (one layer per predicted frame).
 * xl = lin_l(x), xr = lin_r(x) [N][H][C]; att [H][C]; A [N][N] = number of edges j -> i at A[i][j] (mask and weight of the softmax
 * over the incoming edges); out [N][C] = mean over heads of sum_j alpha[i][j][h] * xl[j][h][:] (the layer's bias is added by the
 * caller); alpha [N][N][H] is kept for the backward.  One launch forward, two backward (per-target partials summed in index
 * order: no atomics).  N <= 64, C <= 1024, fp32.                                                                                */
int c2m_gat_dense_fwd(const float* xl, const float* xr, const float* att, const float* A, float* out, float* alpha, int N, int H,
                      int C, float negative_slope, void* stream);
long c2m_gat_dense_bwd_workspace_floats(int N, int H, int C);
int c2m_gat_dense_bwd(const float* xl, const float* xr, const float* att, const float* alpha, const float* gout, float* dxl,
                      float* dxr, float* datt, float* workspace, int N, int H, int C, float negative_slope, void* stream);

/* ---- input pipeline stage upstream of the path (data_prep.hip): SURVEY §8f-3 -------------------------------------
 * src/datasets/cityscapes.py:30-70 (ToTensor of frames, label-id one-hot split 0..10 / 11..19), :212-265 (occlusion
 * PNG -> clip_mask, .flo HWC -> CHW): decoded uint8 / float arrays in, the batch-dict tensors of model.py:124 out.  */
int c2m_prep_video(const uint8_t* frames_bthwc, float* video_bcthw, int B, int T, int H, int W, void* stream);
int c2m_prep_seg_onehot(const uint8_t* labels_bthw, float* bg_mask, float* fg_mask, int B, int T, int H, int W,
                        void* stream);
int c2m_prep_flow_occ(const uint8_t* occ_bthw, const float* flow_bthwc, float* occ_out, float* flow_out, int B, int T,
                      int H, int W, void* stream);

/* ---- measurement (events.hip): timing events without the system-scope fence of a default hipEventRecord; used by the
 * roofline measurement of bench.py (HIP events on the launch stream), never by the product path.        */
int c2m_event_create(void** event_out);
int c2m_event_record(void* event, void* stream);
int c2m_event_elapsed_ms(void* start, void* stop, float* ms);
int c2m_event_destroy(void* event);

/* ---- optimizer (optim.hip): SURVEY §8f-1 --------------------------------------------------------------------
 * The four torch.optim.Adam(betas=(0.5,0.999), eps=1e-7) of modules/model.py:54-99, stepped at trainer/trainer.py:155-165:
 * one launch per parameter group.  table = device int64 [4][ntensors] {param, grad, exp_avg, exp_avg_sq} pointers,
 * sizes = device int64 [ntensors], blockmap = device int32 [nblocks][2] (tensor, chunk of c2m_adam_chunk() elements).
 * step_size = lr / (1 - beta1^t) and bias_correction2_sqrt = sqrt(1 - beta2^t) are computed by the caller in double,
 * exactly as ATen's single-tensor Adam does; the kernel mirrors its fp32 operation order.                         */
int c2m_adam_chunk(void);
int c2m_adam_step(const int64_t* table, const int64_t* sizes, const int32_t* blockmap, int ntensors, int nblocks,
                  double beta1, double beta2, double eps, double step_size, double bias_correction2_sqrt, void* stream);

#ifdef __cplusplus
}
#endif
#endif
