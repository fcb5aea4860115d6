/* gdrn_hip.h -- C ABI of libgdrn_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for GDR-Net's
 * per-RoI hot path (ResNet-34 backbone + geometric head + Patch-PnP + pose decode + losses, forward
 * and backward).
 *
 * This is the drop-in boundary for the path.  The reference has no native interface here: its hot
 * path is Python dispatching ATen/cuDNN ops.  Each entry point below names the
 * reference call site(s) whose ATen dispatch it replaces; paths are relative to the reference
 * checkout.  Binding stub for a reference maintainer: INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-owned device memory (the library never
 *     allocates or frees device memory and keeps no state besides per-kernel launch attributes);
 *   - every call enqueues asynchronously on `stream` (a hipStream_t passed as void*; 0 = null stream)
 *     and returns an int status: 0 ok, <0 error (GDRN_ERR_*); nothing throws across the boundary;
 *   - activations are NHWC with an explicit pixel stride (`*_cs`, in elements); `dtype` selects the
 *     storage/operand type: GDRN_DT_F32 (fp32 MFMA, parity mode), GDRN_DT_BF16 or GDRN_DT_F16 (16-bit MFMA operands,
 *     fp32 accumulate, throughput mode; which of the two: the library build).  Statistics, losses, pose and parameter gradients are fp32.
 *   - re-entrant and thread-compatible: one host thread per stream.
 */
#ifndef GDRN_HIP_H
#define GDRN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: status / dtype enums renamed (GDRN_E_* -> GDRN_ERR_*, GDRN_F32 / GDRN_BF16 -> GDRN_DT_*; the old names stay as deprecated aliases),
 *    gdrn_conv_params.pad0_ became w_frag (values outside 0..2 are rejected), gdrn_wgrad_params.variant is honoured (GDRN_WGRAD_W128). */
/* 3: gdrn_conv_params grew halo_waves (appended; zero = the behaviour of version 2). */
/* 4: entry points added (nothing changed): gdrn_loss_scale_state + gdrn_ranger_multi_dyn / gdrn_loss_scale_update / gdrn_unscale_or_zero /
 *    gdrn_scaled_loss_weights -- the fp16 mode's dynamic loss scale decided on the device; gdrn_bn_relu_upsample2x_fwd, gdrn_upsample2x_bwd_bnsums; gdrn_block64_eval; gdrn_conv3x3s2 / gdrn_conv3x3s2_dgrad (+ gdrn_s2_params, gdrn_s2d_params).
 *    Removed: gdrn_conv3x3_wgrad_multi_w128 and gdrn_wgrad_params.variant = GDRN_WGRAD_W128 (the 128 x 64 weight-gradient tile of rounds 4-5:
 *    never faster inside the step). */
/* 5: gdrn_s2d_params grew stats / bias / act (appended; NULL / 0 = the data gradient of version 4): gdrn_conv3x3s2_dgrad also runs the forward pass of
 *    the head's ConvTranspose2d.  gdrn_conv3x3s2 / gdrn_conv3x3s2_dgrad accept maps 8 pixels wide (two images per pixel tile: an even image count). */
#define GDRN_ABI_VERSION 5
/* `dtype` arguments.  The 16-bit format is a property of the library build: libgdrn_hip.so computes GDRN_DT_BF16, libgdrn_hip_f16.so (the same
 * sources compiled with -DGDRN_HALF_F16: v_mfma_f32_*_f16, IEEE-half storage -- the arithmetic of the reference's fp16 autocast,
 * core/gdrn_modeling/main_gdrn.py:53-56,141, gdrn_evaluator.py:568) computes GDRN_DT_F16; each rejects the other's code with GDRN_ERR_ARG,
 * both compute GDRN_DT_F32.  gdrn_half_format() tells a host which one it opened. */
enum { GDRN_DT_F32 = 0, GDRN_DT_BF16 = 1, GDRN_DT_F16 = 2 };
/* status codes (0 = GDRN_OK) */
enum { GDRN_OK = 0, GDRN_ERR_ARG = -1, GDRN_ERR_SHAPE = -2, GDRN_ERR_LAUNCH = -3 };
/* deprecated aliases of ABI version 1 */
enum { GDRN_F32 = GDRN_DT_F32, GDRN_BF16 = GDRN_DT_BF16, GDRN_E_ARG = GDRN_ERR_ARG, GDRN_E_SHAPE = GDRN_ERR_SHAPE, GDRN_E_LAUNCH = GDRN_ERR_LAUNCH };
/* gdrn_wgrad_params.variant */
enum { GDRN_WGRAD_T64 = 0, GDRN_WGRAD_W128 = 1 /* removed in ABI 4: rejected by gdrn_conv3x3_wgrad_ok */ };

int gdrn_version(void);
/* Diagnostics: the HIP error code (hipError_t) behind the calling thread's most recent GDRN_ERR_LAUNCH, its name copied into `name` (cap bytes,
 * may be NULL); 0 if no launch of this thread has failed.  (ABI 3)  The call CLEARS the slot (ABI 4: a code is reported once, with the failure it
 * belongs to; every launch starts with a clean slot).  Without a launch error `name` reads "hipSuccess", or "stale:<name>" when a launch of this
 * thread found and dropped an error an earlier HIP call of the process had left behind. */
int gdrn_last_hip_error(char* name, int cap);
int gdrn_half_format(void);   /* GDRN_DT_BF16 or GDRN_DT_F16: the 16-bit dtype code this library build accepts */
/* Bytes of caller-owned scratch an entry point needs for the given call (every workspace of the ABI is caller-allocated device
 * memory, no initialisation needed unless stated): `op` selects the buffer, `params` points at the arguments that determine its
 * size.  < 0: GDRN_ERR_*.  (the Python engine sizes its plan buffers with the same rules.) */
enum {
    GDRN_WS_CONV_STATS = 0,     /* const gdrn_conv_params*  -> p->stats of gdrn_conv_gemm */
    GDRN_WS_CONV3X3_STATS = 1,  /* const gdrn_conv_params*  -> p->stats / p->bnb_rows of gdrn_conv3x3_halo */
    GDRN_WS_CONV3X3_WGRAD = 2,  /* const gdrn_wgrad_params* -> p->ws of gdrn_conv3x3_wgrad / one task of gdrn_conv3x3_wgrad_multi */
    GDRN_WS_STEM_WGRAD = 3,     /* const int* N             -> ws of gdrn_stem_wgrad */
    GDRN_WS_STEM_STATS = 4,     /* const int* N             -> stats of gdrn_stem_conv */
    GDRN_WS_LINEAR_SPLITK = 5,  /* const int[2] {M, N}      -> ws of gdrn_linear_splitk */
    GDRN_WS_BN_BWD_ROWS = 6     /* const long long[3] {npix, C, dtype} -> rows of gdrn_bn_bwd_reduce */
};
long long gdrn_workspace_bytes(int op, const void* params);
/* fills name (<=255 chars), compute units and the gcn arch string of device `dev`. */
int gdrn_device_info(int dev, char* name, int* cus, char* arch);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM gather convolution (forward conv, data gradient, transposed stride-2 conv, linear).
 * Replaces F.conv2d / F.conv_transpose2d / F.linear dispatched from
 *   resnet_backbone.py:23,69-80 + torchvision BasicBlock (twin: pvnet_net/resnet.py:44-74),
 *   cdpn_rot_head_region.py:81-135,183-185, conv_pnp_net.py:76-92,140-156,
 * and their autograd data-gradients (engine.py:279).
 *   mode 0: y[m][co] = sum_{ky,kx,c} x[n][oy*stride-pad+ky][ox*stride-pad+kx][c] * w[co][ky*KW+kx][c]
 *           m = (n*Ho+oy)*Wo+ox, M = N*Ho*Wo
 *   mode 1: transposed stride-2 gather (ConvTranspose2d forward / stride-2 data gradient):
 *           y[n][oy][ox][co] = sum x[n][(oy+pad-ky)/2][(ox+pad-kx)/2][c] * w[co][ky*KW+kx][c] over the
 *           taps where both divisions are exact; M = N*(Ho/2)*(Wo/2) (rows per parity class)
 *   w: packed [w_rows][KH*KW][Cin] of dtype, zero padded to w_rows (multiple of the N tile, see
 *      gdrn_conv_tile).  Cin*sizeof(dtype) must be a multiple of 128.
 *   epilogue: + bias[co] (fp32), + addend[m][co] (dtype), act (0 none, 1 ReLU, 2 LeakyReLU 0.1);
 *   stats != NULL: per-M-tile partial sum / sum of squares of the raw accumulators,
 *      stats[tile][0][co], stats[tile][1][co] (gdrn_conv_stats_rows tiles) for the following BatchNorm.
 *   bnb_x != NULL (gdrn_conv3x3_halo only, data-gradient launches): y is the gradient w.r.t. the OUTPUT of a
 *      BatchNorm(+ReLU) whose raw input was bnb_x[m][co] (channel stride bnb_cs).  The epilogue then also applies the
 *      ReLU mask -- (bnb_mask[m][co] > 0) when bnb_mask is given, else (bnb_x*bnb_scale + bnb_shift > 0) when
 *      bnb_scale/bnb_shift are given, else none -- stores the MASKED gradient and writes the two BatchNorm-backward sums
 *      of its pixel tile to bnb_rows[tile][2][Cout] (gdrn_conv3x3_stats_rows tiles, plain stores); gdrn_bn_bwd_coef
 *      turns the rows into the BatchNorm backward's coefficients: the separate reduction pass over (dy, x, mask) disappears.
 *   xf_mode != 0 (gdrn_conv3x3_halo only): the conv's INPUT is v(x, x2) evaluated per element while the patch is staged
 *      in LDS, rounded to bf16, zero outside the image (the padding applies to v, not to x), with per-input-channel
 *      fp32 vectors [Cin] (NULL a / b = 1, NULL c2 = 0):
 *        1: v = a*x + c                             BatchNorm(+ReLU) forward apply of the producer (a = scale, c = shift)
 *        2: v = (a*x + c) + bf16(b*x2 + c2)         ... with a residual (b = 1, c2 = 0) or a second normalised branch (the
 *                                                   BasicBlock downsample path, rounded as the pass that materialised it did)
 *        3: v = a*x + (b*x2 + c)                    BatchNorm backward apply: x = masked dy, x2 = the BN's raw input,
 *                                                   (a, b, c) from gdrn_bn_bwd_coef
 *        4: v = a*(x2*msc + msh > 0 ? x : 0) + (b*x2 + c)   the same with the ReLU mask recomputed from the forward affine
 *      then v = max(v, 0) when xf_relu.  x2 has the geometry and channel stride of x.  xf_out (nullable, same geometry
 *      and channel stride as x): v of every in-image pixel is also stored there once, so the tensor exists for the
 *      weight-gradient launch / the next residual without a separate pass.  Replaces gdrn_bn_apply /
 *      gdrn_bn_bwd_apply launches between two halo convs (BasicBlock, cdpn_rot_head_region.py:103-123 and backward).
 */
/* w_frag (gdrn_conv3x3_halo only): layout of w -- 0 / 1: gdrn_pack_wfrag (16-row fragments, first halo kernel), 2: gdrn_pack_wfrag32
 * (second-generation kernel, see gdrn_pack_wfrag32 below). */
/* halo_waves (gdrn_conv3x3_halo, w_frag 0 / 1; ABI 3): 0 = the library picks, 4 / 8 = force the four- / eight-wave form of the 128-channel tile
 * (gdrn_conv3x3_halo_waves below); v3_min_wg (w_frag 2): smallest grid the second-generation kernel gives its 16x16x256 tile, 0 = the default 256. */
typedef struct gdrn_conv_params {
    const void* x;
    const void* w;
    void* y;
    const float* bias;
    const void* addend;
    float* stats;
    const void* bnb_x;
    const void* bnb_mask;
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb_scale;
    const float* bnb_shift;
    float* bnb_rows;
    int bnb_cs;
    int w_frag;
    int Hi, Wi, Cin, x_cs;
    int Ho, Wo, Cout, y_cs, add_cs;
    int KH, KW, stride, pad;
    int mode, act, out_f32;
    int M, w_rows, dtype;
    int xf_mode, xf_relu;
    const void* xf_x2;
    const float* xf_a;
    const float* xf_b;
    const float* xf_c;
    const float* xf_c2;
    const float* xf_msc;
    const float* xf_msh;
    void* xf_out;
    int halo_waves;
    int v3_min_wg;
} gdrn_conv_params;
int gdrn_conv_gemm(const gdrn_conv_params* p, void* stream);
int gdrn_conv_tile(const gdrn_conv_params* p, int* bm, int* bn);
/* ResNet stem, bf16: 7x7 stride-2 pad-3 conv 3 -> 64 on the NHWC4 canvas of gdrn_pack_image (256x256 images ->
 * [N][262][272][4]), one kernel row = one 32-deep MFMA k-step, pixel fragments straight from global memory, weights in
 * registers (resnet_backbone.py:23,69).  w32 = gdrn_pack_stem_w32(OIHW fp32 weight) = bf16 [64][7][32];
 * y = [N][128][128][64]; stats (nullable) = [gdrn_stem_stats_rows(N)][2][64] partial sums for gdrn_bn_finalize. */
int gdrn_pack_stem_w32(const float* w, void* dst, int dtype, void* stream);
int gdrn_stem_stats_rows(int N);
int gdrn_stem_conv(const void* canvas, const void* w32, void* y, float* stats, int N, int dtype, void* stream);
/* Eval mode (module.eval(): BatchNorm on its running statistics): the stem conv, bn1 as per-channel scale / shift (gdrn_bn_eval_params), ReLU and
 * the 3x3 stride-2 max-pool of resnet_backbone.py:69-72 in one pass -- y = [N][64][64][64], bit-identical to gdrn_stem_conv followed by
 * gdrn_bn_relu_maxpool_fwd, without the 134 MB round trip of the conv output (ABI 3). */
/* Eval mode, 64 regions: the geometric head's 1x1 output conv (nn.Conv2d(256, 69, 1), cdpn_rot_head_region.py:127-135; w = weight rows
 * [>= 69][256] in the 16-bit format, fp32 bias) and gdrn_head_tail_fwd (GDRN.py:156-169) in one launch: the fp32 logits go to `head` only if it
 * is non-NULL ([N*HW][hs >= 72]; callers that return the maps), pnp_in as gdrn_head_tail_fwd writes it (ABI 3). */
int gdrn_head_conv_tail_fwd(const void* x, int x_cs, const void* w, int w_rows, const float* bias, const float* coord2d, const float* extents,
                            float* head, int hs, void* pnp_in, int pcs, int N, int HW, int nreg, int dtype, void* stream);
/* ... and with do_loss=True (GDRN.py:345-400) the same pass also accumulates the map-loss sums as gdrn_head_tail_loss_fwd does, one partial row per
 * workgroup: acc = 8 + 8 * gdrn_head_conv_tail_loss_rows(N, HW) doubles, finished by gdrn_map_loss_finalize_rows; head is mandatory (the backward
 * pass reads the logits). */
int gdrn_head_conv_tail_loss_rows(int N, int HW);
int gdrn_head_conv_tail_loss_fwd(const void* x, int x_cs, const void* w, int w_rows, const float* bias, const float* coord2d, const float* extents,
                                 float* head, int hs, void* pnp_in, int pcs, const float* gt_xyz, const float* mask_visib, const float* mask_trunc,
                                 const long long* gt_region, double* acc, int N, int HW, int nreg, int dtype, void* stream);
/* Data gradient of that 1x1 output conv (autograd, core/gdrn_modeling/engine.py:279) with the ReLU mask and the two BatchNorm-backward sums of the
 * BatchNorm(+ReLU) in front of it in the epilogue (cdpn_rot_head_region.py:120-135): dy = d_logits [N*HW][dy_cs >= 96] (channels >= 69 zero), wd = operand
 * rows [256][wd_cs >= 96] (row ci = W[:, ci]), raw / mean / invstd / scale / shift = that BatchNorm's raw input, batch statistics and forward affine
 * (mask: scale * raw + shift > 0), dx [N*HW][dx_cs] = the masked gradient, rows = [gdrn_head_out_dgrad_rows(N, HW)][2][256] for gdrn_bn_bwd_coef (ABI 3). */
int gdrn_head_out_dgrad_rows(int N, int HW);
int gdrn_head_out_dgrad(const void* dy, int dy_cs, const void* wd, int wd_cs, const void* raw, int raw_cs, const float* mean, const float* invstd,
                        const float* scale, const float* shift, void* dx, int dx_cs, float* rows, int N, int HW, int dtype, void* stream);
int gdrn_stem_conv_pool(const void* canvas, const void* w32, const float* scale, const float* shift, void* y, int N, int dtype, void* stream);
/* Stem weight gradient (backward-weight of nn.Conv2d(3, 64, 7, 2, 3), resnet_backbone.py:23, implicit in engine.py:279) fused with
 * the BatchNorm-backward apply in front of it: the stem has no data gradient, so dy = a*g + (b*raw + c) per channel
 * (gdrn_bn_bwd_apply's formula, rounded to bf16) is evaluated while the tile is staged instead of being written and re-read.
 *   canvas [N][262][272][4] bf16; g, raw [N][128][128][64] bf16 (masked upstream gradient / the conv output BatchNorm saw);
 *   a, b, c [64] from gdrn_bn_bwd_coef.  a == NULL: plain weight gradient with dy = g (raw, b, c ignored).
 *   ws: gdrn_stem_wgrad_parts(N) * 64 * 224 floats of scratch; grad: fp32 OIHW [64][3][7][7], overwritten.  bf16 only. */
int gdrn_stem_wgrad_parts(int N);
int gdrn_stem_wgrad(const void* canvas, const void* g, const void* raw, const float* a, const float* b, const float* c, int N,
                    float* ws, float* grad, int dtype, void* stream);
/* Skinny-M linear layer (M <= 64 rows, bf16): y[m][n] = act(sum_k x[m][k]*w[n][k] + bias[n]) with the K range split
 * over workgroups (the layer is bound by reading w once).  Replaces F.linear + LeakyReLU of Patch-PnP's fc1
 * (conv_pnp_net.py:85-92,152) where the gather kernel would run 8 workgroups.  x_rs / w_rs / y_rs: row strides in
 * elements; ws: GDRN_LINEAR_MAX_SPLITS*M*N floats (per-split partial slabs, no initialisation needed). */
#define GDRN_LINEAR_MAX_SPLITS 16
#define GDRN_LINEAR_NO_FINISH (-1)   /* (ABI 5) act: leave the gdrn_linear_splits(K, N) partial slabs [split][M][N] in ws, no bias / activation / y */
int gdrn_linear_splits(int K, int N);
int gdrn_linear_splitk(const void* x, const void* w, const float* bias, void* y, int M, int K, int N, int x_rs, int w_rs,
                       int y_rs, int act, float* ws, int dtype, void* stream);
int gdrn_conv_stats_rows(const gdrn_conv_params* p);
/* Halo-tiled variant for KH=KW=3, stride 1, pad 1, mode 0, H and W multiples of 8 (same params / epilogue contract):
 * the input patch of a TH x TW pixel tile is staged in LDS once per 128-byte channel chunk and the nine taps read
 * it at shifted offsets.  gdrn_conv3x3_tile reports (th, tw, bn), th = 0 when the shape is not covered;
 * gdrn_conv3x3_stats_rows the number of per-tile partial-statistics rows it writes. */
/* p->w of gdrn_conv3x3_halo is the FRAGMENT-MAJOR operand: gdrn_pack_wfrag permutes the 16-byte granules of the
 * row-major [rows][9][Cin] operand into one contiguous 1 KiB block per (16 rows, tap, 128-byte chunk, k-step) =
 * exactly the 64 lanes of an MFMA A operand, so the kernel streams weights L2 -> registers without LDS. bf16 only.
 * Row order: for operands of more than 64 rows (128-channel tile) the two 16-row blocks of a 32-row group interleave in units of
 * 4 rows, so that an MFMA result lane holds 8 contiguous output channels and the conv's epilogue moves 16 bytes per access. */
int gdrn_pack_wfrag(const void* src, void* dst, int rows, int Cin, int dtype, void* stream);
/* Second-generation halo kernel (conv3x3_v3.hip; Cin a multiple of 64, Cout of 128, W of 16, H of 8, bf16, act <= 1, bf16 output):
 * v_mfma_f32_32x32x16_bf16, the weights stream global -> LDS by LDS-DMA and are shared by the 8 waves of a workgroup, tiles of
 * 16x16 pixels x 256 channels (large maps) or 8x16 pixels x 128 channels with the K range split over two wave groups (small maps).
 * Selected by p->w_frag = 2; w is then the operand of gdrn_pack_wfrag32: one contiguous 1 KiB block = the 64 lanes of a 32x32x16
 * MFMA A operand per (128-byte chunk kc, tap, 16-deep k-substep ks, 32-row fragment f), block index ((kc*9 + tap)*4 + ks)*(rows/32) + f;
 * lane l of a block holds fragment row l & 31, k = 16*ks + 8*(l >> 5) .. +7.  Fragment f, row r is operand row
 * (f>>1)*64 + ((r>>2)&1)*32 + (f&1)*16 + (r>>3)*4 + (r&3): an MFMA result lane then holds 32 contiguous output channels per fragment
 * pair (64-byte epilogue accesses).  rows must be a multiple of 64.
 * gdrn_conv3x3_wfrag(p): operand layout for the shape in p: 2, 1, or 0 = no halo tiling.  p->w_frag is the caller's policy here: 0 = the
 * library's preference (the second-generation kernel where it measured faster), 1 = never that kernel, 2 = wherever it covers the shape.
 * The library reads no environment variables and keeps no configuration state: every choice it makes is a function of the params. */
int gdrn_pack_wfrag32(const void* src, void* dst, int rows, int Cin, int dtype, void* stream);
int gdrn_conv3x3_wfrag(const gdrn_conv_params* p);
int gdrn_conv3x3_halo(const gdrn_conv_params* p, void* stream);
int gdrn_conv3x3_tile(const gdrn_conv_params* p, int* th, int* tw, int* bn);
int gdrn_conv3x3_stats_rows(const gdrn_conv_params* p);
/* The first halo kernel's 128-channel tile exists in two forms: four waves, each walking the whole reduction of its 32 channels, and (ABI 3)
 * eight waves -- a second copy of the four takes k-step 1 of every tap stage, the first k-step 0, the halves are added through LDS in front
 * of the epilogue: two waves per SIMD for launches that give a CU a single workgroup (the 8x8 - 16x16 maps of resnet_backbone.py:53-80 at
 * 64 RoIs).  p->halo_waves = 0 lets the library pick (eight when the grid has <= 256 workgroups), 4 / 8 force one.
 * gdrn_conv3x3_halo_waves(p): waves per workgroup of the launch gdrn_conv3x3_halo makes for p (4 or 8; 0: w_frag = 2 or shape not covered). */
int gdrn_conv3x3_halo_waves(const gdrn_conv_params* p);

/* Weight gradient: dw[co][tap][ci] (fp32, packed, pre-zeroed by the caller) +=
 *   sum_m dy[m][co] * x[pix(m,tap)][ci]  (mode-0 gather; bf16 fragments through the LDS transpose read).  variant: 0 (only gdrn_conv3x3_wgrad* read it, see GDRN_WGRAD_W128).
 *   splits <= 0: automatic pixel-range split.
 * ws (gdrn_conv3x3_wgrad only, else NULL): per-split partial tiles go here (plain stores) instead of atomics on dw.
 * Replaces the autograd weight-gradients of the same layers (engine.py:279). */
typedef struct gdrn_wgrad_params {
    const void* x;
    const void* dy;
    float* dw;
    float* ws;
    int Hi, Wi, Cin, x_cs;
    int Ho, Wo, Cout, dy_cs;
    int KH, KW, stride, pad;
    int M, dtype, splits, variant;
} gdrn_wgrad_params;
int gdrn_conv_wgrad(const gdrn_wgrad_params* p, void* stream);
/* Halo-tiled variant for KH=KW=3, pad 1, bf16, Cin and Cout multiples of 64; stride 1 (Ho, Wo multiples of 8) or stride 2 (Hi = 2*Ho,
 * Wi = 2*Wo, Ho a multiple of 4, Wo of 8: the ResNet stage-entry convs, Patch-PnP's convs, and ConvTranspose2d(3, 2, 1, 1) with x = the
 * gradient of its output and dy = its input): a workgroup
 * accumulates a 64 x 64 (co x ci) tile of all nine taps from one staged 8x8 pixel patch per stage.  Same dw layout and
 * accumulate-with-atomics contract.  gdrn_conv3x3_wgrad_ok returns 1 when the shape is covered.
 * With p->ws != NULL the partial tiles of the gdrn_conv3x3_wgrad_splits(p) pixel-range splits are stored to
 * ws[split][Cout*Cin*9] (fragment order, no atomics, dw unused) and gdrn_wgrad_reduce_multi sums them for a whole
 * table of layers in one launch, writing dst[co*s_co + ci*s_ci + tap*s_t] (e.g. the OIHW .grad: s_co = Cin*9, s_ci = 9,
 * s_t = 1).  blk_start[i] = sum over tasks j < i of Cout_j*Cin_j/256; nblocks = blk_start[ntasks]. */
int gdrn_conv3x3_wgrad(const gdrn_wgrad_params* p, void* stream);
int gdrn_conv3x3_wgrad_ok(const gdrn_wgrad_params* p);
int gdrn_conv3x3_wgrad_splits(const gdrn_wgrad_params* p);
/* Grouped launch: the weight gradients of ntasks layers in one grid.  tasks_dev: device array of gdrn_wgrad_params with
 * ws != NULL and splits = gdrn_conv3x3_wgrad_splits() of an explicit request (no empty split);
 * blk_start[i] = sum_{j<i} (Cout_j/64)*(Cin_j/64)*splits_j, nblocks = blk_start[ntasks]. */
int gdrn_conv3x3_wgrad_multi(const gdrn_wgrad_params* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream);
/* the same with an LDS request of lds_bytes (> 64 KiB: one workgroup per CU, the rest of the CU stays free for another stream's kernels) */
int gdrn_conv3x3_wgrad_multi_lds(const gdrn_wgrad_params* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, int lds_bytes, void* stream);
/* cin_valid: input channels the parameter really has (0 = Cin): padded operand channels are skipped */
typedef struct gdrn_wreduce_task {
    const float* ws;
    float* dst;
    int nsplit, Cout, Cin, cin_valid;
    long long s_co, s_ci, s_t;
} gdrn_wreduce_task;
int gdrn_wgrad_reduce_multi(const gdrn_wreduce_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weight / layout packing.
 * gdrn_pack4: dst[a1][a2][t][b] (contiguous, A1 x A2 x T x B, dtype) = src[a1*s1 + a2*s2 + t'*st + b*sb]
 *   (fp32), zero where a1>=A1v, a2>=A2v or b>=Bv; t' = flip ? T-1-t : t.
 * gdrn_unpack4: the inverse scatter for gradients: src_grad[...] = packed[a1][a2][t][b] (fp32 -> fp32),
 *   only valid indices are written.
 * Used to turn the reference's OIHW / (Cin,Cout,kH,kW) / (out,in) parameter tensors
 * (state_dict schema) into the kernels' [rows][tap][Cin] operand layout. */
int gdrn_pack4(const float* src, void* dst, int A1, int A2, int T, int B, int A1v, int A2v, int Bv,
               long long s1, long long s2, long long st, long long sb, int flip, int dtype, void* stream);
int gdrn_unpack4(const float* packed, float* dst, int A1, int A2, int T, int B, int A1v, int A2v, int Bv,
                 long long s1, long long s2, long long st, long long sb, int flip, void* stream);
/* stem: conv1.weight (64,3,7,7) <-> [64][7][16 px * 4 ch]; image NCHW fp32 -> zero-padded NHWC4 */
int gdrn_pack_stem_w(const float* w, void* dst, int dtype, void* stream);
int gdrn_unpack_stem_w(const float* packed, float* dw, void* stream);
int gdrn_pack_image(const float* img, void* dst, int N, int H, int W, int Hp, int Wp, int dtype, void* stream);
/* generic casts between fp32 and dtype (n elements) */
int gdrn_cast_from_f32(const float* src, void* dst, long long n, int dtype, void* stream);
int gdrn_cast_to_f32(const void* src, float* dst, long long n, int dtype, void* stream);
/* NHWC (dtype, pixel stride cs) -> NCHW fp32 for handing maps back to the caller */
int gdrn_nhwc_to_nchw_f32(const void* src, int cs, int c0, int C, float* dst, int N, int HW, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (train / eval), fused with ReLU, residual add and max-pool where the graph has them.
 * Replaces nn.BatchNorm2d + nn.ReLU + `out += identity` + nn.MaxPool2d at resnet_backbone.py:24-26,
 * BasicBlock, cdpn_rot_head_region.py:92-93,113-114 and their backward. */
/* partial: [rows][2][C] per-tile sums / sums of squares from a conv epilogue; one launch of C/4 workgroups whatever the row
 * count (fp64 accumulation).  ws: unused (kept for ABI stability; was a 64*2*C-double workspace of a two-launch scheme). */
int gdrn_bn_finalize(const float* partial, int rows, int C, double count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, long long* num_batches_tracked, float momentum,
                     float eps, float* mean, float* invstd, float* scale, float* shift, double* ws, void* stream);
int gdrn_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, void* stream);
int gdrn_bn_apply(const void* x, const float* scale, const float* shift, const void* residual, void* y,
                  long long npix, int C, int relu, int dtype, void* stream);
/* BatchNorm backward = (1) per-channel sums of g and g*xhat, g = dy * (ymask > 0) * (x*mask_scale + mask_shift > 0) (each mask
 * optional), as partial rows rows[r][0][c] = sum g, rows[r][1][c] = sum g*xhat: one row per workgroup of gdrn_bn_bwd_reduce
 * (gdrn_bn_bwd_reduce_rows(npix, C, dtype) <= 1024 of them, plain stores: deterministic, nothing to pre-zero), or one per pixel
 * tile straight from the epilogue of the data-gradient conv that produced dy (gdrn_conv_params.bnb_*);
 * (2) gdrn_bn_bwd_coef: rows -> dgamma, dbeta and the coefficients (a, b, c) of dx = a*g + (b*x + c);
 * (3) the apply pass gdrn_bn_bwd_apply (optional g_out = g), or no pass at all when the consumer of dx is a halo conv
 * (xf_mode 3 / 4) or the stem weight gradient (gdrn_stem_wgrad), which evaluate it while staging their operand.
 * mask_scale/mask_shift (both or neither): the ReLU mask recomputed from the forward affine of a BN->ReLU without residual
 * instead of reading the stored activation (one tensor pass less). */
int gdrn_bn_bwd_reduce_rows(long long npix, int C, int dtype);
int gdrn_bn_bwd_reduce(const void* dy, const void* ymask, const void* x, const float* mean, const float* invstd,
                       const float* mask_scale, const float* mask_shift, long long npix, int C, float* rows, int dtype,
                       void* stream);
int gdrn_bn_bwd_apply(const void* dy, const void* ymask, const void* x, const float* a, const float* b, const float* c,
                      const float* mask_scale, const float* mask_shift, long long npix, int C, void* dx, void* g_out,
                      int dtype, void* stream);
/* rows [nrows][2][C] (fp64 accumulation, one launch of C/4 workgroups) -> a = gamma*invstd, b = -a*invstd*sum(g*xhat)/npix,
 * c = -a*sum(g)/npix - b*mean; dgamma = sum g*xhat, dbeta = sum g (both or neither NULL); npix = elements per channel. */
int gdrn_bn_bwd_coef(const float* rows, int nrows, int C, long long npix, const float* gamma, const float* mean,
                     const float* invstd, float* a, float* b, float* c, float* dgamma, float* dbeta, void* stream);
int gdrn_bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, unsigned char* idx,
                             int N, int H, int W, int C, int dtype, void* stream);
/* g = gradient w.r.t. the BatchNorm OUTPUT in front of the ReLU + max-pool (ReLU mask recomputed from x*scale + shift).
 * rows != NULL (then mean, invstd too): also that BatchNorm's backward sums of g, one partial row [2][C] per workgroup
 * (gdrn_maxpool_bwd_rows(N, H, W, C, dtype) rows, for gdrn_bn_bwd_coef) -- the separate gdrn_bn_bwd_reduce pass over g and x
 * disappears. */
int gdrn_maxpool_bwd_rows(int N, int H, int W, int C, int dtype);
int gdrn_maxpool_bwd(const void* dy, const unsigned char* idx, const void* x, const float* scale, const float* shift,
                     void* g, int N, int H, int W, int C, const float* mean, const float* invstd, float* rows, int dtype,
                     void* stream);

/* nn.UpsamplingBilinear2d(scale_factor=2) (align_corners=True), cdpn_rot_head_region.py:102 */
int gdrn_upsample2x_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
int gdrn_upsample2x_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, void* stream);
/* (ABI 4) BatchNorm + ReLU + 2x bilinear upsampling in one launch: y = upsample2x(relu(scale * x_raw + shift)), every source value rounded to the
 * storage format first (= gdrn_bn_apply followed by gdrn_upsample2x_fwd, bit for bit; cdpn_rot_head_region.py:103-123). */
int gdrn_bn_relu_upsample2x_fwd(const void* x_raw, const float* scale, const float* shift, void* y, int N, int H, int W, int C, int dtype, void* stream);
/* (ABI 4) gdrn_upsample2x_bwd + the reduction pass of the BatchNorm(+ReLU) backward that reads its result (gdrn_bn_bwd_reduce with the affine mask
 * mask_scale * x_raw + mask_shift > 0) in one launch: dx as gdrn_upsample2x_bwd writes it (unmasked), rows [gdrn_bn_bwd_reduce_rows(N*H*W, C,
 * dtype)][2][C] for gdrn_bn_bwd_coef. */
int gdrn_upsample2x_bwd_bnsums(const void* dy, void* dx, const void* x_raw, const float* mean, const float* invstd, const float* mask_scale,
                               const float* mask_shift, int N, int H, int W, int C, float* rows, int dtype, void* stream);

/* nn.GroupNorm(G, C) + ReLU (conv_pnp_net.py:78-80) and backward.  dgamma/dbeta: fp32 [C], accumulated
 * over samples (zeroed inside). */
int gdrn_gn_relu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int N,
                     int HW, int C, int G, float eps, int dtype, void* stream);
int gdrn_gn_relu_bwd(const void* dy, const void* y, const void* x, const float* gamma, const float* mean_rstd,
                     void* dx, float* dgamma, float* dbeta, int N, int HW, int C, int G, int dtype, void* stream);

/* elementwise LeakyReLU(0.1) backward on [n] elements given the activation output y: dx = dy*(y>0?1:0.1) */
int gdrn_leaky_bwd(const void* dy, const void* y, void* dx, long long n, int dtype, void* stream);
/* bias gradient: db[c] = sum_rows dy[r][c]  (dy of dtype, row stride cs) */
int gdrn_bias_grad(const void* dy, int cs, int rows, int C, float* db, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Head tail: slicing of the 1x1 output conv into mask / xyz / region, channel softmax over
 * region[:,1:], concat with roi_coord_2d, extent de-normalisation -> Patch-PnP input
 * (GDRN.py:156-169, conv_pnp_net.py:121-125).  head: fp32 [N*HW][hs] = (mask, x, y, z, region[nreg+1]);
 * pnp_in: dtype [N*HW][pcs], channels (xyz(3), coord2d(2), softmax(nreg), zero pad). */
int gdrn_head_tail_fwd(const float* head, int hs, const float* coord2d, const float* extents, void* pnp_in, int pcs,
                       int N, int HW, int nreg, int dtype, void* stream);
/* Train mode: gdrn_head_tail_fwd and gdrn_map_loss_fwd in ONE pass over the logits (acc as below, zeroed inside).
 * dtype | GDRN_PREZEROED (here and in gdrn_head_tail_fwd / gdrn_head_tail_bwd): the channels >= 72 of the pcs / dcs wide rows are
 * already zero and stay the caller's (the engine zero-fills its buffers once instead of re-writing 56 pad channels per pixel and step). */
int gdrn_head_tail_loss_fwd(const float* head, int hs, const float* coord2d, const float* extents, void* pnp_in, int pcs,
                            const float* gt_xyz, const float* mask_visib, const float* mask_trunc, const long long* gt_region,
                            double* acc, int N, int HW, int nreg, int dtype, void* stream);
/* dtype | GDRN_ACC_ROWS (gdrn_head_tail_loss_fwd): acc is fp64 [8 + 8 * gdrn_head_tail_loss_rows(...)]; every workgroup STORES its partial
 * sums as row acc[8 + 8 r .. 8 + 8 r + 5] (no memset, no atomics: 24 k atomic adds onto one cache line cost the kernel 15 us, and their
 * order made the sums run-to-run different in the last bits); gdrn_map_loss_finalize_rows adds the rows in a fixed order into acc[0..7]
 * (what gdrn_head_tail_bwd reads) and writes the five map losses. */
#define GDRN_ACC_ROWS 0x200
int gdrn_head_tail_loss_rows(int N, int HW, int nreg, int hs, int pcs);
int gdrn_map_loss_finalize_rows(double* acc, int nrows, int N, int HW, float* losses, void* stream);
/* Map losses (GDRN.py:345-400): acc[0..2] = sum|x*m-gt*m| per coordinate, acc[3] = sum|mask-trunc|,
 * acc[4] = CE_sum(region*m, gt_region*m), acc[5] = sum m  (acc: fp64 [8], zeroed inside). */
int gdrn_map_loss_fwd(const float* head, int hs, const float* gt_xyz, const float* mask_visib, const float* mask_trunc,
                      const long long* gt_region, int N, int HW, int nreg, double* acc, void* stream);
/* d_head[m][0..nreg+4] (dtype, stride dcs, rest zero) = grad of sum_k gw[k]*loss_k through the map
 * losses plus the chain through the head tail from d_pnp_in (NULL: none).
 * gw: device fp32 [5] = dL/d(loss_coor_x, loss_coor_y, loss_coor_z, loss_mask, loss_region). */
int gdrn_head_tail_bwd(const float* head, int hs, const void* pnp_in, const void* d_pnp_in, int pcs,
                       const float* extents, const float* gt_xyz, const float* mask_visib, const float* mask_trunc,
                       const long long* gt_region, const double* acc, const float* gw, void* d_head, int dcs, int N,
                       int HW, int nreg, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pose decode + pose losses for one batch, one thread block per RoI:
 *   rot6d -> R_allo (rot_reps.py:34-49); (dcx,dcy,z_rel) -> t (pose_from_pred_centroid_z.py:176-212);
 *   allocentric -> egocentric (train: core/utils/utils.py:208-236 with eps; test: utils.py:39-94, no eps);
 *   point-matching loss (pm_loss.py:82-114, misc.py:930-949, symmetric: pose_utils.py:430-482),
 *   centroid / z L1 (GDRN.py:444-471).
 * fc: fp32 [N][fs] rows = (rot6d[6], t_[3]).  sym: fp32 [N][Kmax][9] + sym_count[N] (NULL: none).
 * outputs: rot [N][9], trans [N][3]; losses[3] = (loss_PM_R, loss_centroid, loss_z) (zeroed inside);
 * dfc: fp32 [3][N][fs] unit gradients d loss_k / d fc (NULL in test mode); vis: fp32 [N][2] per-RoI
 * rotation error (deg) / translation error. */
typedef struct gdrn_pose_params {
    const float* fc;
    int fs;
    const float* cams;
    const float* centers;
    const float* whs;
    const float* ratios;
    const float* extents;
    const float* gt_rot;
    const float* gt_trans;
    const float* gt_trans_ratio;
    const float* points;
    int npts;
    const float* sym;
    const int* sym_count;
    int Kmax;
    int N;
    int train;
    float* rot;
    float* trans;
    float* losses;
    float* dfc;
    float* vis;
    float* loss_rows;
    const float* gw;
    void* dfc_comb;
    const float* fc2_ws;
    const float* fc2_bias;
    void* f2_out;
    const void* w_rt;
    const float* b_rt;
    float* fc_w;
    int fc2_splits;
} gdrn_pose_params;
int gdrn_pose_loss(const gdrn_pose_params* p, void* stream);
/* (ABI 5, all nullable = the version-4 behaviour) loss_rows [N][4]: the three pose losses as one row per RoI instead of atomics onto `losses`
 * (no memset; gdrn_loss_finalize adds them in RoI order).  gw [3] + dfc_comb [N][fs] (16-bit): dL/dfc = sum_k gw[k] * unit gradient k, written
 * directly (train; `dfc` is then not written).  fc2_ws .. fc2_splits: the tail of Patch-PnP's fully connected stack in the same launch
 * (conv_pnp_net.py:152-160) -- fc2_ws = the gdrn_linear_splits(1024, 256) slabs [split][N][256] gdrn_linear_splitk(act = GDRN_LINEAR_NO_FINISH)
 * left, fc2_bias fp32 [256], LeakyReLU(0.1), f2_out [N][256] 16-bit (fc2's activation), w_rt 16-bit [>= 9][256] row-major (fc_r | fc_t),
 * b_rt fp32 [9], fc_w [N][fs] fp32 receives the nine outputs (`fc` is then not read). */
int gdrn_loss_finalize(double* acc, int nrows, int N, int HW, const float* pose_rows, float* losses, const float* w, float* weighted,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small fp32 helpers */
/* out[r][c] = sum_k w[k] * in[k][r][c]  (combine the unit gradients with the incoming loss grads) */
int gdrn_combine3(const float* in, const float* w, float* out, int n, void* stream);
/* map-loss finalisation: losses[0..4] from acc (GDRN.py:347-400) */
int gdrn_map_loss_finalize(const double* acc, int N, int HW, float* losses, void* stream);

/* Fused Ranger step (RAdam + gradient centralisation + Lookahead) over one parameter tensor
 * (lib/torch_utils/solver/ranger.py:100-200).  The tensor is viewed as [rows][cols]; gc != 0 subtracts
 * the per-row gradient mean first (ranger.py:144-145).  step_size / adaptive are the RAdam
 * rectification terms of ranger.py:154-186 (host scalars); lookahead != 0 applies
 * slow += alpha*(p - slow); p = slow (ranger.py:192-198). */
int gdrn_ranger_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* slow, int rows, int cols,
                     int gc, float lr, float beta1, float beta2, float eps, float weight_decay, float step_size,
                     int adaptive, int lookahead, float alpha, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-tensor variants: ONE launch for all parameter tensors.  Task tables are arrays in DEVICE memory built once by
 * the host; `*_start` are int prefix arrays [ntasks + 1] (workgroups for pack/unpack: ceil(n / gdrn_pack_chunk()) per
 * task; rows for Ranger).
 * gdrn_pack_task: gdrn_pack4 semantics (dst[a1][a2][t][b] = src[a1*s1 + a2*s2 + t'*st + b*sb], zero padded); frag = 1 / 2:
 *   dst is the fragment-major permutation (gdrn_pack_wfrag / gdrn_pack_wfrag32) of that [A1][1][9][B] operand.  For gdrn_unpack_multi the
 *   same struct describes gdrn_unpack4 (src = packed fp32, dst = parameter-layout gradient, n = A1v*A2v*T*Bv). */
/* Workgroups per task (blk_start prefix sums): ceil(n / gdrn_pack_chunk()) for row-major destinations; bf16 fragment-major
 * (frag = 1, T = 9) tasks take (A1/16) * (B/64) workgroups -- one per brick of 16 rows x 64 b x 9 taps. */
/* gdrn_pack_task.scale (pack only, nullable): per-row factor, dst[a1][..] = src[..] * scale[a1] -- eval mode folds the
 * BatchNorm scale gamma/sqrt(var+eps) of the following BN into the conv operand this way. */
/* gdrn_pack_task.pad_ (frag tasks): 1 + log2(B*sizeof(dtype)/128) when that chunk count is a power of two (the kernel
 * then shifts instead of dividing), else 0. */
/* (ABI 5) frag = 3: tiled transpose -- a ROW-MAJOR copy (gdrn_pack4 / gdrn_unpack4 semantics, flip = 0, no scale) whose unit-stride source index
 * is not b: pad_ = 1 / 2 / 3 names it (a1 / a2 / t); a workgroup moves a 64 x 64 tile of (that index, b) through LDS, reading the source along it
 * and writing the destination along b (fc1's operand copies and gradient: 8.4 M elements with b strides of 64 / 8192 floats).  Workgroups per
 * task: gdrn_pack_transpose_blocks (<= 0: the task does not qualify). */
typedef struct gdrn_pack_task {
    const float* src;
    void* dst;
    const float* scale;
    int A1, A2, T, B, A1v, A2v, Bv;
    int flip;
    long long s1, s2, st, sb;
    long long n;
    int frag;
    int pad_;
} gdrn_pack_task;
typedef struct gdrn_ranger_task {
    float* p;
    const float* g;
    float* m;
    float* v;
    float* slow;
    int rows, cols, gc;
    float lr;
} gdrn_ranger_task;
/* gdrn_zero_multi: zero ntasks device regions (p 16-byte aligned, n16 granules of 16 bytes) in one launch; blk_start prefix sums of
 * ceil(n16 / gdrn_zero_chunk()).  With it a backward pass clears all its atomically-accumulated gradients (packed weight gradients
 * of the generic kernel, GroupNorm / bias / stem gradients) once, and the entry points below are called with GDRN_PREZEROED. */
typedef struct gdrn_zero_task {
    void* p;
    long long n16;
} gdrn_zero_task;
int gdrn_zero_chunk(void);
int gdrn_zero_multi(const gdrn_zero_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream);
/* gdrn_nonfinite_flag: *flag |= 1 if any of the n floats at x (16-byte aligned) is inf or NaN; *flag is never cleared by the library.  The
 * found_inf test of torch.cuda.amp.GradScaler (core/gdrn_modeling/main_gdrn.py:53-56, engine.py:276-283) for the fp16 arithmetic mode: the
 * host skips the optimizer step and backs the loss scale off when it is raised (ABI 3). */
int gdrn_nonfinite_flag(const float* x, long long n, int* flag, void* stream);
/* Dynamic loss scale of the fp16 arithmetic mode kept ON THE DEVICE (ABI 4, round 6): what torch.cuda.amp.GradScaler holds on the host and reads
 * back once per step (core/gdrn_modeling/engine.py:276-283 `scaler.step(optimizer)` / `scaler.update()`).  `flag` is the target of
 * gdrn_nonfinite_flag (first member: a gdrn_loss_scale_state* is a valid `int* flag`).  A step then is: backward pass with
 * gdrn_scaled_loss_weights' dL/dloss -> gdrn_nonfinite_flag over the gradients -> gdrn_ranger_multi_dyn (a no-op when the flag is raised; the RAdam
 * step index = base_step + applied + 1 counts applied steps only; the gradients are divided by `scale` inside the kernel) -> gdrn_loss_scale_update
 * (overflow: scale x 0.5 (>= 1), skipped + 1, flag cleared; clean: applied + applied_step, good + 1, and after `growth` clean steps in a row
 * scale x 2 (<= 65536)).  No host read anywhere; a host reads the struct when it wants to log or checkpoint. */
typedef struct gdrn_loss_scale_state {
    int flag;            /* raised by gdrn_nonfinite_flag, cleared by gdrn_loss_scale_update */
    int applied;         /* optimizer steps applied since the host wrote base_step */
    int skipped;         /* overflowed steps so far */
    int good;            /* clean steps since the scale last changed */
    int growth;          /* growth interval (0: the scale never grows) */
    float scale;         /* the loss scale on dL/dloss */
    float inv_scale;     /* 1 / scale */
    int base_step;       /* the optimizer's step count when `applied` was last zeroed */
    int last_overflowed; /* 1 if the step gdrn_loss_scale_update closed last had overflowed */
    int pad_[7];
} gdrn_loss_scale_state;
int gdrn_loss_scale_update(gdrn_loss_scale_state* state, int applied_step, void* stream);
/* g[i] = flag ? 0 : g[i] * factor / scale  (gradients handed to an external optimizer: unscaled, or zeroed when the step overflowed) */
int gdrn_unscale_or_zero(float* g, long long n, float factor, const gdrn_loss_scale_state* state, void* stream);
/* out[k] = w[k] * (w2 ? w2[k] : 1) * scale, k < n <= 64: dL/dloss of the scaled backward pass */
int gdrn_scaled_loss_weights(const float* w, const float* w2, int n, const gdrn_loss_scale_state* state, float* out, void* stream);
/* OR-ed into the `dtype` argument of gdrn_gn_relu_bwd / gdrn_bias_grad / gdrn_stem_wgrad: the gradient outputs they accumulate
 * into with atomics were zeroed by the caller (gdrn_zero_multi) -- skip the internal hipMemsetAsync (one launch each). */
#define GDRN_PREZEROED 0x100
int gdrn_pack_chunk(void);
int gdrn_pack_transpose_blocks(const gdrn_pack_task* task);
int gdrn_pack_multi(const gdrn_pack_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, int dtype, void* stream);
int gdrn_unpack_multi(const gdrn_pack_task* tasks_dev, const int* blk_start_dev, int ntasks, int nblocks, void* stream);
/* grad_scale: every gradient element is multiplied by it first (1/world_size when g holds the all-reduced SUM of the ranks'
 * gradients: the averaging pass over the 140 MB buffer disappears; 1.0 otherwise). */
int gdrn_ranger_multi(const gdrn_ranger_task* tasks_dev, const int* row_start_dev, int ntasks, int total_rows, float beta1,
                      float beta2, float eps, float weight_decay, float step_size, int adaptive, int lookahead, float alpha,
                      float grad_scale, void* stream);
/* the same update under a device-resident loss-scale state (see gdrn_loss_scale_state): step size / rectification / lookahead phase are
 * evaluated inside the kernel from state->base_step + state->applied + 1, the gradients are additionally divided by state->scale, and a raised
 * state->flag turns the launch into a no-op */
int gdrn_ranger_multi_dyn(const gdrn_ranger_task* tasks_dev, const int* row_start_dev, int ntasks, int total_rows, float beta1,
                          float beta2, float eps, float weight_decay, int n_sma_threshold, int lookahead_k, float alpha,
                          float grad_scale, const gdrn_loss_scale_state* state, void* stream);

/* (ABI 4) 3x3 STRIDE-2 pad-1 conv, forward, on a halo-tiled MFMA kernel (csrc/conv3x3s2.hip): ResNet-34's three stage-entry convs
 * (resnet_backbone.py:69-80, torchvision BasicBlock stride 2) and Patch-PnP's stride-2 convs (conv_pnp_net.py:76-92) -- optionally with the block's
 * 1x1 stride-2 shortcut conv (`downsample.0`) evaluated in the same launch from the same staged input (wd / yd / stats_d / bias_d).
 *   x [N][Hi][Wi][x_cs], y [N][Ho][Wo][y_cs], yd [N][Ho][Wo][yd_cs]: NHWC 16-bit; Hi = 2 Ho, Wi = 2 Wo, Ho % 4 == 0, Wo % 16 == 0 (or Wo % 8 == 0 and N even), Cin % 64 == 0,
 *   Cout % 128 == 0.  w: the FRAGMENT-MAJOR operand gdrn_pack_wfrag makes of the row-major [w_rows][9][Cin] weights; wd: ROW-MAJOR [wd_rows][Cin].
 *   stats / stats_d (nullable): [gdrn_conv3x3s2_stats_rows][2][Cout] partial sums for gdrn_bn_finalize; bias / bias_d (nullable) fp32 [Cout];
 *   act: 0 none, 1 ReLU (main conv only; the shortcut branch has none).
 *   (ABI 5) bnb_* (nullable, not combined with wd / stats / bias / act): the launch is a data gradient -- the ConvTranspose2d backward of the head is
 *   this conv applied to the output gradient -- w.r.t. a BatchNorm(+ReLU)'s output: y is masked where the stored activation bnb_mask <= 0 and rows
 *   [gdrn_conv3x3s2_stats_rows][2][Cout] of (sum g, sum g * (bnb_x - mean) * invstd) go to bnb_rows for gdrn_bn_bwd_coef (bnb_x / bnb_mask:
 *   [N][Ho][Wo][bnb_cs] 16-bit). */
typedef struct gdrn_s2_params {
    const void* x;
    const void* w;
    void* y;
    const float* bias;
    float* stats;
    const void* wd;
    void* yd;
    const float* bias_d;
    float* stats_d;
    int Hi, Wi, Cin, x_cs;
    int Ho, Wo, Cout, y_cs, yd_cs;
    int N, w_rows, wd_rows, act, dtype;
    const void* bnb_x;
    const void* bnb_mask;
    const float* bnb_mean;
    const float* bnb_invstd;
    float* bnb_rows;
    int bnb_cs;
} gdrn_s2_params;
int gdrn_conv3x3s2_ok(const gdrn_s2_params* p);
int gdrn_conv3x3s2_stats_rows(const gdrn_s2_params* p);
int gdrn_conv3x3s2(const gdrn_s2_params* p, void* stream);

/* (ABI 4) ... and its DATA GRADIENT (csrc/conv3x3s2_dgrad.hip; autograd backward of those convs, engine.py:279): dx [N][Hi][Wi][dx_cs] (Cin
 * channels) from dy [N][Ho][Wo][dy_cs] (Cout channels); w = the fragment-major operand gdrn_pack_wfrag makes of the row-major data-gradient
 * operand [w_rows >= Cin][9 taps, NOT flipped][Cout].  Optional: dyd / wdd -- the output gradient of the block's 1x1 stride-2 shortcut and its
 * ROW-MAJOR operand [wdd_rows >= Cin][Cout], whose data gradient is added in the same launch (even / even input pixels); bnb_* -- dx is the gradient
 * w.r.t. a BatchNorm(+ReLU)'s output: masked where the stored activation bnb_mask <= 0, and rows [gdrn_conv3x3s2_dgrad_rows][2][Cin] of
 * (sum g, sum g * (bnb_x - mean) * invstd) for gdrn_bn_bwd_coef.  Ho % 4 == 0, Wo % 16 == 0 (or Wo % 8 == 0 and N even), Cin % 64 == 0, Cout % 64 == 0.
 * (ABI 5) stats / bias / act -- the forward-conv epilogue, for the launch that IS the forward pass of nn.ConvTranspose2d(3, stride 2, pad 1,
 * output_padding 1) (dy = its input, dx = its output, w rows = its output channels): stats [gdrn_conv3x3s2_dgrad_rows][2][Cin] partial BatchNorm
 * sums of dx, bias fp32 [Cin], act 0 / 1 = ReLU.  All three NULL / 0: a data gradient.  Not combined with dyd / bnb_*. */
typedef struct gdrn_s2d_params {
    const void* dy;
    const void* w;
    void* dx;
    const void* dyd;
    const void* wdd;
    const void* bnb_x;
    const void* bnb_mask;
    const float* bnb_mean;
    const float* bnb_invstd;
    float* bnb_rows;
    int bnb_cs;
    int Hi, Wi, Cin, dx_cs;
    int Ho, Wo, Cout, dy_cs, dyd_cs;
    int N, w_rows, wdd_rows, dtype;
    float* stats;
    const float* bias;
    int act;
} gdrn_s2d_params;
int gdrn_conv3x3s2_dgrad_ok(const gdrn_s2d_params* p);
int gdrn_conv3x3s2_dgrad_rows(const gdrn_s2d_params* p);
int gdrn_conv3x3s2_dgrad(const gdrn_s2d_params* p, void* stream);

/* (ABI 4) One 64-channel ResNet BasicBlock in EVAL mode as one launch: y = relu(conv2(relu(conv1(x) + b1)) + b2 + x), both convs 3x3 stride 1
 * pad 1 with the BatchNorms folded into weights / biases (resnet_backbone.py:69-80 under model.eval(): ResNet-34's layer1 at inference).  x, y:
 * [N][H][W][64] NHWC 16-bit, y != x; w1, w2: the fragment-major operands gdrn_pack_wfrag makes of the row-major [64][9][64] weights; b1, b2: fp32
 * [64].  Bit-identical to the two gdrn_conv3x3_halo launches it replaces; the 64-channel intermediate stays in LDS.  gdrn_block64_eval_ok: 1 if
 * the shape is covered (H % 8 == 0, W % 16 == 0). */
int gdrn_block64_eval_ok(int N, int H, int W, int dtype);
int gdrn_block64_eval(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int N, int H, int W, int dtype,
                      void* stream);

/* ------------------------------------------------------------------------------------------------
 * Inference post-processing on the device: get_out_coor + get_out_mask
 * (core/gdrn_modeling/engine_utils.py:92-126, L1 branches) and GDRN_Evaluator.get_img_model_points_with_coords2d
 * (gdrn_evaluator.py:89-126, max_num_points < 4) for the whole batch, one workgroup per RoI.
 *   mask / coor_{x,y,z}: fp32 maps, element (n, pixel) at base[n*roi_stride + pixel*pix_stride]
 *     (NCHW [N,1,H,W] tensors: roi_stride = HW, pix_stride = 1; the engine's NHWC head output: hs*HW, hs)
 *   coord2d [N][2][HW], extents [N][3], im_hw [N][2] = (im_H, im_W), mask_thr = cfg.MODEL.CDPN.ROT_HEAD.MASK_THR_TEST
 *   out_mask [N][HW] = (mask - min)/(max - min) per RoI;  out_xyz [N][3][HW] (either may be NULL)
 *   img_pts [N][HW][2], model_pts [N][HW][3]: the selected pixels of RoI n in row-major order, counts[n] of them
 *     (both NULL: only the maps / counts).  fp32, the reference's operation order: bit-identical results. */
int gdrn_correspondences(const float* mask, const float* coor_x, const float* coor_y, const float* coor_z, long long roi_stride,
                         int pix_stride, const float* coord2d, const float* extents, const float* im_hw, float mask_thr, int N,
                         int HW, float* out_mask, float* out_xyz, float* img_pts, float* model_pts, int* counts, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GPU RoI cropper / target builder: what the reference's data loader does per instance on
 * the host with cv2 + numpy, for a whole batch of RoIs cut from frames that are already resident in HBM.
 *   replaces  crop_resize_by_warp_affine / get_affine_transform     core/utils/data_utils.py:80-137
 *             roi_img + normalize_image, roi_coord_2d                core/gdrn_modeling/data_loader.py:425-439, 487-498;
 *                                                                     core/base_data_loader.py:114-118
 *             roi_mask_{trunc,visib,obj}, roi_xyz, xyz_to_region,    data_loader.py:460-545, 617-632;
 *             roi_wh / resize_ratio / trans_ratio                     core/utils/data_utils.py:213-219
 * cv2.getAffineTransform / cv2.warpAffine (INTER_LINEAR on u8 and fp32, INTER_NEAREST, BORDER_CONSTANT 0) are followed
 * operation by operation (double LU + inversion, 10-bit fixed-point walk, 1/32-pixel bilinear table), rot = 0.
 * One gdrn_roi_task per RoI, as a device array.  Train-only members may be NULL / 0 when only gdrn_roi_crop_inputs is
 * used (test mode). */
/*   image     [H][W][3] u8 frame (the reference reads BGR) this RoI is cut from
 *   coord2d   [H][W][2] fp32 get_2d_coord_np(W, H, fmt="HWC") of that frame size
 *   xyz_crop  [y2-y1+1][x2-x1+1][3] fp32 object coordinates (xyz_info["xyz_crop"]); x1..y2 = xyz_info["xyxy"], inclusive (train)
 *   seg       [H][W] u8 0/1 visible-instance mask (anno["segmentation"]) (train)
 *   trunc     [H][W] u8 0/1 mask from background replacement, or NULL: mask_trunc = mask_visib (train)
 *   cx, cy, scale   bbox_center and the (clamped) square crop size in source pixels
 *   bw, bh          max(x2 - x1, 1), max(y2 - y1, 1) of the annotated box
 *   ox, oy, tz      anno["centroid_2d"] and trans[2] (train);  cls = row of fps_points / extents */
typedef struct gdrn_roi_task {
    const unsigned char* image;
    const float* coord2d;
    const float* xyz_crop;
    const unsigned char* seg;
    const unsigned char* trunc;
    double cx, cy, scale, bw, bh, ox, oy, tz;
    int H, W, x1, y1, x2, y2, cls, pad_;
} gdrn_roi_task;
/* minv [B][2][6] doubles: warpAffine's inverted 2x3 matrix (destination pixel -> source position) for the in_res and the
 * out_res crop of every RoI;  roi_wh [B][2], resize_ratio [B], trans_ratio [B][3] fp32 (each may be NULL). */
int gdrn_roi_affine(const gdrn_roi_task* tasks_dev, int B, int in_res, int out_res, double* minv, float* roi_wh, float* resize_ratio,
                    float* trans_ratio, void* stream);
/* roi_img [B][3][in_res][in_res] = (bilinear u8 crop - pixel_mean) / pixel_std (host arrays of 3 doubles, evaluated in
 * double like numpy);  roi_coord_2d [B][2][out_res][out_res] bilinear fp32 crop.  Either output may be NULL. */
int gdrn_roi_crop_inputs(const gdrn_roi_task* tasks_dev, const double* minv, int B, int in_res, int out_res, const double* pixel_mean,
                         const double* pixel_std, float* roi_img, float* roi_coord_2d, void* stream);
/* Train-mode targets at out_res (nearest crops): roi_xyz [B][3][r][r] = xyz / extent + 0.5, roi_mask_* [B][r][r] fp32,
 * roi_region [B][r][r] int32 = 1 + argmin_k |xyz - fps_points[cls][k]| on object pixels, 0 elsewhere (NULL together with
 * fps_points when NUM_REGIONS <= 1).  fps_points [ncls][nfps][3] double, extents [ncls][3] fp32, device arrays. */
int gdrn_roi_targets(const gdrn_roi_task* tasks_dev, const double* minv, int B, int out_res, const double* fps_points, int nfps,
                     const float* extents, float* roi_xyz, float* roi_mask_trunc, float* roi_mask_visib, float* roi_mask_obj,
                     int* roi_region, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDRN_HIP_H */
