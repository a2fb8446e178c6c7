/*
 * pixelnerf_hip.h -- C ABI of libpixelnerf_hip.so: the MI355X (gfx950) native pixelNeRF
 * volume-rendering hot path.
 *
 * The reference (sxyu/pixel-nerf) has no FFI of its own: its seam is two Python classes,
 * NeRFRenderer (src/render/nerf.py:45-371) and PixelNeRFNet (src/model/models.py:14-316).
 * Every entry point below replaces a span of those classes; the span is cited per function
 * (paths relative to the reference repository root).  The Python host layer
 * (pixel-nerf_amd/render, pixel-nerf_amd/model) binds these through ctypes and presents the
 * reference's API on top; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, current device) unless marked "host";
 *   - tensors are dense, row-major, fp32 unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued on it, nothing synchronises;
 *   - return value: 0 = ok, negative = PNR_E_* ; pnr_last_error() gives a message (host,
 *     thread-local);
 *   - memory is owned by the caller (torch tensors in the Python host); the library
 *     allocates nothing.
 */
#ifndef PIXELNERF_HIP_H
#define PIXELNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNR_OK 0
#define PNR_E_INVALID (-1)   /* bad argument / unsupported shape          */
#define PNR_E_HIP (-2)       /* a HIP runtime call or kernel launch failed */

/* arithmetic of the 512-wide linear layers (operands of the MFMA; accumulation is fp32) */
#define PNR_PREC_F16 0       /* fp16 operands, v_mfma_f32_32x32x16_f16  */
#define PNR_PREC_BF16 1      /* bf16 operands, v_mfma_f32_32x32x16_bf16 */
#define PNR_PREC_F32 2       /* exact fp32 validation path (pnr_eval_*_f32 only), v_mfma_f32_32x32x2_f32 */
#define PNR_PREC_F16X3 3     /* fp32-class: every operand a (head, tail) pair of fp16, 3 f16 MFMAs per product,
                                fp32 tables (pnr_*_split entries; pnr_render_forward_folded accepts it) */

/* Encoded-scene state: exactly what PixelNeRFNet.encode() leaves in module buffers
 * (src/model/models.py:111-141, src/model/encoder.py:160-163), except that the feature grid
 * is channel-last so one bilinear corner is a contiguous 2 KiB row. */
typedef struct PnrScene {
    const float *latent_nhwc; /* (SB*NS, Hl, Wl, 512)  encoder.latent permuted NCHW->NHWC   */
    const float *poses;       /* (SB*NS, 3, 4) world->camera [R^T | -R^T t]  models.py:112-114 */
    const float *focal;       /* (n_focal, 2)  (fx, -fy)                     models.py:129-130 */
    const float *c;           /* (n_c, 2)      principal point               models.py:132-141 */
    int32_t SB;               /* objects                                                    */
    int32_t NS;               /* source views per object (row = obj*NS + view)              */
    int32_t Hl, Wl;           /* latent grid size                                           */
    int32_t n_focal, n_c;     /* 1 (broadcast) or SB (per object)            models.py:207-212 */
    float img_w, img_h;       /* net.image_shape = (W, H)                    models.py:116-117 */
} PnrScene;

/* One ResnetFC (src/model/resnetfc.py:66-130) at the only shape the reference ships:
 * d_in=42, d_latent=512, d_hidden=512, n_blocks=5, combine_layer=3, d_out=4, ReLU, average
 * (or max) pooling.  nn.Linear layout: weight (out, in) row-major, bias (out). */
typedef struct PnrMlpWeights {
    const float *lin_in_w, *lin_in_b;       /* (512,42), (512)  */
    const float *lin_z_w[3], *lin_z_b[3];   /* (512,512), (512) */
    const float *fc0_w[5], *fc0_b[5];       /* blocks[b].fc_0   */
    const float *fc1_w[5], *fc1_b[5];       /* blocks[b].fc_1   */
    const float *lin_out_w, *lin_out_b;     /* (4,512), (4)     */
    int32_t combine_max;                    /* pooling over the source views (util.combine_interleaved, util.py:461-471):
                                               0 = "average" (every shipped config), 1 = "max" (inference entries only) */
} PnrMlpWeights;

const char *pnr_last_error(void);

/* Library / device facts (host out-params may be NULL). */
int pnr_version(int *major, int *minor);
/* ABI revision of THIS header: bumped whenever a struct layout or an entry point's argument list changes.  The
 * library returns the value it was compiled with; a binding must compare it with the header it was written against
 * before the first call (pixelnerf_amd/_lib.py does, and refuses a stale or foreign .so). */
#define PNR_ABI_VERSION 8
int pnr_abi_version(void);
int pnr_device_info(int *num_cus, int *lds_bytes_per_block);

/* 64-bit content fingerprint of the 30 parameter tensors, computed and (optionally) compared on the device with no host
 * synchronisation: ws = pnr_params_checksum_ws_bytes() of device scratch, first 8 bytes zero (left so); sum_out (device,
 * nullable) receives it; with `expect`
 * (device) a differing value sets *mismatch_flag (device int) to 1.  Lets a caller that caches packed streams notice
 * parameter writes that bypass its cache key (the Python layer: `p.data.copy_()` bumps no tensor version). */
size_t pnr_params_checksum_ws_bytes(void);
int pnr_params_checksum(const PnrMlpWeights *w /*host*/, void *ws, unsigned long long *sum_out,
                        const unsigned long long *expect, int *mismatch_flag, void *stream);

/* ---- one-time weight repack ------------------------------------------------------------
 * Replaces nothing in the reference (it feeds nn.Linear weights to addmm directly,
 * resnetfc.py:147,175,55-62,183); needed because the fused kernel streams MFMA fragments.
 * Re-run whenever the parameters change.  `packed` must hold pnr_packed_mlp_bytes() bytes. */
size_t pnr_packed_mlp_bytes(void);
int pnr_pack_mlp(const PnrMlpWeights *w /*host struct of device ptrs*/, int precision,
                 void *packed, void *stream);

/* ---- folded inference form -------------------------------------------------------------------
 * lin_z[b] (resnetfc.py:175-180) acts on the bilinearly interpolated latent, and both are linear:
 * lin_z[b](sum_c w_c grid[c]) = sum_c w_c (W_z[b] grid[c] + b_z[b]) because the bilinear weights sum to 1.
 * pnr_fold_latent applies W_z[b] to every texel of the encoded grid once per scene and network (fp32 MFMA,
 * 3 tables of (SB*NS,Hl,Wl,512) 16-bit, pnr_folded_tables_bytes), pnr_pack_mlp_folded packs the stream
 * without the three lin_z GEMMs, and the *_folded entry points replace those GEMMs (22-28 % of the
 * per-point FLOPs) by one bilinear lookup per table.  Same results to the stated tolerance; inference
 * only.  Re-fold when the grid or the lin_z parameters change. */
size_t pnr_folded_tables_bytes(const PnrScene *scene /*host*/);
int pnr_fold_latent(const PnrScene *scene /*host*/, const PnrMlpWeights *w /*host*/, int precision,
                    void *tables, void *stream);
int pnr_pack_mlp_folded(const PnrMlpWeights *w /*host*/, int precision, void *packed /*pnr_packed_mlp_bytes()*/,
                        void *stream);
int pnr_eval_ray_samples_folded(const PnrScene *scene /*host*/, const void *packed_folded, const void *tables,
                                int precision, const float *rays, const float *z, int R, int rays_per_obj,
                                int K, float *rgbsigma, void *stream);
int pnr_eval_points_folded(const PnrScene *scene /*host*/, const void *packed_folded, const void *tables,
                           int precision, const float *xyz, const float *viewdirs, int B, float *rgbsigma,
                           void *stream);
int pnr_render_forward_folded(const PnrScene *scene /*host*/, const void *packed_coarse,
                              const void *tables_coarse, const void *packed_fine /*nullable*/,
                              const void *tables_fine, int precision, const float *rays, int R,
                              int rays_per_obj, int Kc, int Kf, int Kfd, float depth_std, int white_bkgd,
                              int lindisp, const float *u1, const float *u2, const float *u3,
                              const float *n4, float *rgb_c, float *depth_c, float *weights_c,
                              float *rgb_f, float *depth_f, float *weights_f, void *workspace,
                              void *stream);

/* ---- fp32-class accuracy on the f16 matrix cores (PNR_PREC_F16X3): same spans as the folded entries above
 * (src/model/models.py:146-266, src/model/resnetfc.py:132-184), any number of source views, inference.
 * w = wh + wl and x = xh + xl with f16 heads/tails, w x ~= wh xh + wh xl + wl xh accumulated in fp32 (error
 * 2^-22 per product instead of 2^-11); lin_z folded into fp32 per-texel tables.  Agrees with the reference's fp32
 * arithmetic to the bars of the exact-fp32 path (per-point |rgb| <= 2e-5) at several times the fp32-MFMA ceiling.
 * packed_split: pnr_packed_mlp_split_bytes() bytes from pnr_pack_mlp_split; tables_f32: pnr_folded_tables_f32_bytes()
 * bytes from pnr_fold_latent_f32 (layout of the 16-bit tables, 4 bytes per entry). */
size_t pnr_packed_mlp_split_bytes(void);
int pnr_pack_mlp_split(const PnrMlpWeights *w /*host*/, void *packed_split, void *stream);
size_t pnr_folded_tables_f32_bytes(const PnrScene *scene /*host*/);
int pnr_fold_latent_f32(const PnrScene *scene /*host*/, const PnrMlpWeights *w /*host*/, float *tables_f32, void *stream);
/* The same tables for the texels ONE training pass reads (ABI rev 8).  A training step re-folds lin_z every pass -- the
 * weights moved -- and on a large grid most texels are not near any ray of the pass (DTU, 128 rays x 3 views: 37-43 k of
 * 90 k).  rays (R,8), z (R,K) = the pass's samples: every (view, point) is projected with the forward kernels' own code, its
 * four corner rows are marked, the marked rows are folded (same sums, same order: the bits of pnr_fold_latent_f32) and written
 * at their texels' places; all other rows of tables_f32 keep what they held -- hand in a buffer that was zero-initialised once
 * and use it only for a pnr_eval_ray_samples_split_train call on the SAME rays and z.  workspace:
 * pnr_fold_latent_f32_rows_workspace_bytes() bytes of device memory, 16-byte aligned, caller-owned (graph capture).
 * Takes grids of >= 8192 texels in total (smaller ones: pnr_fold_latent_f32 is one short launch). */
size_t pnr_fold_latent_f32_rows_workspace_bytes(const PnrScene *scene /*host*/);
int pnr_fold_latent_f32_rows(const PnrScene *scene /*host*/, const PnrMlpWeights *w /*host*/, const float *rays, const float *z,
                             int R, int rays_per_obj, int K, float *tables_f32, void *workspace, size_t workspace_bytes,
                             void *stream);
int pnr_eval_ray_samples_split(const PnrScene *scene /*host*/, const void *packed_split, const void *tables_f32,
                               const float *rays, const float *z, int R, int rays_per_obj, int K, float *rgbsigma,
                               void *stream);
int pnr_eval_points_split(const PnrScene *scene /*host*/, const void *packed_split, const void *tables_f32,
                          const float *xyz, const float *viewdirs, int B, float *rgbsigma, void *stream);

/* encoder.latent NCHW -> NHWC (layout change for the lookup in src/model/encoder.py:80-109). */
int pnr_nchw_to_nhwc(const float *in, float *out, int N, int C, int H, int W, void *stream);

/* ---- ray sampling ------------------------------------------------------------------------
 * NeRFRenderer.sample_coarse, src/render/nerf.py:98-118.  rays (R,8), u1 (R,Kc) uniforms
 * (the reference's torch.rand_like at :111) -> z (R,Kc). */
int pnr_sample_coarse(const float *rays, const float *u1, int R, int Kc, int lindisp, float *z,
                      void *stream);

/* NeRFRenderer.sample_fine + sample_fine_depth + cat + sort, src/render/nerf.py:120-161 and
 * :285-295.  weights_c (R,Kc), depth_c (R), z_coarse (R,Kc); u2,u3 (R,Kimp) uniforms (:135,
 * :141), n4 (R,Kfd) normals (:158); Kimp = n_fine - n_fine_depth.  Either count may be 0
 * (the pointers are then ignored).  z_sorted (R, Kc+Kimp+Kfd) ascending.  depth_ranks (may be
 * NULL): (R,Kfd) position of every depth sample in z_sorted (training: the sort's permutation). */
int pnr_sample_fine(const float *rays, const float *weights_c, const float *depth_c,
                    const float *z_coarse, const float *u2, const float *u3, const float *n4,
                    int R, int Kc, int Kimp, int Kfd, float depth_std, int lindisp,
                    float *z_sorted, int32_t *depth_ranks, void *stream);

/* ---- the fused per-point network ---------------------------------------------------------
 * PixelNeRFNet.forward, src/model/models.py:146-266, including PositionalEncoding
 * (src/model/code.py:30-42), SpatialEncoder.index (src/model/encoder.py:80-109),
 * ResnetFC.forward (src/model/resnetfc.py:132-184), view mean pooling
 * (src/util/util.py:461-471) and the output activations (models.py:260-265).
 *
 * Variant A (renderer path; also fuses nerf.py:185,204 "points = o + z d, viewdirs = d"):
 *   rays (R,8), z (R,K) -> rgbsigma (R,K,4).  rays_per_obj = R / SB. */
int pnr_eval_ray_samples(const PnrScene *scene /*host*/, const void *packed, int precision,
                         const float *rays, const float *z, int R, int rays_per_obj, int K,
                         float *rgbsigma, void *stream);
/* Variant B (direct net(xyz, viewdirs) calls): xyz, viewdirs (SB,B,3) -> rgbsigma (SB,B,4). */
int pnr_eval_points(const PnrScene *scene /*host*/, const void *packed, int precision,
                    const float *xyz, const float *viewdirs, int B, float *rgbsigma,
                    void *stream);

/* ResnetFC.forward on explicit rows (src/model/resnetfc.py:132-184): zx (rows, 512 + 42) fp32 = [latent | code+viewdir]
 * per row, rows ordered [group][view][point] with combine_inner_dims = (NS, B) (mean over the NS views before block 3,
 * util.py:461-471; NS = 1: no pooling, B ignored beyond divisibility).  out (rows / NS, 4) is lin_out's RAW output
 * (no sigmoid / relu: those are PixelNeRFNet.forward's, models.py:260-265).  Unfused fp32 linears, same kernels as the
 * exact-fp32 path below; workspace = pnr_resnetfc_forward_f32_workspace_bytes(rows, NS). */
size_t pnr_resnetfc_forward_f32_workspace_bytes(long long rows, int NS);
int pnr_resnetfc_forward_f32(const PnrMlpWeights *w /*host*/, const float *zx, long long rows, int NS, int B, float *out,
                             void *workspace, size_t workspace_bytes, void *stream);

/* ---- nn.Linear as a stand-alone operator pair ----------------------------------------------------
 * One `nn.Linear` of ResnetFC / ResnetBlockFC (src/model/resnetfc.py:53-62: fc_0, fc_1, shortcut; :147,175-183: lin_in, lin_z,
 * scale_z, lin_out) with the ReLU in front of it and the residual behind it folded in, for ANY rows / d_in / d_out: the
 * operator the host side composes ResnetFCs of NON-shipped shapes from (the shipped 42+512 -> 512 x 5 -> 4 shape runs in the
 * fused kernels above).
 *   Y (rows, d_out) = [Yin +] [relu](X (rows, d_in)) W^T + b      W (d_out, d_in) as nn.Linear stores it; b, Yin nullable;
 *                                                                 Yin may alias Y
 * precision PNR_PREC_F32: exact fp32 products (v_mfma_f32_32x32x2_f32); PNR_PREC_F16X3: fp32-class split operands on the f16
 * matrix cores (~6x faster, the arithmetic of the fused fp32-class kernels). */
int pnr_linear(const float *X, const float *W, const float *b /*nullable*/, const float *Yin /*nullable*/, float *Y, long long rows,
               int d_in, int d_out, int relu_in, int precision, void *stream);
/* its backward (what torch autograd derives for the lines above):
 *   dX (rows, d_in)  = (dY W) . [X > 0 if relu_in]      (nullable)
 *   dW (d_out, d_in) = dY^T [relu](X),  db (d_out) = sum_rows dY      (nullable; db only together with dW)
 * The reduction over the rows is split into fixed slices summed in a fixed order (bit-reproducible, no atomics):
 * workspace = pnr_linear_backward_workspace_bytes(d_in, d_out), needed when dW is requested.
 * PNR_PREC_F16X3 carries dY as fp16 (head, tail) pairs at a power-of-two scale: grad_scale = device [s, 1/s] from
 * pnr_grad_scale(dY) (required for that precision, ignored for PNR_PREC_F32). */
size_t pnr_linear_backward_workspace_bytes(int d_in, int d_out);
int pnr_linear_backward(const float *dY, const float *X, const float *W, long long rows, int d_in, int d_out, int relu_in,
                        float *dX /*nullable*/, float *dW /*nullable*/, float *db /*nullable*/, const float *grad_scale /*device [s,1/s]*/,
                        void *workspace, size_t workspace_bytes, int precision, void *stream);

/* ---- exact-fp32 evaluation (validation grade) --------------------------------------------------
 * Same contract as pnr_eval_ray_samples / pnr_eval_points (PixelNeRFNet.forward, models.py:161-265)
 * with every operand in fp32: unfused, one GEMM launch per nn.Linear on the fp32 MFMA, activations
 * in HBM.  Takes the raw nn.Linear tensors (no repack).  Results track the reference's fp32 path
 * to rounding level; throughput is ~1/20 of the 16-bit fused kernel.  The points are processed in
 * chunks sized by the workspace: pnr_eval_f32_workspace_bytes(NS, chunk_points) bytes hold one
 * chunk of chunk_points points (>= 64). */
size_t pnr_eval_f32_workspace_bytes(int NS, long long chunk_points);
int pnr_eval_ray_samples_f32(const PnrScene *scene /*host*/, const PnrMlpWeights *w /*host*/,
                             const float *rays, const float *z, int R, int rays_per_obj, int K,
                             float *rgbsigma, void *workspace, size_t workspace_bytes, void *stream);
int pnr_eval_points_f32(const PnrScene *scene /*host*/, const PnrMlpWeights *w /*host*/,
                        const float *xyz, const float *viewdirs, int B, float *rgbsigma,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- training (autograd) support -------------------------------------------------------------
 * The reference trains through this path with plain autograd (train/train.py:199-215): MSE on
 * coarse + fine rgb, gradients w.r.t. both ResnetFCs and encoder.latent.  Here:
 *   forward : pnr_eval_ray_samples_train = pnr_eval_ray_samples + 16-bit dumps of every linear
 *             layer's input operand (PnrTrainDumps);
 *   backward: pnr_composite_backward (d rgb/depth/weights -> d rgbsigma per point),
 *             pnr_mlp_backward (fused data-gradient chain with transposed weight streams: consumes
 *             d(pre-activation output), writes the per-layer output gradients dY as 16-bit rows),
 *             pnr_latent_scatter (d interpolated latent -> d feature grid, bilinear scatter-add);
 *             pnr_weight_grad_batched / pnr_lin_out_grad (dW = dY^T X, db = sum dY from those dumps),
 *             pnr_depth_sample_backward (the position gradient of the depth samples, nerf.py:292).
 * Array shapes: rows_v = NS*P for per-view layers (row = view*P + point), rows_p = P pooled;
 * 512-wide dims of activation dumps / gradients are in "storage order" (pnr_storage_perm). */
typedef struct PnrTrainDumps {
    void *d_in;     /* (rows_v, 64)  lin_in operand: code(39) | viewdir(3) | 0-pad, natural order */
    void *d_z;      /* (rows_v, 512) interpolated latent, natural channel order                  */
    void *d_a[5];   /* relu(x) in front of blocks[b].fc_0: b<3 (rows_v,512), b>=3 (rows_p,512)   */
    void *d_n[5];   /* relu(net) in front of blocks[b].fc_1, same shapes                         */
    void *d_x5;     /* (rows_p, 512) relu(x) in front of lin_out                                 */
    void *d_mask;   /* pnr_train_masks_bytes(P, NS) bytes: 1 bit per element of the 11 activation dumps ("was dumped non-zero"
                       = the relu derivative), 64-bit word per (layer, view, 64-point tile, kernel thread); what the
                       backward chain reads instead of the 1 KiB dump rows (which only the weight-gradient GEMMs read) */
} PnrTrainDumps;
size_t pnr_train_masks_bytes(long long P, int NS);

typedef struct PnrBackwardDumps {
    void *g_fc1[5]; /* dL/d(blocks[b].fc_1 output) = dL/d(residual stream after block b), shapes as d_n */
    void *g_fc0[5]; /* dL/d(blocks[b].fc_0 output), shapes as d_a                                */
    void *g_x0;     /* (rows_v, 512) dL/d(residual stream in front of block 0) = dY of lin_in, lin_z[0] */
    float *d_zlat;  /* (rows_v, 512) fp32, natural channel order: d(interpolated latent) = sum_b dY_b W_z[b]
                       (resnetfc.py:175-180 backward), unscaled; REQUIRED (pnr_mlp_backward rejects NULL)     */
    float *d_in;    /* (rows_v, 42) fp32: d(positional code | view direction) = dY W_in (resnetfc.py:147
                       backward), unscaled; NULL = not wanted                                            */
} PnrBackwardDumps;

int pnr_eval_ray_samples_train(const PnrScene *scene /*host*/, const void *packed, int precision,
                               const float *rays, const float *z, int R, int rays_per_obj, int K,
                               float *rgbsigma, const PnrTrainDumps *dumps /*host*/, void *stream);

/* feature index held at storage position e (0..511) of an activation dump row: perm[e]. host out. */
int pnr_storage_perm(int32_t *perm512 /*host*/);

/* transposed weight streams for the backward chain (fc_1^T, fc_0^T of every block, lin_out^T). */
size_t pnr_packed_mlp_bwd_bytes(void);
int pnr_pack_mlp_bwd(const PnrMlpWeights *w /*host struct of device ptrs*/, int precision,
                     void *packed_bwd, void *stream);

/* Backward of pnr_composite (src/render/nerf.py:178-182,223-249 under autograd).
 * d_depth / d_weights may be NULL.  d_rgbsigma (R,K,4) = dL/d(model output) after sigmoid/relu,
 * or, with pre_activation != 0, dL/d(lin_out output) (through rgb = sigmoid(.), sigma = relu(.),
 * src/model/models.py:260-263) -- the g_out that pnr_mlp_backward consumes;
 * d_z (R,K) (may be NULL) = dL/dz through the deltas and depth = sum w z. */
int pnr_composite_backward(const float *rays, const float *z, const float *rgbsigma, int R, int K,
                           int white_bkgd, const float *d_rgb, const float *d_depth,
                           const float *d_weights, float *d_rgbsigma, float *d_z,
                           int pre_activation, void *stream);

/* dL/dz of the sample positions through the network inputs (x = o + z d: positional code,
 * models.py:169-182 / code.py:37-41, and projection + bilinear lookup, models.py:206-215 /
 * encoder.py:96-109 with grid_sample's border-clip gradient).  d_in42 (rows_v,42) and d_zlat
 * (rows_v,512) fp32 are dL/d(lin_in operand) and dL/d(interpolated latent); d_z (R,K) is
 * accumulated (atomic adds; caller zero-initialises or passes the compositing d_z). */
int pnr_position_backward(const PnrScene *scene /*host*/, const float *rays, const float *z, int R,
                          int rays_per_obj, int K, const float *d_in42, const float *d_zlat,
                          float *d_z, void *stream);

/* The same gradient for the DEPTH samples only -- the only sample positions that carry gradient in the reference (the
 * importance samples are drawn from detached weights, nerf.py:288; z = clamp(depth_c + n4 * depth_std), nerf.py:157-160,292)
 * -- pushed through the sort and the clamp towards dL/d(coarse depth): for ray r, depth sample j at sorted position
 * ranks[r][j] and view v, contrib[v][r][j] = [near < depth_c[r] + n4[r][j] depth_std < far] * (network term of view v
 * [+ dz_comp[r][ranks[r][j]] for v = 0]); dL/d depth_c[r] = sum over (v, j) -- left to the caller so that the sum has a
 * fixed order.  dz_comp (R,K) = the compositing dL/dz of pnr_composite_backward (nullable); contrib (NS,R,Kfd) out. */
int pnr_depth_sample_backward(const PnrScene *scene /*host*/, const float *rays, const float *z, int R, int rays_per_obj,
                              int K, const int *ranks, const float *n4, int Kfd, const float *depth_c, float depth_std,
                              const float *d_in42, const float *d_zlat, const float *dz_comp, float *contrib, void *stream);

/* Fused data-gradient chain of one ResnetFC (reverse of src/model/resnetfc.py:132-184).
 * g_out (P,4) = dL/d(lin_out output) (pre sigmoid/relu), grad_scale = power of two the chain is
 * run at (16-bit range management; every dump is scaled by it).  NS = views per object. */
int pnr_mlp_backward(const void *packed_bwd, int precision, const PnrTrainDumps *fwd /*host*/,
                     const float *g_out, float grad_scale, const float *grad_scale_dev /*nullable:
                     device scalar that overrides grad_scale (no host sync to pick it)*/,
                     long long P, int NS, const PnrBackwardDumps *out /*host*/, void *stream);

/* Picks that scale on the device: scales[0] = 2^(6 - ceil(log2 max|g|)) (1 for g == 0), scales[1] =
 * 1/scales[0]; both NaN when g holds a non-finite value (the gradients then come out NaN).  g: n floats. */
int pnr_grad_scale(const float *g, long long n, float *scales /*device, 2 floats*/, void *stream);

/* Weight gradient of one 512x512 linear from the 16-bit dumps: dW (512,512) fp32 = out_scale *
 * dY^T X, db (512) fp32 = out_scale * column sums of dY (db may be NULL); dY, X (rows,512) 16-bit
 * row-major at `precision`.  Split over row slices with a fixed-order reduction (bit-reproducible);
 * workspace: pnr_weight_grad_workspace_bytes().  rows_storage_order / cols_storage_order: dY / X
 * columns are in storage order (pnr_storage_perm); dW and db are always written in feature order. */
size_t pnr_weight_grad_workspace_bytes(void);
int pnr_weight_grad(const void *dY, const void *X, long long rows, int precision, float out_scale,
                    int rows_storage_order, int cols_storage_order, float *dW, float *db,
                    void *workspace, void *stream);

/* The same for up to 16 linears in ONE launch pair (all 13 512x512 layers of a ResnetFC): fewer, fuller
 * launches and 4-8x fewer split slices to reduce.  jobs: host array.
 * workspace: pnr_weight_grad_batched_workspace_bytes(n_jobs, largest rows of the jobs).
 * precision PNR_PREC_F16X3: dY and X are (head | tail) f16 row sets (the tail array behind the head array), three MFMAs per
 * product -- the fused fp32-class training path. */
typedef struct PnrWeightGradJob {
    const void *dY, *X;          /* (rows,512) 16-bit row-major dumps                  */
    long long rows;
    int rows_storage_order, cols_storage_order;
    float *dW, *db;              /* (512,dw_cols), (512) fp32 outputs; db may be NULL */
    int x_cols;                  /* columns (= row stride) of X: 0 -> 512; 64 for the lin_in operand (natural order) */
    int dw_cols;                 /* columns (= row stride) of dW written: 0 -> x_cols; 42 for lin_in                 */
} PnrWeightGradJob;
size_t pnr_weight_grad_batched_workspace_bytes(int n_jobs, long long max_rows);
int pnr_weight_grad_batched(const PnrWeightGradJob *jobs /*host*/, int n_jobs, int precision,
                            float out_scale, const float *out_scale_dev /*nullable device scalar,
                            multiplies out_scale*/, void *workspace, void *stream);

/* lin_out (4 x 512): dW = g_out^T x5, db = column sums of g_out; g_out (P,4) fp32 = dL/d(lin_out output),
 * x5 (P,512) 16-bit dump (storage order); dW written in feature order.  Fixed-order reduction. */
size_t pnr_lin_out_grad_workspace_bytes(void);
int pnr_lin_out_grad(const float *g_out, const void *x5, long long P, int precision, float *dW,
                     float *db, void *workspace, void *stream);

/* d(encoder.latent) += bilinear scatter of d_zlat (rows_v,512) fp32 (natural channel order) to
 * d_latent_nhwc (SB*NS,Hl,Wl,512) fp32 (caller zero-initialises; accumulates).  Grids of up to 2274
 * texels per image (32 x 32 and the like) are accumulated in fp64 LDS slabs, one per (image, 16- / 8-channel slice),
 * fed per ray segment (consecutive samples in one grid cell); a slab reaches HBM with plain read-add-write when one
 * workgroup owns its (image, slice), and with one atomic per touched element otherwise -- never more than two workgroups
 * per pair (the slice width follows the image count), so onto a ZEROED buffer the result is bit-reproducible.  Larger grids
 * (64 x 64, DTU's 150 x 200) are cut into tiles of 32 x 32 texels: one owner workgroup per (image, tile, 16-channel slice)
 * takes the segments whose corners touch its tile from a binned list and writes the tile back with plain read-add-write
 * (every texel has one owner; fp64 sums, no HBM atomics).  Objects of 2^29 samples or more, grids of more than 8192 tiles
 * and PIXELNERF_SCATTER_TILED=0 take global fp32 atomics (order-dependent in the last bit).
 * workspace: pnr_latent_scatter_workspace_bytes() bytes of device memory, 16-byte aligned (projected positions, segment
 * lists, tile lists; 0 only on the global-atomic path, workspace may then be NULL), owned by the caller so that the call can
 * sit inside a HIP-graph capture (ABI rev 7; rev 6 kept a per-stream scratch inside the library).  (encoder.py:96-109 backward) */
size_t pnr_latent_scatter_workspace_bytes(const PnrScene *scene /*host*/, int R, int rays_per_obj, int K);
/* 1 when this call shape takes the tiled form (one owner workgroup per grid element, plain read-add-write): several calls may
 * then accumulate into ONE buffer order-independently (a training step's fine and coarse pass); 0: give every call its own
 * zeroed buffer and sum them for a bit-reproducible gradient (ABI rev 8) */
int pnr_latent_scatter_single_owner(const PnrScene *scene /*host*/, int R, int rays_per_obj, int K);
int pnr_latent_scatter(const PnrScene *scene /*host*/, const float *rays, const float *z, int R,
                       int rays_per_obj, int K, const float *d_zlat, float *d_latent_nhwc, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ---- alpha compositing -------------------------------------------------------------------
 * NeRFRenderer.composite after the model call, src/render/nerf.py:178-182 and :223-249.
 * rays (R,8), z (R,K), rgbsigma (R,K,4) -> weights (R,K) (may be NULL), rgb (R,3), depth (R). */
int pnr_composite(const float *rays, const float *z, const float *rgbsigma, int R, int K,
                  int white_bkgd, float *weights, float *rgb, float *depth, void *stream);

/* ---- whole renderer forward --------------------------------------------------------------
 * NeRFRenderer.forward, src/render/nerf.py:251-303 (inference; no autograd).
 * Noise pointers follow the reference's draw order (u1 :111, u2 :135, u3 :141, n4 :158).
 * packed_fine == NULL falls back to the coarse network (models.py:242); the fine pass then evaluates only the Kf
 * new samples and merges them with the coarse pass's outputs at the Kc shared positions (same network, same
 * points: bit-identical to evaluating all Kc+Kf).
 * Outputs: *_c (coarse), *_f (fine; ignored when Kf == 0); weights pointers may be NULL.
 * workspace: pnr_render_workspace_bytes(R,Kc,Kf) bytes of scratch. */
size_t pnr_render_workspace_bytes(int R, int Kc, int Kf);
int pnr_render_forward(const PnrScene *scene /*host*/, const void *packed_coarse,
                       const void *packed_fine, int precision, const float *rays, int R,
                       int rays_per_obj, int Kc, int Kf, int Kfd, float depth_std,
                       int white_bkgd, int lindisp, const float *u1, const float *u2,
                       const float *u3, const float *n4, float *rgb_c, float *depth_c,
                       float *weights_c, float *rgb_f, float *depth_f, float *weights_f,
                       void *workspace, void *stream);

/* ---- next-row helpers -----------------------------------------------
 * util.gen_rays / unproj_map, src/util/util.py:113-143,238-276 (ndc=False).
 * poses (NV,4,4) camera-to-world -> rays (NV,H,W,8). */
int pnr_gen_rays(const float *poses, int NV, int W, int H, float fx, float fy, float cx, float cy,
                 float z_near, float z_far, float *rays, void *stream);

/* ---- counter-based random draws (production mode; SURVEY.md 7 "hard parts", nerf.py:111,135,141,158) ------------
 * The reference draws its jitter / importance / depth-sample noise with four torch.rand launches per call.  The
 * seeded entries below draw inside the sampling kernels instead: Philox4x32-10, key = 64-bit seed, counter =
 * (global ray id, value index / 4, draw id), uniforms = 24 random bits in [0,1), normals by Box-Muller.  A value
 * depends only on (seed, global ray id, draw, index): an image does not change with chunking or with sharding
 * the rays across GPUs (ray_id_offset / ray_id_stride place a shard inside the whole ray set: id =
 * (r / rays_per_obj) * ray_id_stride + r % rays_per_obj + ray_id_offset; 0 / 0 = the call IS the whole set).
 * pnr_philox_noise writes the same draws out as the explicit tensors of pnr_render_forward (u1 (R,Kc), u2, u3
 * (R,Kimp), n4 (R,Kfd)); pnr_philox_raw is the bare block function on the host, for known-answer tests. */
int pnr_philox_noise(unsigned long long seed, long long ray_id_offset, int ray_id_stride, int rays_per_obj, int R,
                     int Kc, int Kimp, int Kfd, float *u1, float *u2, float *u3, float *n4, void *stream);
int pnr_philox_raw(const uint32_t *counter4 /*host*/, const uint32_t *key2 /*host*/, uint32_t *out4 /*host*/);
/* NeRFRenderer.forward (nerf.py:251-303) with in-kernel draws; tables_* NULL = unfolded streams. */
int pnr_render_forward_seeded(const PnrScene *scene /*host*/, const void *packed_coarse, const void *tables_coarse,
                              const void *packed_fine /*nullable*/, const void *tables_fine, int precision,
                              const float *rays, int R, int rays_per_obj, int Kc, int Kf, int Kfd, float depth_std,
                              int white_bkgd, int lindisp, unsigned long long seed, long long ray_id_offset,
                              int ray_id_stride, float *rgb_c, float *depth_c, float *weights_c, float *rgb_f,
                              float *depth_f, float *weights_f, void *workspace, void *stream);

/* ---- next-row helpers (continued): render whole target views ----------
 * util.gen_rays + NeRFRenderer.forward in ONE call (what eval/eval.py:247-279 and eval/gen_video.py do per
 * object: build all rays of the target views on the host, copy, render): poses_c2w (NV,4,4) device, views
 * grouped per object (NV = SB * views_per_object).  The rays are NOT materialised: the sampling, network and
 * compositing kernels regenerate the ray of a pixel from (pose, intrinsics, pixel id) where they need it, in
 * gen_rays' operation order (bit-identical to pnr_gen_rays + pnr_render_forward).  tables_* NULL = unfolded
 * streams.  Noise: explicit tensors laid out as in pnr_render_forward with R = NV*H*W, or all four NULL ->
 * in-kernel draws from `seed` (ray id = pixel index in the (NV,H,W) order). */
size_t pnr_render_views_workspace_bytes(int NV, int W, int H, int Kc, int Kf);
int pnr_render_views(const PnrScene *scene /*host*/, const void *packed_coarse, const void *tables_coarse,
                     const void *packed_fine /*nullable*/, const void *tables_fine, int precision,
                     const float *poses_c2w, int NV, int W, int H, float fx, float fy, float cx, float cy,
                     float z_near, float z_far, int Kc, int Kf, int Kfd, float depth_std, int white_bkgd,
                     int lindisp, const float *u1, const float *u2, const float *u3, const float *n4,
                     unsigned long long seed, float *rgb_c, float *depth_c, float *weights_c, float *rgb_f,
                     float *depth_f, float *weights_f, void *workspace, void *stream);

/* ---- fp16-range guard of the fp32-class ("f16x3") kernels ---------------------------------------
 * The split-operand kernels carry every operand as an fp16 (head, tail) pair: values up to 65504 are
 * represented to ~2^-22, beyond that the head SATURATES (MODE.FP16_OVFL) and the result silently leaves
 * the reference's arithmetic class (src/model/resnetfc.py:132-184 computes in fp32, whatever the
 * magnitude).  Random-init networks stay four orders of magnitude below the limit; a checkpoint need not.
 * pnr_saturation_guard(flags) arms the guard for the CALLING HOST THREAD: until it is called again with
 * NULL, every launch of a split-operand network kernel on that thread runs the instantiation that
 * follows the largest operand head produced (one v_pk_maximum3_f16 per four values in the split epilogue,
 * 1.15 % of the kernel; a head of 65504 means a value >= 65488) and ORs into flags[0] (coarse-network launches and the direct pnr_eval_*_split
 * entries) / flags[1] (fine-network launches of pnr_render_*):
 *   bit 2b    a value >= 65504 in relu(x) entering blocks[b].fc_0        (b = 0..4)
 *   bit 2b+1  a value >= 65504 in relu(net) entering blocks[b].fc_1
 *   bit 10    a value >= 65504 in the stream in front of lin_out
 *   bit 11    a non-finite network output
 *   bit 12    (pnr_fold_latent_f32, called while armed) a feature-grid value or lin_z weight beyond the fp16 range
 * flags: DEVICE array of two 32-bit words, zeroed by the caller, read back by the caller (asynchronously:
 * pixelnerf_amd copies it to pinned memory and looks at it on the next call, like the parameter check).
 * The guard never changes a result. */
int pnr_saturation_guard(unsigned int *flags_dev);

/* ---- SpatialEncoder.index as a stand-alone operator ------------------------------------------
 * src/model/encoder.py:80-109 (SpatialEncoder.index): F.grid_sample(latent, uv[:, :, None], mode "bilinear",
 * padding_mode "border", align_corners=True)[..., 0] on the encoded grid -- latent_nhwc (NV,Hl,Wl,C) is the
 * channel-last copy (pnr_nchw_to_nhwc / pnr_pyramid_to_latent), uv (NV,N,2) the NORMALISED coordinates in
 * [-1,1] (x, y) the reference forms at encoder.py:96-99 (`uv * latent_scaling / image_size - 1`),
 * out (NV,C,N) as the reference returns it.  ATen's corner arithmetic in its operation order; a NaN
 * coordinate reads texel 0 (what the fused kernels do, pinned by the adv_plane golden).  The fused
 * network kernels do not call this: they gather the same rows themselves.
 * _backward: d_latent_nhwc (NV,Hl,Wl,C) is ACCUMULATED into (zero it first; fp32 atomics), d_uv (NV,N,2) is
 * written; either may be NULL.  Gradient through the border clip as ATen's clip_coordinates_set_grad
 * (zero outside the open interval (0, size-1)). */
int pnr_grid_index(const float *latent_nhwc, int NV, int Hl, int Wl, int C, const float *uv, long long N,
                   float *out, void *stream);
int pnr_grid_index_backward(const float *latent_nhwc, int NV, int Hl, int Wl, int C, const float *uv,
                            long long N, const float *g_out, float *d_latent_nhwc, float *d_uv, void *stream);

/* ---- PositionalEncoding as a stand-alone operator --------------------------------------------
 * src/model/code.py:30-42 (PositionalEncoding.forward): x (N, d_in) ->
 * out (N, d_out), d_out = d_in * (2 num_freqs + (include_input ? 1 : 0)):
 *   out[n] = [x[n]] ++ sin(phases2[j] + x[n][d] * freqs2[j]),  j = 0 .. 2 num_freqs - 1 outer, d inner,
 * freqs2 / phases2 = DEVICE arrays of 2 num_freqs floats: the module's `_freqs` / `_phases` buffers
 * (code.py:24-28: every frequency twice, phases 0 and pi/2), so a loaded checkpoint's buffers are
 * what is evaluated.  The sine argument is one fused multiply-add (ATen's addcmul).  The fused
 * kernels do not call this: they form the code of their own 3-vectors in registers.
 * _backward: g_x (N, d_in) = d/dx of sum(out * g_out)  (torch autograd of the same lines). */
int pnr_positional_encoding(const float *x, long long N, int d_in, int num_freqs, const float *freqs2,
                            const float *phases2, int include_input, float *out, void *stream);
int pnr_positional_encoding_backward(const float *x, const float *g_out, long long N, int d_in,
                                     int num_freqs, const float *freqs2, const float *phases2,
                                     int include_input, float *g_x, void *stream);

/* ---- next-row helpers: encoder output formatting --------------------
 * src/model/encoder.py:150-163: F.interpolate(bilinear, align_corners=True) of every ResNet stage
 * to stage 0's size + channel concat.  stages: HOST array of n_stages device pointers, stage s is
 * (NV, channels[s], heights[s], widths[s]) NCHW fp32; channels multiples of 64.  Writes the grid
 * channel-last (NV, H0, W0, sum channels) -- the layout PnrScene.latent_nhwc wants -- and, if
 * latent_nchw != NULL, the reference's NCHW `latent` tensor as well, in one pass. */
int pnr_pyramid_to_latent(const float *const *stages, const int *channels, const int *heights,
                          const int *widths, int n_stages, int NV, float *latent_nhwc,
                          float *latent_nchw, void *stream);

/* ---- next-row helpers: eval epilogue on device ----------------------
 * eval/eval.py:283-290,327-329 and util.psnr (src/util/util.py:474-481): rgb (n_views,pixels,3) ->
 * clamp to [0,1] [-> uint8 = trunc(x*255)]; depth -> (d - z_near)/(z_far - z_near); per-view sum
 * of squared errors against gt_rgb (same shape, [0,1]) in fp64 (PSNR_v = -10 log10(sse_v /
 * (3*pixels))).  Every output pointer may be NULL. */
int pnr_eval_epilogue(const float *rgb, const float *depth, int n_views, int pixels_per_view,
                      float z_near, float z_far, const float *gt_rgb, unsigned char *rgb_u8,
                      float *rgb_clamped, float *depth_norm, double *sq_err_sum, void *stream);

/* ---- next-row helpers: training-ray selection on device ---------------
 * train/train.py:143-182 with util.bbox_sample (src/util/util.py:220-235): for each of SB objects
 * pick B pixels among its NV views and emit their rays and ground-truth colours (image*0.5+0.5)
 * directly -- the reference builds all NV*H*W rays per object to index 128 of them.
 * poses (SB,NV,4,4) camera-to-world; images (SB,NV,3,H,W) in [-1,1]; focal (SB,2) = (fx,fy);
 * c (SB,2) or NULL (image centre).  Random draws are inputs, in the reference's order:
 *   bboxes != NULL: bboxes (SB,NV,4) float [x0,y0,x1,y1]; ids (SB,B) int64 = randint(0,NV);
 *                   ux, uy (SB,B) = rand;   x = long(ux*(x1+1-x0)+x0), y likewise;
 *   bboxes == NULL: ids (SB,B) int64 = randint(0,NV*H*W) flat pixel indices; ux, uy ignored.
 * Out: rays (SB,B,8), rgb_gt (SB,B,3). */
int pnr_sample_training_rays(const float *poses, const float *images, const float *focal,
                             const float *c, const float *bboxes, const long long *ids,
                             const float *ux, const float *uy, int SB, int NV, int W, int H, int B,
                             float z_near, float z_far, float *rays, float *rgb_gt, void *stream);

/* ---- exact-fp32 TRAINING path (validation grade; precision "f32" under autograd).  The unfused fp32 chain of
 * pnr_eval_ray_samples_f32 with every activation the backward needs kept in fp32, and its backward on fp32 MFMA:
 * the fp32 yardstick of the 16-bit training kernels (ResnetFC.forward / autograd through it, src/model/resnetfc.py:132-184;
 * train/train.py:199-215).  rows_v = NS * P ordered [view][point], rows_p = P. */
typedef struct PnrF32Saved {
    float *in42;    /* (rows_v, 64)  lin_in operand: positional code | rotated view direction | 0 pad   */
    float *zlat;    /* (rows_v, 512) interpolated latent (operand of lin_z[0..2])                      */
    float *xin[5];  /* residual stream entering blocks[b] (after + lin_z[b]): b < 3 (rows_v,512), else (rows_p,512) */
    float *net[5];  /* blocks[b].fc_0 output (pre-relu), same shapes                                   */
    float *x5;      /* (rows_p, 512) residual stream in front of lin_out                               */
    float *pool_in; /* (rows_v, 512) residual stream after block 2, in front of the view mean; NS == 1: unused, may be NULL */
} PnrF32Saved;
/* split_gemm = 0: every product on the exact fp32 MFMA (precision "f32": the yardstick).  split_gemm = 1: the same chain with
 * every product formed from (head, tail) fp16 operand pairs -- 3 f16 MFMAs, fp32 accumulate, the arithmetic of
 * PNR_PREC_F16X3 -- about 6x faster at fp32-class accuracy (precision "f16x3" under autograd). */
int pnr_eval_ray_samples_f32_train(const PnrScene *scene /*host*/, const PnrMlpWeights *w /*host*/, const float *rays,
                                   const float *z, int R, int rays_per_obj, int K, float *rgbsigma,
                                   const PnrF32Saved *saved /*host*/, int split_gemm, void *stream);
/* ---- fp32-class TRAINING, fused (the default of precision "f16x3" under autograd).  Forward: the inference kernel of
 * PNR_PREC_F16X3 (one launch per network pass, lin_z through the folded fp32 tables) in its training instantiation, which also
 * keeps what the backward needs -- every 512-wide linear's operand as the (head, tail) f16 images the kernel multiplies from,
 * copied out of LDS, and 1-bit relu masks.  Backward: one launch for all 15 transposed products of the network (lin_out^T,
 * ten fc^T, lin_z[2..0]^T, lin_in^T; gradient images copied out the same way), one batched split-operand weight-gradient
 * launch pair over those images (pnr_weight_grad_batched at PNR_PREC_F16X3) and lin_out's 4 x 512 gradient.  Same reference
 * lines and the same accuracy class as the GEMM-per-layer entries above (the tests hold both to the same gradient bars).
 * "(head | tail)": two f16 arrays of the stated shape, the tail array directly behind the head array.  Storage order:
 * pnr_storage_perm.  packed_split = pnr_pack_mlp_split stream, tables_f32 = pnr_fold_latent_f32 tables of the CURRENT
 * parameters and grid (re-pack / re-fold after every optimizer step / encode()). */
typedef struct PnrSplitSaved {
    void *in_op;   /* (rows_v, 64)  (head | tail), natural order: lin_in operand (code | view direction | 0 pad) */
    void *zlat;    /* (rows_v, 512) (head | tail), natural channel order: interpolated latent                     */
    void *a[5];    /* relu(x) entering blocks[b].fc_0, (head | tail) rows in storage order: b < 3 (rows_v,512), else (rows_p,512) */
    void *n[5];    /* relu(fc_0 output) entering blocks[b].fc_1, same shapes                                       */
    float *x5;     /* (rows_p, 512) fp32, natural order: residual stream in front of lin_out                       */
    void *masks;   /* pnr_train_masks_bytes(P, NS) bytes: 1-bit relu masks of the 11 activations                   */
} PnrSplitSaved;
int pnr_eval_ray_samples_split_train(const PnrScene *scene /*host*/, const void *packed_split, const void *tables_f32,
                                     const float *rays, const float *z, int R, int rays_per_obj, int K, float *rgbsigma,
                                     const PnrSplitSaved *saved /*host*/, void *stream);
/* grad_scale = device [s, 1/s] from pnr_grad_scale(g_out); outputs as pnr_mlp_backward_f32 (d_zlat required, d_in nullable). */
size_t pnr_mlp_backward_split_workspace_bytes(long long P, int NS);
int pnr_mlp_backward_split(const PnrMlpWeights *w /*host*/, const PnrSplitSaved *saved /*host*/, const float *g_out, long long P,
                           int NS, const PnrMlpWeights *grads /*host*/, float *d_zlat, float *d_in /*nullable*/,
                           const float *grad_scale, void *workspace, size_t workspace_bytes, void *stream);
/* All parameter gradients of one ResnetFC + d(interpolated latent) [+ d(lin_in operand)] from g_out (P,4) =
 * dL/d(lin_out output): `grads` holds device pointers of the 30 gradient tensors in PnrMlpWeights' layout (same shapes as
 * the parameters, overwritten); d_zlat (rows_v,512), d_in (rows_v,42) or NULL. */
size_t pnr_mlp_backward_f32_workspace_bytes(long long P, int NS);
/* split_gemm = 1 additionally takes grad_scale = device [s, 1/s] from pnr_grad_scale(g_out): the gradient chain runs at the
 * power-of-two scale s (fp16 range of the operand heads), results are un-scaled exactly on their way out. */
int pnr_mlp_backward_f32(const PnrMlpWeights *w /*host*/, const PnrF32Saved *saved /*host*/, const float *g_out, long long P,
                         int NS, const PnrMlpWeights *grads /*host*/, float *d_zlat, float *d_in, int split_gemm,
                         const float *grad_scale, void *workspace, size_t workspace_bytes, void *stream);

/* The feature phase of the exact-fp32 path on its own -- PositionalEncoding.forward (src/model/code.py:30-42) on the
 * rotated point + the rotated view direction (src/model/models.py:161-196) and SpatialEncoder.index
 * (src/model/encoder.py:80-109) -- for B points per object: in42 (NS*SB*B, 64) = [code(39) | R d (3) | zero pad],
 * zlat (NS*SB*B, 512), rows ordered [view][object][point].  What lin_in / lin_z consume in pnr_eval_points_f32. */
int pnr_point_features_f32(const PnrScene *scene /*host*/, const float *xyz, const float *viewdirs, int B,
                           float *in42, float *zlat, void *stream);

/* Timing hook for bench.py: seconds spent in the fused network kernel launches issued on
 * `stream` since the last reset, measured with HIP events recorded around each launch on
 * that stream (call only after the stream has been synchronised). */
int pnr_profile_enable(int on);
int pnr_profile_read(double *mlp_kernel_ms, int *mlp_launches);

#ifdef __cplusplus
}
#endif
#endif /* PIXELNERF_HIP_H */
