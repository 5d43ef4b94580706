/*
 * include/dcd_hip.h -- C ABI of libdcd_hip.so, the MI355X (gfx950) implementation of the DGDE hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers, sizes and a HIP stream
 * (passed as void* so the header needs no HIP include), enqueues work on that stream without
 * any host synchronisation, and returns an int status:
 *     0 = DCD_OK, 1 = DCD_ERR_BAD_ARG (shape / null pointer / unsupported value),
 *     2 = DCD_ERR_WORKSPACE (workspace too small), 3 = DCD_ERR_LAUNCH (hipGetLastError != success).
 * Nothing is printed; the caller raises.  All tensors are contiguous fp32 NCHW unless stated.
 *
 * Each function cites the reference interface it replaces (paths relative to /root/reference).
 */
#ifndef DCD_HIP_H
#define DCD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCD_OK 0
#define DCD_ERR_BAD_ARG 1
#define DCD_ERR_WORKSPACE 2
#define DCD_ERR_LAUNCH 3

/* Library / build identification: returns e.g. "dcd_hip 0.1 gfx950". */
const char *dcd_version(void);

/* ------------------------------------------------------------------------------------------------
 * Environment.  The library's RESULTS never depend on the process environment (beyond summation order); which of two equivalent
 * launch sequences a call takes can: the variables below are A/B-timing and test switches read through dcd_env()
 * (csrc/tuning_env.h) -- some once per process (static), some per call (marked *).  Building with -DDCD_NO_TUNING_ENV compiles
 * every read out (each switch then keeps the default given here).  tests/test_abi.py checks this list against the sources.
 *   DCD_DCN_HANDOVER     auto (default) | always | never: who takes the far samples of a one-pass DCN backward -- per layer from
 *                        its previous call's far count (auto), the device-side hand-over to the generic kernels armed on every
 *                        call (always), or never armed.  See dcd_dcn_v2_forget below.
 *   DCD_FAR_DIV, DCD_FAR_DIV_WIDE   integer d > 0: hand a one-pass backward call to the generic kernels above 1 far coordinate in
 *                        d (Cout <= 64 / wider).  Default 24 / 16 at >= 4 images, 12 / 8 below.
 *   DCD_BWD_SWEEP *      0: three-pass DCN backward everywhere; 2: layers the dense path takes stay on it; default 1: one-pass
 *                        backward wherever it applies.
 *   DCD_SWEEP_WIDE *     0: one-pass backward for Cout <= 64 only (round 3's limit).  Default: Cout <= 128, and 256 where the
 *                        dense path does not apply.
 *   DCD_SWEEP_SLOTS *    waves of a one-pass backward launch (0: one work unit per wave).  Default: one round of resident waves.
 *   DCD_SWEEP_MIN_ROWS   fewest rows of a one-pass backward work unit (a unit pays ~5 rows of ring warm-up).  Default 3
 *                        (8: the segment choice of rounds 3-4 on under-filled launches).
 *   DCD_DCN_DENSE *      0: never the column-buffer path; 1: every geometry it can take; default: Cin >= 256 only.
 *   DCD_NO_TILE *        set: the workgroup-tiled LDS kernels (forward, generic backward) stand down for the register-gather ones.
 *   DCD_TILE_ROWS        4: forward regions of 4 rows always.  Default 8 rows when that fills the chip.
 *   DCD_FWD_RESCUE_TAPS  1..10: far taps in a region's worst wave that hand the region to the rescue kernel.  Default 5.
 *   DCD_BD_TILE128       1: the tiled generic data-gradient kernel also for Cout 128.  Default off.
 *   DCD_BI_TILE_NBLK     input-channel blocks (of 32) up to which the tiled generic grad_input kernel is used.  Default 2.
 *   DCD_NO_BI_TILE *     set: never the tiled generic grad_input kernel.
 *   DCD_BI_HYBRID        0: one generic grad_input kernel per call, chosen by the call-wide offset radius.  Default: per tile.
 *   DCD_BI_MB *          1 | 2 | 4 | 8: channel blocks per workgroup of the register-gather grad_input kernel.  Default: by size.
 *   DCD_DW_GEN           1: first-generation generic grad_weight kernel.  Default 2.
 *   DCD_DW_ORDER         0: the generic grad_weight kernel walks its tiles row by row.  Default: down column strips.
 *   DCD_CONV_GEOM *      0 | 1: pin the Winograd region shape (8 x 32 / 12 x 20 px), no split contraction.  Default: by cost model.
 *   DCD_CONV_MINCHUNK    smallest number of 8-channel chunks a split of the Winograd contraction keeps.  Default 4.
 *   DCD_CONV_DIRECT_MINCHUNK  the same for the direct bf16 form's 16-channel chunks.  Default 16.
 *   DCD_CONV_DIRECT_P *  4: 16 x 32 px regions (one workgroup per CU) instead of 8 x 32 in the direct bf16 form.  Default 2.
 *   DCD_CONV_WRW_DIRECT * 0: the one-product weight gradient stays on the Winograd-domain kernel.  Default 1 (direct, Cout > 32).
 *   DCD_CONV_DIRECT_MODE * operand residency of the direct bf16 form: 0 weights in registers, 1 window operands in registers, 2 both
 *                        streamed from LDS (four waves per SIMD).  Default 1.
 *   DCD_CONV_DIRECT_WX * 2: 8 x 64 px regions (eight waves) in the direct bf16 form.  Default 1 (8 x 32).
 *   DCD_BN_SMALL         0: no single-workgroup-per-channel BatchNorm kernels for small maps.  Default on.
 *   DCD_CHANNEL_SUM_ONE_LAUNCH   1 | 0: pin the per-channel sums to the one-launch / two-launch form.  Default: by grid size.
 *   DCD_UP_FWD_OLD, DCD_UP_BWD_OLD   set: the round-1 depthwise up-sampling kernels.
 * ---------------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------------
 * DCNv2 (modulated deformable convolution).
 * Replaces `_ext.dcn_v2_forward` / `_ext.dcn_v2_backward`
 *   DGDE/model/backbone/DCNv2/DCN/src/dcn_v2.h:9-46, :48-92            (dispatch)
 *   DGDE/model/backbone/DCNv2/DCN/src/cuda/dcn_v2_cuda.cu:42-172, :206-341 (host)
 *   DGDE/model/backbone/DCNv2/DCN/src/cuda/dcn_v2_im2col_cuda.cu:125-327   (kernels)
 * and the three extern "C" launchers in cuda/dcn_v2_im2col_cuda.h:68-99 (fused away: no column buffer).
 *
 * input  (B,Cin,H,W)            weight (Cout,Cin,kh,kw)       bias (Cout)
 * offset (B,dg*2*kh*kw,Ho,Wo)   interleaved (dh,dw) per tap   mask (B,dg*kh*kw,Ho,Wo)
 * output / grad_output (B,Cout,Ho,Wo), Ho = (H+2ph-(dh*(kh-1)+1))/sh+1, likewise Wo.
 *
 * precision: 0 = DCD_PREC_F32    exact fp32 MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32)
 *            1 = DCD_PREC_BF16X3 split-bf16 (hi*hi + hi*lo + lo*hi on the bf16 MFMAs, fp32 accumulate;
 *                                ~2^-16 relative per product).  Permits, does not oblige: geometries
 *                                without a split kernel run the exact fp32 kernels.
 *            2 = DCD_PREC_BF16   mixed precision (MODEL.FP16, DGDE/model/detector.py:34-36): both operands of the weight
 *                                contraction rounded to bf16 (nearest even), ONE product on the bf16 MFMAs, fp32
 *                                accumulate; sampling / coordinate arithmetic, storage and every sum stay fp32.
 *                                ~2^-9 relative per operand.  Permits, does not oblige (kernels without the one-product
 *                                form run the split-bf16 or the exact fp32 one).
 * workspace: device scratch of at least dcd_dcn_v2_workspace_bytes(...) bytes, 256-byte aligned,
 *            owned by the caller; contents are dead after the call's kernels complete.
 * ---------------------------------------------------------------------------------------------- */
#define DCD_PREC_F32 0
#define DCD_PREC_BF16X3 1
#define DCD_PREC_BF16 2

size_t dcd_dcn_v2_workspace_bytes(int B, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw,
                                  int ph, int pw, int dh, int dw, int dg);

int dcd_dcn_v2_forward(void *stream, const float *input, const float *weight, const float *bias,
                       const float *offset, const float *mask, float *output, int B, int Cin, int H, int W,
                       int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int dg,
                       int precision, void *workspace, size_t workspace_bytes);

/* All five gradients are fully overwritten (the callee zero-fills what it accumulates into).
 * grad_input and grad_weight are summed with fp32 atomics where work units overlap, like the reference's col2im
 * (cuda/dcn_v2_im2col_cuda.cu:249: order-nondeterministic sums); grad_offset / grad_mask are bit-reproducible on the
 * one-pass path (3x3, stride 1, pad 1, one group, Cout <= 64: csrc/dcn_bwd_sweep.inc) when no sample is displaced by
 * 3 px or more. */
int dcd_dcn_v2_backward(void *stream, const float *input, const float *weight, const float *bias,
                        const float *offset, const float *mask, const float *grad_output, float *grad_input,
                        float *grad_offset, float *grad_mask, float *grad_weight, float *grad_bias, int B,
                        int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                        int dw, int dg, int precision, void *workspace, size_t workspace_bytes);

/* Per-layer launch policy of the DCNv2 op (csrc/dcn_v2.hip, "Who takes the far samples").  The library remembers, per
 * (device, weight pointer), the far-sample count of the layer's last backward call (a word of mapped pinned host memory the
 * call's last kernel stores to) and chooses the next call's launch sequence from it -- never its result.
 *   dcd_dcn_v2_forget(weight)   the owner of `weight` (current device) tells the library that this address no longer names that
 *                               layer (tensor freed, moved, re-used): the entry is dropped, the next call on that address starts as
 *                               "unknown" (hand-over armed).  weight = NULL: every entry of the current device.  The dropped
 *                               entry's report word is reset and recycled, not unmapped (a HIP graph captured earlier may still
 *                               store to it).
 *   dcd_dcn_v2_policy_state(weight, &far)   0: no entry; 1: entry, no report yet; 2: *far = the last reported far count; -1: error.
 *   dcd_dcn_v2_policy_free()    frees all policy state of the process (all devices).  Only when no captured graph containing
 *                               DCN backward calls will be replayed again.
 *   dcd_dcn_v2_set_handover(mode)   pins the policy for the process: 0 never hand over, 1 always armed, 2 auto (the default), -1 back
 *                               to the DCD_DCN_HANDOVER environment default.  For callers that need two runs of one step to take the
 *                               same launch sequence (a captured graph and its eager twin).
 * A CAPTURED call freezes its decision: a graph captured while a layer's offsets were small replays the "never hand over" sequence
 * (correct for any offsets, slower beyond the far-count limit) until it is captured again. */
int dcd_dcn_v2_forget(const float *weight);
int dcd_dcn_v2_policy_state(const float *weight, unsigned *far_count);
int dcd_dcn_v2_policy_free(void);
int dcd_dcn_v2_set_handover(int mode);

/* ------------------------------------------------------------------------------------------------
 * Edge-constraint depth solver.
 * Replaces Anno_Encoder.decode_pairs_kpts_depth + get_up  (DGDE/model/anno_encoder.py:313-390),
 * PostProcessor.compute_pairs_kpts_depth                  (DGDE/model/head/detector_infer.py:215-225)
 * and GMW compute_z                                        (GMW/main.py:373-416).
 *
 * kps (N,K,2) image pixels; kps3d (N,K,3) object frame; rot_y (N); P (N,3,4) projection matrices;
 * kmask (N,K) uint8 or NULL.  npairs = K*(K-1)/2 pairs enumerated row-major over the upper triangle.
 *   z_ij = |(Y_i-Y_j) + (v'_i C_i - v'_j C_j)| / max(|v'_i - v'_j|, 1e-10), clamped to [zmin,zmax],
 *   v' = (v - P[1][2]) / P[1][1] (or v itself if normalized != 0), C = X sin(rot) - Z cos(rot).
 * topk == 0 : depth (N,npairs) in pair order                (eval; anno_encoder.py:383-384)
 * topk  > 0 : the topk pairs with the largest |v'_i-v'_j|, ordered like torch.topk (descending value,
 *             ties by lower pair index first); depth (N,topk), pair_idx (N,topk) int32,
 *             pair_mask (N,topk) float = kmask_i*kmask_j (only if kmask)      (anno_encoder.py:377-382)
 * Finally subtracts P[2][3] when sub_b3 != 0 (anno_encoder.py:385; not in GMW).
 * K <= 128.
 * ---------------------------------------------------------------------------------------------- */
int dcd_edge_depth_forward(void *stream, const float *kps, const float *kps3d, const float *rot_y,
                           const float *P, const uint8_t *kmask, int N, int K, int topk, float zmin,
                           float zmax, int normalized, int sub_b3, float *depth, int32_t *pair_idx,
                           float *pair_mask);

/* Backward of the above w.r.t. kps (v coordinate only; u does not enter) and kps3d.
 * grad_depth (N,M) with M = topk ? topk : npairs; pair_idx as produced by the forward (NULL if topk==0).
 * grad_kps (N,K,2) and grad_kps3d (N,K,3) are overwritten.  No gradient flows through the clamp's
 * saturated branch, the top-k selection, or max(.,1e-10) when saturated (autograd semantics). */
int dcd_edge_depth_backward(void *stream, const float *kps, const float *kps3d, const float *rot_y,
                            const float *P, const float *grad_depth, const int32_t *pair_idx, int N, int K,
                            int topk, float zmin, float zmax, int normalized, float *grad_kps,
                            float *grad_kps3d);

/* ------------------------------------------------------------------------------------------------
 * Penalty-reduced focal loss.  Replaces FocalLoss.forward (DGDE/model/layers/focal_loss.py:57-86).
 * pred, target: n elements.  out[0] = loss sum, out[1] = number of positives (target == 1).
 * grad_pred (n) may be NULL; if given it receives d(loss_sum)/d(pred) (to be scaled by the caller).
 * `out` must be zero-filled by the callee (it is) -- two floats.
 * ---------------------------------------------------------------------------------------------- */
int dcd_focal_loss(void *stream, const float *pred, const float *target, int64_t n, float alpha, float beta,
                   float *out, float *grad_pred);

/* ------------------------------------------------------------------------------------------------
 * GIoU loss on (l,t,r,b) distances.  Replaces IOULoss.forward (DGDE/model/layers/iou_loss.py:12-49).
 * pred, target (N,4) -> losses (N) = 1 - giou, ious (N); grad_pred (N,4) optional = d losses_i / d pred_i.
 * ---------------------------------------------------------------------------------------------- */
int dcd_giou_loss(void *stream, const float *pred, const float *target, int N, float *losses, float *ious,
                  float *grad_pred);

/* ------------------------------------------------------------------------------------------------
 * Heat-map decode: 3x3 max-pool NMS and per-image top-K.
 * Replaces nms_hm (DGDE/model/layers/utils.py:45-58) and select_topk (:61-100).
 *
 * dcd_nms_hm: out = heat * (maxpool3x3(heat) == heat), same shape (B,C,H,W).
 *
 * dcd_heatmap_topk: heat (B,C,H,W) -> scores (B,K) descending, inds (B,K) int64 = y*W+x of the winner,
 * clses (B,K) float = (c*K + rank_in_class)/K as the reference's true division yields (utils.py:91),
 * ys = floor(ind / W), xs = ind % W as float (utils.py:80-81).  fuse_nms != 0 applies the 3x3 NMS on
 * the fly (one launch for PostProcessor's nms_hm -> select_topk, detector_infer.py:102-106);
 * fuse_nms == 0 ranks `heat` as given (select_topk semantics).
 * Ties: lower linear index first (torch.topk leaves tie order unspecified; pinned by tests).
 * Requires K <= 128, K <= H*W, C*K <= 4096 and H*W <= 2^24.
 * ---------------------------------------------------------------------------------------------- */
int dcd_nms_hm(void *stream, const float *heat, int B, int C, int H, int W, float *out);
int dcd_heatmap_topk(void *stream, const float *heat, int B, int C, int H, int W, int K, int fuse_nms,
                     float *scores, int64_t *inds, float *clses, float *ys, float *xs, void *workspace,
                     size_t workspace_bytes);
size_t dcd_heatmap_topk_workspace_bytes(int B, int C, int H, int W, int K);

/* ------------------------------------------------------------------------------------------------
 * Point-of-interest gather.  Replaces select_point_of_interest (DGDE/model/layers/utils.py:120-145)
 * without the NCHW->NHWC copy.  feat (B,C,H,W), index (B,M) int64 linear y*W+x -> out (B,M,C).
 * Backward scatters grad_out (B,M,C) into grad_feat (B,C,H,W) with atomics (duplicate indices add);
 * grad_feat must be zero-filled by the caller.
 * ---------------------------------------------------------------------------------------------- */
int dcd_poi_gather(void *stream, const float *feat, const int64_t *index, int B, int C, int H, int W, int M,
                   float *out);
int dcd_poi_scatter_add(void *stream, const float *grad_out, const int64_t *index, int B, int C, int H, int W,
                        int M, float *grad_feat);
/* out (B, C, plane) += the gradient of 3x3 patches gathered at listed cells: grad_patches (B, C*9, M) [row c*9 + tap], base (B, M)
 * = index of the window's top-left element inside the plane, whose rows are `pitch` apart (head trunks evaluated at the object
 * centres only, dcd_amd/model/head/trunk_moments.py; no reference counterpart: the reference evaluates the trunks densely). */
int dcd_patch_scatter_add(void *stream, const float *grad_patches, const int64_t *base, int B, int C, int64_t plane, int pitch, int M,
                          float *out);

/* ------------------------------------------------------------------------------------------------
 * 3-D IoU of N box pairs (a logging metric of the train step).  Replaces get_iou_3d
 * (DGDE/model/layers/iou_loss.py:99-136, a per-object shapely loop on the host).
 * corners (N,8,3): 0..3 bottom face, 4..7 top face, camera frame (y down).  iou (N).
 * ---------------------------------------------------------------------------------------------- */
int dcd_iou3d(void *stream, const float *pred_corners, const float *target_corners, int N, float *iou);

/* ------------------------------------------------------------------------------------------------
 * The 1x1 output layers of the regression heads evaluated at listed rows: the reference applies one nn.Conv2d(256, out_j, 1)
 * per head to its trunk's dense output and concatenates the maps (DGDE/model/head/detector_predictor.py:84-101, :160-170,
 * :198-203); training reads them at the object centres only (detector_loss.py:231-233), so here
 *     y[r][ch0_j + o] = bias_j[o] + sum_k feat[trunk_j][r][k] * weight_j[o][k]        r = 0 .. R-1 (R = B * MAX_OBJECTS)
 * for all heads in ONE launch, and the backward in two (feature gradient per trunk; weight and bias gradients).
 * feat (T, R, K) trunk outputs at the rows, heads ordered by trunk, ch0_j = running sum of out_j, C = their total.
 * ---------------------------------------------------------------------------------------------- */
#define DCD_HEADS_MAX 16
typedef struct dcd_head_rows_args {
    int n_heads, T, R, K, C;
    int trunk[DCD_HEADS_MAX], ch0[DCD_HEADS_MAX], out[DCD_HEADS_MAX];
    const float *weight[DCD_HEADS_MAX];      /* (out_j, K) */
    const float *bias[DCD_HEADS_MAX];        /* (out_j) or NULL */
    const float *feat;                       /* (T, R, K) */
    float *y;                                /* (R, C): forward output */
    const float *grad_y;                     /* (R, C) */
    float *grad_feat;                        /* (T, R, K): every element written */
    float *grad_weight[DCD_HEADS_MAX];       /* (out_j, K) */
    float *grad_bias[DCD_HEADS_MAX];         /* (out_j) or NULL */
} dcd_head_rows_args;

int dcd_head_rows_forward(void *stream, const dcd_head_rows_args *args);
int dcd_head_rows_backward(void *stream, const dcd_head_rows_args *args);

/* ------------------------------------------------------------------------------------------------
 * Per-object rows of the training loss: every term of Loss_Computation.forward that is a sum over the
 * annotated objects (DGDE/model/head/detector_loss.py:405-583) and the decodes of
 * prepare_predictions that feed them (:217-403; DGDE/model/anno_encoder.py:93-128 encode_box3d, :130-145
 * decode_depth, :147-161 decode_location_flatten, :193-224 decode_depth_from_keypoints_batch, :226-252
 * decode_dimension, :254-304 decode_axes_orientation, :392-393 decode_kpts_2d_img), evaluated for all B*M object
 * slots at once: one wave per slot, empty slots masked out of every sum (they read the first annotated object so
 * that all arithmetic stays finite).  Supported configuration = the DGDE run: multi-bin orientation with 4 bins,
 * 10 box keypoints, 'inv_sigmoid' depth, 'exp' dimensions scaled by the class mean, L1 regression losses, corner
 * depth from the edge solver, depth and keypoint-depth uncertainties.
 *
 * Columns (sums over the slots in `sums`, per slot in `cols` (DCD_LOSS_ROWS_NCOL, B*M)), masks already applied:
 *   0 annotated, 1 GIoU loss, 2 IoU, 3 box valid, 4 depth L1 (log), 5 depth L1 with uncertainty, 6 truncated offset,
 *   7 offset, 8 multi-bin, 9 dimensions, 10 3-D IoU, 11 corners L1, 12 keypoints L1, 13 dense 2-D keypoints,
 *   14 their mask count, 15 dense 3-D keypoints, 16 their mask count, 17 pair depth L1 on valid pairs, 18 on invalid
 *   pairs (no gradient), 19 / 20 the pair counts, 21 pair relative error (log), 22 keypoint depth L1 (log),
 *   23 with uncertainty on visible groups, 24 uncertainty-scaled L1 on invisible groups (gradient to the uncertainty).
 *
 * Call order, all on one stream:
 *   dcd_loss_rows_prepare   fills kps_pred/kps_tgt (B*M,K,2 image pixels), kps3d_pred/kps3d_tgt (B*M,K,3), and, each
 *                           stored twice, rot (2,B*M), P_rows (2,B*M,3,4), kmask (2,B*M,K) -- the inputs of
 *                           dcd_edge_depth_forward; with kps_pred|kps_tgt and kps3d_pred|kps3d_tgt adjacent in memory
 *                           one solver call over 2*B*M rows serves both
 *   dcd_edge_depth_forward  (caller): predictions -> pair_depth (+ pair_idx); targets -> pair_mask
 *   dcd_loss_rows_forward   cols, corners_pred / corners_tgt (B*M,8,3), iou3d (B*M), sums (NCOL)
 *   dcd_loss_rows_backward  grad_sums (NCOL) -> grad_pois (B*M,C; every element written), grad_pair (B*M,NP)
 *   dcd_edge_depth_backward (caller): grad_pair -> grad_kps, grad_kps3d
 *   dcd_loss_rows_finish    grad_pois += the solver's gradients (4 * grad_kps on the 2-D keypoint channels)
 * ---------------------------------------------------------------------------------------------- */
#define DCD_LOSS_ROWS_NCOL 25
typedef struct dcd_loss_rows_args {
    int B, M, C, K, NP, num_classes;          /* images, slots per image, head channels, dense keypoints, pairs, classes */
    /* first channel of each regression head inside the C channels (Converter_key2channel, layers/utils.py:22-37) */
    int ch_box2d, ch_offset, ch_corner, ch_corner_unc, ch_dims, ch_ori_cls, ch_ori_off, ch_depth, ch_depth_unc,
        ch_kpts2d, ch_kpts3d;
    int trunc_log;                            /* TRUNCATION_OFFSET_LOSS: 1 = log(1 + l1), 0 = l1 */
    float depth_lo, depth_hi, unc_lo, unc_hi; /* DEPTH_RANGE, UNCERTAINTY_RANGE */
    float depth_weight;                       /* loss weight of depth_loss (inside the uncertainty form, :438-441) */
    float dim_weight[3];                      /* DIMENSION_WEIGHT */
    float down_ratio, kd_eps;                 /* BACKBONE.DOWN_RATIO; Anno_Encoder.EPS */
    const float *pois;                        /* (B*M, C) head outputs at the object centres */
    /* targets, (B, M, ...) as ParamsList stacks them */
    const uint8_t *reg_mask, *trunc_mask, *find_pcl, *ori_mask;
    const int32_t *cls_ids, *centers;         /* (B,M), (B,M,2) */
    const int64_t *pad_size;                  /* (B,2) */
    const float *bboxes, *locations, *rotys, *offset_3D, *dimensions, *orientations;   /* (..,4) (..,3) (..) (..,2) (..,3) (..,8) */
    const float *keypoints, *kp_depth_mask;   /* (B,M,10,3), (B,M,3) */
    const float *kpts2d, *kpts3d;             /* (B,M,K,3) each */
    const float *calib_P;                     /* (B,M,3,4) */
    const float *calib;                       /* (B,6): c_u, c_v, f_u, f_v, b_x, b_y */
    const float *dim_mean;                    /* (num_classes,3) */
    /* solver side */
    float *kps_pred, *kps_tgt, *kps3d_pred, *kps3d_tgt, *rot, *P_rows;
    uint8_t *kmask;
    const float *pair_depth, *pair_mask;      /* (B*M, NP) */
    /* forward outputs */
    float *cols, *corners_pred, *corners_tgt, *iou3d, *sums;
    /* backward */
    const float *grad_sums;
    float *grad_pois, *grad_pair;
    const float *grad_kps, *grad_kps3d;
} dcd_loss_rows_args;

int dcd_loss_rows_prepare(void *stream, const dcd_loss_rows_args *args);
int dcd_loss_rows_forward(void *stream, const dcd_loss_rows_args *args);
int dcd_loss_rows_backward(void *stream, const dcd_loss_rows_args *args);
int dcd_loss_rows_finish(void *stream, const dcd_loss_rows_args *args);

/* out = srcs[0] + ... + srcs[n-1] (n <= 16 device tensors of `numel` floats; `srcs` is a HOST array of
 * device pointers).  Replaces autograd's chain of n-1 pairwise gradient additions where one feature map feeds the twelve
 * head trunks (DGDE/model/head/detector_predictor.py:149-160 call every trunk on the same `features`). */
int dcd_sum_tensors(void *stream, const float *const *srcs, int n, float *out, int64_t numel);

/* The glue between `conv_offset_mask` and the deformable convolution in `DCN.forward` (DGDE/model/backbone/DCNv2/dcn_v2.py:118-123:
 * chunk into o1, o2, mask; offset = cat(o1, o2); mask = sigmoid(mask)) and its adjoint, one launch each.
 *   split: out (B, 3*taps, HW) -> offset (B, 2*taps, HW) = the first 2*taps channels, mask (B, taps, HW) = sigmoid of the rest
 *   merge: grad_out (B, 3*taps, HW) = [grad_offset | grad_mask * mask * (1 - mask)]                       (taps = dg * kh * kw) */
int dcd_dcn_offset_mask_split(void *stream, const float *out, float *offset, float *mask, int B, int taps, int64_t HW);
int dcd_dcn_offset_mask_merge(void *stream, const float *grad_offset, const float *grad_mask, const float *mask, float *grad_out, int B,
                              int taps, int64_t HW);

/* Batched fp32 product on the matrix pipe whose second operand is a set of SHIFTED VIEWS of one buffer:
 *     C[z][s] (M x N, row-major, ldc) = A[z] (M x K, row-major, lda) * B[z] (K x N)   over k in split s,   (+ bias[m] when nsplit == 1)
 *     B[z](k, n) = b_kcontig ? Bbase[z*strideB + b_off[n] + k] : Bbase[z*strideB + b_off[k] + n]
 * b_off: device array (N resp. K entries) of element offsets, any 4-byte aligned position -- e.g. row c of a zero-padded image
 * plane shifted by (dy, dx).  64-row tiles (M is the 64 channels of the head feature map).  It evaluates the 3x3-patch Gram
 * matrix of the regression-head input through its 25 autocorrelation matrices R_d[c,c'] = sum_p x[c,p] x[c',p+d] and the
 * gradient dX = sum_d K_d Xshift_d (dcd_amd/model/head/trunk_moments.py): what the reference spends on eleven dense
 * 3x3 convolutions + BatchNorm statistics of one shared input (DGDE/model/head/detector_predictor.py:104-120, 149-160).
 * Partials of split s go to C + z*strideC + s*strideCs; the caller sums them (in fp64). */
int dcd_sgemm_shifted(void *stream, const float *A, int lda, long long strideA, const float *Bbase, const long long *b_off,
                      long long strideB, int b_kcontig, const float *bias, float *C, int ldc, long long strideC,
                      long long strideCs, int M, int N, int K, int Z, int nsplit);

/* Batched fp32 GEMM on the matrix pipe (csrc/sgemm_f32.inc, 128 x 128 tiles):
 *     C[z] (M x N, row-major, ldc) = alpha * A[z] B[z]  (+ C[z] when accumulate)
 *     A(m,k) = a_kcontig ? A[m*lda + k] : A[k*lda + m]        B(k,n) = b_kcontig ? B[n*ldb + k] : B[k*ldb + n]
 * lower_only: tiles entirely above the diagonal are skipped (symmetric results; entries above the diagonal of the remaining
 * tiles are still written).  All pointers 16-byte aligned, lda / ldb / strides multiples of 4.  Used for the Schur complement
 * S = diag(c) - G^T diag(1/r) G of the transport layer's backward (GMW/lib/optimal_transport.py:93-100), written straight into
 * the buffer dcd_spd_solve factorises. */
int dcd_sgemm(void *stream, const float *A, int lda, long long strideA, int a_kcontig, const float *B, int ldb, long long strideB,
              int b_kcontig, float *C, int ldc, long long strideC, int M, int N, int K, int Z, float alpha, int accumulate,
              int lower_only);

/* Batched SPD solve  y[b] = S[b]^-1 r[b]  (one right-hand side per matrix) on the fp32 matrix pipe: blocked Cholesky whose panel
 * and trailing updates are MFMA GEMMs, forward substitution carried along with the factorisation, blocked backward substitution
 * (csrc/spd.hip).  Replaces `torch.cholesky` + `torch.cholesky_inverse` of the Schur complement in the backward of the
 * optimal-transport layer (GMW/lib/optimal_transport.py:102-128; n = 2628 edges).
 *   S: batch x rows x n fp32 (row stride n, rows >= n + 1, 16-byte aligned, n % 4 == 0): rows 0..n-1 = the symmetric positive
 *   definite matrix (lower triangle read, overwritten by the factor), row n = the right-hand side (overwritten).  y: batch x n.
 *   info (batch ints, zeroed by the caller, may be NULL): 0, or 1 + the first row of the 128-block in which a pivot was not
 *   positive (the reference raises there; here y is meaningless for that matrix). */
size_t dcd_spd_solve_workspace_bytes(int batch, int n);
int dcd_spd_solve(void *stream, float *S, float *y, int batch, int n, int rows, int *info, void *workspace, size_t workspace_bytes);

/* Context normalisation of GMW's feature extractor (`gcn`, GMW/model/yi2018cvpr/ops.py:5-17): x (rows, K) -> y = (x - mean) /
 * sqrt(var_unbiased + eps) per row, inv (rows) = the scale; backward from (grad_y, y, inv).  rows = batch * channels. */
int dcd_context_norm_forward(void *stream, const float *x, float *y, float *inv, int rows, int K, float eps);
int dcd_context_norm_backward(void *stream, const float *grad_y, const float *y, const float *inv, float *grad_x, int rows, int K);

/* ------------------------------------------------------------------------------------------------
 * Batch normalisation fused with the residual add and ReLU that follow it.  Replaces the stock-op chains
 *   bn -> relu            DGDE/model/backbone/dla_dcn.py:91-93 (BasicBlock), :272-283 (conv levels), :403-410
 *                         (DeformConv.actf); DGDE/model/head/detector_predictor.py:52-60,112-120 (head trunks)
 *   bn -> (+residual) -> relu   dla_dcn.py:95-99 (BasicBlock), :199-205 (Root)
 *   bn                    dla_dcn.py:237-240 (Tree.project)
 * i.e. torch.nn.functional.batch_norm + add + relu and their autograd.  x, residual, y: (B,C,HW) fp32 contiguous
 * (NCHW with HW = H*W).  Statistics are exchanged as fp64 per-channel sums so that a multi-GPU job all-reduces
 * `stats` / `sums` (C x 2 doubles) between the two calls -- the SyncBatchNorm of MODEL.USE_SYNC_BN
 * (DGDE/tools/plain_train_net.py:56-57) -- and passes the global element count.
 *
 * dcd_bn_stats:           stats[c] = (sum x, sum x^2) over this rank's B*HW elements.
 * dcd_bn_train_apply:     mean = S0/count, var = S1/count - mean^2 (biased); y = act((x-mean)*rsqrt(var+eps)*w + b
 *                         [+ residual]); writes save_mean / save_invstd (C); when running_mean/var are given,
 *                         running = (1-momentum)*running + momentum*(mean | var*count/(count-1)) and
 *                         *num_batches_tracked += 1 (all nullable).  weight / bias nullable (1 / 0).
 * dcd_bn_eval_apply:      same formula with the running statistics.
 * dcd_bn_backward_stats:  sums[c] = (sum dz, sum dz*(x-mean)), dz = grad_y * [y > 0] when `y` (the forward output,
 *                         ReLU fused) is given, dz = grad_y when y is NULL.
 * dcd_bn_backward_apply:  grad_x = (dz - S0/count - (x-mean)*invstd^2*S1/count) * invstd * w;
 *                         grad_residual = dz (nullable); grad_weight = S1*invstd, grad_bias = S0 of THIS call's sums
 *                         (nullable; under SyncBN pass the local sums' result, see dcd_amd/norm.py).
 * workspace: dcd_bn_workspace_bytes(C) bytes of device scratch for the two-stage reductions.
 * ---------------------------------------------------------------------------------------------- */
size_t dcd_bn_workspace_bytes(int C);
int dcd_bn_stats(void *stream, const float *x, int B, int C, int64_t HW, double *stats, void *workspace,
                 size_t workspace_bytes);

/* sums[c] = sum over b and positions of x[b][c][...] (fp64 two-stage sum, rounded once): a bias gradient
 * (torch.autograd's grad_bias of the reference's nn.Conv2d, DGDE/model/backbone/DCNv2/dcn_v2.py:111-128 conv_offset_mask).
 * One launch.  workspace: dcd_bn_workspace_bytes(C) bytes whose SECOND half starts with C arrival counters (unsigned) that must
 * be ZERO on entry; the call leaves them zero, so a buffer zeroed once can be reused by later calls ON THE SAME STREAM (calls on
 * different streams need different buffers). */
int dcd_channel_sums(void *stream, const float *x, int B, int C, int64_t HW, float *sums, void *workspace, size_t workspace_bytes);
int dcd_bn_train_apply(void *stream, const float *x, const float *residual, const float *weight, const float *bias,
                       const double *stats, double count, float *running_mean, float *running_var,
                       int64_t *num_batches_tracked, float momentum, float eps, int relu, float *y, float *save_mean,
                       float *save_invstd, int B, int C, int64_t HW);
int dcd_bn_eval_apply(void *stream, const float *x, const float *residual, const float *weight, const float *bias,
                      const float *running_mean, const float *running_var, float eps, int relu, float *y, int B, int C,
                      int64_t HW);
int dcd_bn_backward_stats(void *stream, const float *grad_y, const float *y, const float *x, const float *save_mean,
                          int B, int C, int64_t HW, double *sums, void *workspace, size_t workspace_bytes);
/* dcd_bn_backward for a layer with fused ReLU and NO residual that does not read the forward output: the ReLU mask y > 0 is
 * recomputed as fma(x, scale, shift) > 0 from the saved statistics, weight and bias -- the forward's own expression -- so the
 * backward reads two tensors (grad_y, x) per pass instead of three.  (The reference's nn.BatchNorm2d + F.relu keep and read the
 * output, DGDE/model/backbone/dla_dcn.py:76-101.) */
int dcd_bn_backward_relu_from_x(void *stream, const float *grad_y, const float *x, const float *weight, const float *bias,
                                const float *save_mean, const float *save_invstd, float *grad_x, float *grad_weight, float *grad_bias,
                                int B, int C, int64_t HW, void *workspace, size_t workspace_bytes);
/* The two halves of the same backward under data parallelism (sums all-reduced in between), mask recomputed from x: */
int dcd_bn_backward_stats_params_relu_from_x(void *stream, const float *grad_y, const float *x, const float *weight, const float *bias,
                                             const float *save_mean, const float *save_invstd, int B, int C, int64_t HW, double *sums,
                                             float *grad_weight, float *grad_bias, void *workspace, size_t workspace_bytes);
int dcd_bn_backward_apply_relu_from_x(void *stream, const float *grad_y, const float *x, const float *weight, const float *bias,
                                      const float *save_mean, const float *save_invstd, const double *sums, double count, float *grad_x,
                                      int B, int C, int64_t HW);
/* dcd_bn_backward_stats that also writes THIS rank's parameter gradients from the same sums (grad_weight[c] = sum dz*(x-mean) *
 * invstd, grad_bias[c] = sum dz; either may be NULL): under data parallelism they stay local (DDP averages parameter gradients)
 * while `sums` is all-reduced for the input gradient -- torch.nn.SyncBatchNorm's backward, torch/nn/modules/_functions.py. */
int dcd_bn_backward_stats_params(void *stream, const float *grad_y, const float *y, const float *x, const float *save_mean,
                                 const float *save_invstd, int B, int C, int64_t HW, double *sums, float *grad_weight,
                                 float *grad_bias, void *workspace, size_t workspace_bytes);
int dcd_bn_backward_apply(void *stream, const float *grad_y, const float *y, const float *x, const float *weight,
                          const float *save_mean, const float *save_invstd, const double *sums, double count,
                          float *grad_x, float *grad_residual, float *grad_weight, float *grad_bias, int B, int C,
                          int64_t HW);
/* Batch-norm parameters of the head's regression trunks from the Gram form (model/head/trunk_moments.py; the reference runs the
 * eleven dense conv + BatchNorm2d trunks, DGDE/model/head/detector_predictor.py:78-101).  Row r = (trunk, output channel), R rows,
 * K = 9 Cin.  WG (R, K + 1) = W [G | S1] and Wd (R, K) = W, both fp64.
 *   row_sums:           sums (R, 2) = (sum y, sum y^2) = (WG[r][K], sum_k WG[r][k] Wd[r][k])   [all-reduced by the caller under SyncBN]
 *   finalize_forward:   stats (R, 3) = (mean, biased variance clamped at 0, 1 / sqrt(var + eps)); scale = gamma / sqrt(var + eps),
 *                       shift = beta - mean * scale (fp32)
 *   finalize_backward:  (grad_scale, grad_shift) -> grad_sums (R, 2) fp64, grad_gamma, grad_beta
 *   grad_wg:            grad_WG (R, K + 1): columns k < K = grad_sums[r][1] * Wd[r][k], column K = grad_sums[r][0] */
int dcd_trunk_row_sums(void *stream, const double *WG, const double *Wd, int R, int K, double *sums);
int dcd_trunk_finalize_forward(void *stream, const double *sums, const float *gamma, const float *beta, double count, double eps, int R,
                               float *scale, float *shift, double *stats);
int dcd_trunk_finalize_backward(void *stream, const float *grad_scale, const float *grad_shift, const float *gamma, const double *stats,
                                double count, int R, double *grad_sums, float *grad_gamma, float *grad_beta);
int dcd_trunk_grad_wg(void *stream, const double *grad_sums, const double *Wd, int R, int K, double *grad_WG);

/* BN (+ReLU), training mode, evaluated at listed positions only (the regression-head trunks: the loss reads their output at
 * the object centres and, for one head, the border cells).  pos (B,N) int64 linear pixel indices; x_at, y_at (B,N,C).
 * stats NULL: statistics computed here (workspace needed); else the C x 2 combined (all-reduced) sums with the global count.
 * Backward: dcd_bn_at_backward_sums -> [all-reduce sums] -> dcd_bn_backward_apply(grad_y = NULL: zeros, y = NULL) gives the
 * dense part of grad_x, and dzk (B*N,C) is scatter-added at the positions (dcd_poi_scatter_add). */
int dcd_bn_at_forward(void *stream, const float *x, const int64_t *pos, const float *weight, const float *bias,
                      const double *stats, double count, float *running_mean, float *running_var,
                      int64_t *num_batches_tracked, float momentum, float eps, int relu, float *x_at, float *y_at,
                      float *save_mean, float *save_invstd, int B, int C, int64_t HW, int N, void *workspace,
                      size_t workspace_bytes);
int dcd_bn_at_backward_sums(void *stream, const float *grad_at, const float *x_at, const float *y_at, const float *weight,
                            const float *save_mean, const float *save_invstd, int relu, int total, int C, double *sums,
                            float *dzk);
/* Single-rank shortcuts (no statistics exchange): stats + apply fused into two launches, the apply kernels sum the
 * partials of the two-stage reduction themselves.  Same arithmetic and outputs as the call pairs above with
 * count = B*HW. */
int dcd_bn_train_forward(void *stream, const float *x, const float *residual, const float *weight, const float *bias,
                         float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                         int relu, float *y, float *save_mean, float *save_invstd, int B, int C, int64_t HW,
                         void *workspace, size_t workspace_bytes);
int dcd_bn_backward(void *stream, const float *grad_y, const float *y, const float *x, const float *weight,
                    const float *save_mean, const float *save_invstd, float *grad_x, float *grad_residual,
                    float *grad_weight, float *grad_bias, int B, int C, int64_t HW, void *workspace,
                    size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 / dilation 1 / groups 1 convolution: forward and backward-data (Winograd
 * F(2x2,3x3) on the fp32 matrix pipe).  Replaces the stock `nn.Conv2d(.., 3, padding=1, bias=False)` calls of
 *   DGDE/model/backbone/dla_dcn.py:76-82 (BasicBlock.conv1/conv2 at stride 1) and
 *   DGDE/model/head/detector_predictor.py:52-58,112-118 (the 64->256 trunks of the class / regression heads) and
 *   DGDE/model/backbone/DCNv2/dcn_v2.py:107-116 (DCN's `conv_offset_mask`, Cin -> 27 with a bias),
 * i.e. torch's cudnn/MIOpen convolution and its input gradient (weight gradient: dcd_conv3x3_wrw below).
 * weight (Cout,Cin,3,3), bias (Cout) or NULL (forward only); residual: NULL or an image of the output's shape that is added to the
 * result (it may BE the output buffer: a gradient accumulated in place, what autograd's separate addition would do).
 *                         backward_data = 0: input (B,Cin,H,W) -> output (B,Cout,H,W);
 *                         backward_data = 1: input = grad_output (B,Cout,H,W) -> output = grad_input (B,Cin,H,W).
 * Requires W % 4 == 0 and H even (bad-argument otherwise).  workspace: dcd_conv3x3_workspace_bytes(B, Cin, H, W, Cout) bytes
 * (transformed weights; partial images when few regions make the call split its contraction), dead after the call's
 * kernels complete.
 * ---------------------------------------------------------------------------------------------- */
size_t dcd_conv3x3_workspace_bytes(int B, int Cin, int H, int W, int Cout);
int dcd_conv3x3(void *stream, const float *input, const float *weight, const float *bias, const float *residual, float *output,
                int B, int Cin, int H, int W, int Cout, int backward_data, void *workspace, size_t workspace_bytes);

/* The same convolution with its Winograd-domain weights prepared ahead: dcd_conv3x3 transforms the weights on every call (one
 * small launch); a training step can prepare both directions in ONE launch during the forward call and hand the backward-data
 * call its weights (they do not change in between).  transform_weights writes dcd_conv3x3_weights_bytes(Cin, Cout, direction)
 * bytes per requested direction (either output may be NULL); dcd_conv3x3_prepared is dcd_conv3x3 with `transformed` (the
 * buffer of ITS direction) in place of `weight`; its workspace (dcd_conv3x3_workspace_bytes, may then be smaller) only holds the
 * partial images of a split contraction. */
size_t dcd_conv3x3_weights_bytes(int Cin, int Cout, int backward_data);
int dcd_conv3x3_transform_weights(void *stream, const float *weight, int Cin, int Cout, float *forward_out, float *backward_out);
/* transform_weights for MANY layers in one launch (a train step: once, right after the optimizer step, instead of one launch
 * per layer).  table: DEVICE array of `entries` records of five 64-bit words {weight pointer, forward_out pointer,
 * backward_out pointer, Cin, Cout}; either output pointer may be 0.  The table is read by the kernel: keep it alive and
 * unchanged until the launch has run. */
int dcd_conv3x3_transform_weights_table(void *stream, const long long *table, int entries);
int dcd_conv3x3_prepared(void *stream, const float *input, const float *transformed, const float *bias, const float *residual,
                         float *output, int B, int Cin, int H, int W, int Cout, int backward_data, void *workspace,
                         size_t workspace_bytes);

/* The same convolution with the Winograd-domain products in split-bf16 (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32
 * accumulate: ~2^-16 relative per product, the form DCD_PREC_BF16X3 names for the deformable convolution): transforms stay fp32,
 * operands stay fp32 in HBM.  Weights are prepared in their own layout (dcd_conv3x3_split_weights_bytes /
 * dcd_conv3x3_split_transform_weights: both directions, either pointer may be NULL); workspace holds the partial images of a
 * split contraction only.  Same shape rules as dcd_conv3x3.
 * precision: DCD_PREC_BF16X3, or DCD_PREC_BF16 = one product of bf16-rounded operands (the same prepared weights: their low halves
 * are not read). */
size_t dcd_conv3x3_split_weights_bytes(int Cin, int Cout, int backward_data);
int dcd_conv3x3_split_transform_weights(void *stream, const float *weight, int Cin, int Cout, void *forward_out, void *backward_out);
/* The split-layout transform for many layers in ONE launch: the table of dcd_conv3x3_transform_weights_table (five 64-bit words per
 * entry: weight, forward_out, backward_out, Cin, Cout) with the layer's split-layout buffers as outputs. */
int dcd_conv3x3_split_transform_weights_table(void *stream, const long long *table, int entries);
size_t dcd_conv3x3_split_workspace_bytes(int B, int Cin, int H, int W, int Cout);
int dcd_conv3x3_split_prepared(void *stream, const float *input, const void *transformed, const float *bias, const float *residual,
                               float *output, int B, int Cin, int H, int W, int Cout, int backward_data, int precision, void *workspace,
                               size_t workspace_bytes);

/* The ONE-product form (DCD_PREC_BF16, MODEL.FP16: DGDE/model/detector.py:34-36) as a DIRECT implicit GEMM on
 * v_mfma_f32_32x32x16_bf16 (csrc/conv_direct_bf16.inc): inputs and weights rounded to bf16 once (nearest even), no Winograd
 * transform -- on the bf16 matrix pipe the transform's vector / LDS work, not the multiplies, was the kernel's time -- fp32
 * accumulate, fp32 tensors in HBM.  Own weight layout (both directions, either pointer may be NULL; the table form takes the
 * five-word entries of dcd_conv3x3_transform_weights_table with buffers of dcd_conv3x3_bf16_weights_bytes); workspace: the partial
 * images of a split contraction.  W % 4 == 0; any Cin / Cout / H with (max(Cin, Cout) + 16) H W < 2^29; a split contraction
 * (dcd_conv3x3_bf16_workspace_bytes > 16) needs B Cout H W % 4 == 0.  bias only with backward_data == 0. */
size_t dcd_conv3x3_bf16_weights_bytes(int Cin, int Cout, int backward_data);
int dcd_conv3x3_bf16_transform_weights(void *stream, const float *weight, int Cin, int Cout, void *forward_out, void *backward_out);
int dcd_conv3x3_bf16_transform_weights_table(void *stream, const long long *table, int entries);
size_t dcd_conv3x3_bf16_workspace_bytes(int B, int Cin, int H, int W, int Cout);
int dcd_conv3x3_bf16_prepared(void *stream, const float *input, const void *transformed, const float *bias, const float *residual,
                              float *output, int B, int Cin, int H, int W, int Cout, int backward_data, void *workspace,
                              size_t workspace_bytes);

/* 1x1 convolutions (DLA's Roots and projections: DGDE/model/backbone/dla_dcn.py:187-207, 239-245) in the one-product form
 * (DCD_PREC_BF16; csrc/conv1x1_bf16.inc), fp32 tensors, operands rounded to bf16, fp32 accumulate:
 *   dcd_conv1x1_bf16      output (B, M, HW) = A x, x = the concatenation along the channels of n_inputs (<= 4) tensors
 *                         (B, channels[i], HW) that is never formed; A[m][k] = weight[m * ldw + k], or weight[k * ldw + m] when
 *                         `transposed` (the input gradient of a column slice: pass weight + first column).  inputs / channels are
 *                         HOST arrays.  HW % 4 == 0, channels[i] % 16 == 0.
 *   dcd_conv1x1_wrw_bf16  grad_weight[o * ldw + c] = sum_{b, p} grad_output[b][o][p] input[b][c][p] (overwrites the O x C block;
 *                         partial sums in a fixed order); workspace: dcd_conv1x1_wrw_bf16_workspace_bytes. */
int dcd_conv1x1_bf16(void *stream, const float *weight, int ldw, int transposed, int n_inputs, const float *const *inputs,
                     const int *channels, float *output, int B, int M, long long HW);
size_t dcd_conv1x1_wrw_bf16_workspace_bytes(int B, int O, int C, long long HW);
int dcd_conv1x1_wrw_bf16(void *stream, const float *grad_output, const float *input, float *grad_weight, int ldw, int B, int O, int C,
                         long long HW, void *workspace, size_t workspace_bytes);

/* The same layers in EXACT fp32 (csrc/conv1x1_f32.inc, v_mfma_f32_32x32x2_f32; round 6): same arguments, same restrictions.  They
 * replace, in the fp32 train step, the batched library GEMMs behind `torch.cat` + `nn.Conv2d(.., 1)` of the reference's Root
 * (DGDE/model/backbone/dla_dcn.py:199-205): one launch over all concatenated inputs instead of one GEMM per input. */
int dcd_conv1x1_f32(void *stream, const float *weight, int ldw, int transposed, int n_inputs, const float *const *inputs,
                    const int *channels, float *output, int B, int M, long long HW);
size_t dcd_conv1x1_wrw_f32_workspace_bytes(int B, int O, int C, long long HW);
int dcd_conv1x1_wrw_f32(void *stream, const float *grad_output, const float *input, float *grad_weight, int ldw, int B, int O, int C,
                        long long HW, void *workspace, size_t workspace_bytes);

/* 3x3 / stride 2 / pad 1 convolution in exact fp32 (csrc/conv_s2_f32.inc, round 6): the first convolution of every DLA level
 * (DGDE/model/backbone/dla_dcn.py:76-78, 313-326: `nn.Conv2d(c, k, 3, stride=2, padding=1, bias=False)`).  No LDS for the
 * activations: a lane's nine taps x four output pixels are 27 registers of one channel.  input (B, Cin, H, W) -> output
 * (B, Cout, H/2, W/2); H even, W % 8 == 0, Cin % 16 == 0. */
int dcd_conv3x3_s2_f32(void *stream, const float *input, const float *weight, float *output, int B, int Cin, int H, int W, int Cout);
/* ... its input gradient (grad_output (B, Cout, H/2, W/2) -> grad_input (B, Cin, H, W), every element written; Cout % 16 == 0 too) and
 * its weight gradient (grad_weight (Cout, Cin, 3, 3) overwritten; partial sums per pixel split in a fixed order; H % 4 == 0;
 * workspace: dcd_conv3x3_s2_f32_wrw_workspace_bytes). */
int dcd_conv3x3_s2_f32_backward_data(void *stream, const float *grad_output, const float *weight, float *grad_input, int B, int Cin, int H,
                                     int W, int Cout);
size_t dcd_conv3x3_s2_f32_wrw_workspace_bytes(int B, int Cin, int H, int W, int Cout);
int dcd_conv3x3_s2_f32_wrw(void *stream, const float *input, const float *grad_output, float *grad_weight, int B, int Cin, int H, int W,
                           int Cout, void *workspace, size_t workspace_bytes);

/* Weight gradient of the same convolution (torch's `convolution_backward(..., output_mask=[0,1,0])` for those call sites),
 * also in the Winograd domain: grad_weight (Cout,Cin,3,3) = correlation of input (B,Cin,H,W) with grad_output (B,Cout,H,W).
 * Overwrites grad_weight; the partial sums of the workgroups are added in a fixed order (bitwise reproducible).
 * Same shape requirements; workspace: dcd_conv3x3_wrw_workspace_bytes(B, Cin, H, W, Cout) bytes, dead after the call.
 * precision: DCD_PREC_F32 exact; DCD_PREC_BF16 the sixteen Winograd-domain products on v_mfma_f32_32x32x16_bf16 with both transformed
 * operands rounded to bf16 (fp32 accumulate over the tiles); DCD_PREC_BF16X3 runs the exact kernel. */
size_t dcd_conv3x3_wrw_workspace_bytes(int B, int Cin, int H, int W, int Cout);
int dcd_conv3x3_wrw(void *stream, const float *input, const float *grad_output, float *grad_weight, int B, int Cin, int H, int W,
                    int Cout, int precision, void *workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------
 * The low-channel, full-resolution convolutions of DLA-34's stem (csrc/stem.hip, v_mfma_f32_16x16x4_f32):
 *   base_layer Conv2d(3, 16, 7, padding 3, bias=False)  DGDE/model/backbone/dla_dcn.py:236-240   (Cin 3,  ksize 7)
 *   level0     Conv2d(16, 16, 3, padding 1, bias=False) DGDE/model/backbone/dla_dcn.py:241-242   (Cin 16, ksize 3)
 * i.e. torch's conv2d / convolution_backward for those two call sites; any other (Cin, Cout, ksize) is a bad argument.
 * stride 1, padding ksize/2, W % 4 == 0.  weight (16,Cin,k,k).  backward_data (Cin 16 only): input = grad_output -> output =
 * grad_input.  dcd_conv_stem_wrw overwrites grad_weight (partials are combined with float atomics: order-dependent rounding).
 * Workspaces: dcd_conv_stem_workspace_bytes / dcd_conv_stem_wrw_workspace_bytes, dead after the call's kernels complete.
 * ---------------------------------------------------------------------------------------------- */
size_t dcd_conv_stem_workspace_bytes(int Cin, int Cout, int ksize);
int dcd_conv_stem(void *stream, const float *input, const float *weight, float *output, int B, int Cin, int H, int W, int Cout,
                  int ksize, int backward_data, void *workspace, size_t workspace_bytes);
size_t dcd_conv_stem_wrw_workspace_bytes(int Cin, int Cout, int ksize);
int dcd_conv_stem_wrw(void *stream, const float *input, const float *grad_output, float *grad_weight, int B, int Cin, int H, int W,
                      int Cout, int ksize, void *workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------
 * Depthwise transposed convolution of IDAUp: nn.ConvTranspose2d(C, C, 2f, stride=f, padding=f/2, groups=C, bias=False)
 * (DGDE/model/backbone/dla_dcn.py:416-421; weights from fill_up_weights :386-395, learnable), f in {2,4,8}.
 * x (B,C,H,W) -> y (B,C,H*f,W*f); weight (C,1,2f,2f).  Requires (W*f) % 4 == 0.  Backward overwrites grad_x and grad_weight.
 * ---------------------------------------------------------------------------------------------- */
int dcd_upsample_dw_forward(void *stream, const float *x, const float *weight, float *y, int B, int C, int H, int W, int f);
/* y = up(x) + skip: the sum that feeds IDAUp's node (`layers[i] = node(layers[i] + layers[i-1])`, dla_dcn.py:430-436) leaves the
 * up-sampling kernel directly; skip (B, C, H*f, W*f). */
int dcd_upsample_dw_forward_add(void *stream, const float *x, const float *weight, const float *skip, float *y, int B, int C, int H,
                                int W, int f);

/* 2x2 / stride-2 max pooling of (planes, H, W) maps (H even, W % 4 == 0): the `downsample` of every DLA Tree
 * (`nn.MaxPool2d(stride, stride=stride)`, DGDE/model/backbone/dla_dcn.py:228).  The backward re-derives the arg-max from x
 * (first maximum in scan order, NaN takes over: the stock kernel's rule) instead of keeping an index tensor. */
int dcd_maxpool2x2_forward(void *stream, const float *x, float *y, int64_t planes, int H, int W);
int dcd_maxpool2x2_backward(void *stream, const float *x, const float *grad_y, float *grad_x, int64_t planes, int H, int W);
int dcd_upsample_dw_backward(void *stream, const float *x, const float *weight, const float *grad_y, float *grad_x,
                             float *grad_weight, int B, int C, int H, int W, int f);

/* ------------------------------------------------------------------------------------------------
 * Training-target encoding of a whole batch on the device (csrc/targets.hip).
 * Replaces the numpy work of `KITTIDataset.__getitem__` (DGDE/data/datasets/kitti.py:354-606): box / key-point projection and
 * visibility, truncated-object centres (`approx_proj_center`, kitti_utils.py:1040-1078), Gaussian heat map
 * (`gaussian_radius`, `draw_umich_gaussian`, `draw_umich_gaussian_2D`, DGDE/model/heatmap_coder.py:37-124), multi-bin
 * orientation (`encode_alpha_multibin`, kitti.py:225-244) and the border walk (`get_edge_utils`, kitti.py:165-223).
 *   objs     (B, M, 16) float64: truncation, occlusion, box x1 y1 x2 y2, h, w, l, t x y z, ry, alpha, find_pcl, class id
 *            (the values an `Object3d` holds, kitti_utils.py:61-113; box and t are float32 there and are rounded as such)
 *   kpts3d   (B, M, n_extra, 3) float64  object-frame key points, already shifted by -h/2 (kitti_utils.py:112)
 *   P        (B, 3, 4) float64; img_size (B, 2) int32 (w, h) before padding; n_obj (B) int32 objects per image (<= M)
 *   outputs  26 device pointers, caller-allocated, ALL ZERO-FILLED except ori_mask (which the caller keeps as ones), in this
 *            order: hm (B,n_classes,H/down,W/down) f32 | cls_ids (B,M) i32 | target_centers (B,M,2) i32 | gt_bboxes (B,M,4) f32 |
 *            2d_bboxes (B,M,4) f32 | keypoints (B,M,10,3) f32 | keypoints_depth_mask (B,M,3) f32 | extra_kpts_2d (B,M,K,3) f32 |
 *            extra_kpts_3d (B,M,K,3) f32 | extra_kpts_depth_mask (B,M,K) f32 | Calib_P (B,M,3,4) f32 | find_pcl (B,M) u8 |
 *            dimensions (B,M,3) f32 | locations (B,M,3) f32 | rotys (B,M) f32 | alphas (B,M) f32 | offset_3D (B,M,2) f32 |
 *            occlusions (B,M) f64 | truncations (B,M) f64 | orientations (B,M,8) f32 | reg_mask (B,M) u8 | trunc_mask (B,M) u8 |
 *            reg_weight (B,M) f32 | pad_size (B,2) i64 | edge_indices (B,2(W+H)/down,2) i64 | edge_len (B) i64;  K = n_extra + 10.
 * The configuration is the reference's: INPUT.HEATMAP_CENTER '3D', KEYPOINT_VISIBLE_MODIFY, ADJUST_BOUNDARY_HEATMAP,
 * CONSIDER_OUTSIDE_OBJS with APPROX_3D_CENTER 'intersect', ORIENTATION 'multi-bin' with 4 bins (anything else: bad argument).
 * ---------------------------------------------------------------------------------------------- */
int dcd_encode_targets(void *stream, const double *objs, const double *kpts3d, const double *P, const int32_t *img_size,
                       const int32_t *n_obj, int B, int M, int n_extra, int in_w, int in_h, int down_ratio, double filter_trunc,
                       double filter_size, double edge_heatmap_ratio, int num_bin, int n_classes, void *const *outputs,
                       int n_outputs);

/* ------------------------------------------------------------------------------------------------
 * The optimizer end of the train step (csrc/optim.hip): `torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)` followed by
 * `AdamW.step()` (DGDE/engine/trainer.py:144-147; DGDE/solver/__init__.py:10-62: AdamW, betas (0.9, 0.99), one learning rate per
 * parameter) over LISTS of fp32 tensors.  The host arrays of pointers / element counts are read during the call only (they travel
 * in the kernel arguments); every tensor has fewer than 2^31 elements.
 *
 * dcd_clip_grad_norm_scalars: 2-norm over all listed gradients -> scal[0] = the norm, scal[1] = min(1, max_norm / (norm + 1e-6))
 *   (1 when max_norm <= 0: no clipping), scal[2] = 1 when the norm is not finite else 0; three floats on the device.  The gradients
 *   are not touched here.  workspace: dcd_clip_adamw_workspace_bytes(ntensors, numel) bytes, dead after the call.
 * dcd_adamw_apply: one parameter group.  Unless scal[2] is set (then NOTHING changes: the library's found_inf contract), every
 *   listed step counter (a float on the device each, as the library's capturable AdamW keeps them) += 1 and
 *     g *= scal[1] (written back only when < 1)         p -= lr * weight_decay * p
 *     m = beta1 m + (1 - beta1) g                       v = beta2 v + (1 - beta2) g^2
 *     p -= (lr / (1 - beta1^step)) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 *   in the library's fused kernel's order and types (double hyper-parameters against float operands).  lr: a float on the device. */
size_t dcd_clip_adamw_workspace_bytes(int ntensors, const int64_t *numel);
int dcd_clip_grad_norm_scalars(void *stream, int ntensors, const void *const *grads, const int64_t *numel, float max_norm, void *workspace,
                               size_t workspace_bytes, float *scal);
int dcd_adamw_apply(void *stream, int ntensors, void *const *params, void *const *grads, void *const *exp_avg, void *const *exp_avg_sq,
                    void *const *steps, const int64_t *numel, const float *lr, double beta1, double beta2, double eps, double weight_decay,
                    const float *scal);

#ifdef __cplusplus
}
#endif
#endif /* DCD_HIP_H */
