"""Eval-time decode (mirrors `PostProcessor`, DGDE/model/head/detector_infer.py:27-243).

heat-map NMS + top-K (one fused HIP launch) -> POI gather -> threshold -> decode -> edge-constraint depth
(mean over all 2628 pairs, one launch) -> KITTI rows [cls, alpha, x1,y1,x2,y2, h,w,l, x,y,z, roty, score].
Like the reference the decode assumes batch size 1 (calib[0], zero batch_idxs: detector_infer.py:173,186,221).
"""
import torch
from torch import nn
from torch.nn import functional as F

from dcd_amd.structures.params_3d import stack_field
from dcd_amd.model.anno_encoder import Anno_Encoder
from dcd_amd.model.layers.utils import Converter_key2channel, select_point_of_interest, select_topk


def make_post_processor(cfg):
    anno_encoder = Anno_Encoder(cfg)
    key2channel = Converter_key2channel(keys=cfg.MODEL.HEAD.REGRESSION_HEADS, channels=cfg.MODEL.HEAD.REGRESSION_CHANNELS)
    return PostProcessor(cfg=cfg, anno_encoder=anno_encoder, key2channel=key2channel)


class PostProcessor(nn.Module):
    """Stages of the decode, each a method: `_detections` (NMS + top-K + threshold + POI gather), `_decode_heads` (per-head
    decodes of the kept rows), `_fuse_depths` (uncertainty-weighted direct / keypoint depths), then the edge-constraint depth,
    the final locations and the (N, 14) rows.  Outputs, dict keys and quirks are the reference's."""

    def __init__(self, cfg, anno_encoder, key2channel):
        super().__init__()
        self.anno_encoder, self.key2channel = anno_encoder, key2channel
        test, head = cfg.TEST, cfg.MODEL.HEAD
        self.det_threshold, self.max_detection = test.DETECTIONS_THRESHOLD, test.DETECTIONS_PER_IMG
        self.eval_dis_iou, self.eval_depth = test.EVAL_DIS_IOUS, test.EVAL_DEPTH
        self.pred_2d, self.use_only_extra_kpts = test.PRED_2D, test.USE_ONLY_EXTRA_KPTS
        self.uncertainty_as_conf, self.generate_data = test.UNCERTAINTY_AS_CONFIDENCE, test.GENERATE_GMW
        self.extra_kpts_num, self.output_depth = head.EXTRA_KPTS_NUM, head.OUTPUT_DEPTH
        self.img_width = cfg.INPUT.WIDTH_TRAIN
        heads = set(key2channel.keys)
        self.pred_direct_depth = 'depth' in heads
        self.depth_with_uncertainty = 'depth_uncertainty' in heads
        self.regress_keypoints = 'corner_offset' in heads
        self.keypoint_depth_with_uncertainty = 'corner_uncertainty' in heads
        self.use_extra_kpts = 'extra_kpts_2d' in heads
        self.gen_infer_records = []

    _OPTIONAL_FIELDS = ("cls_ids", "target_centers", "dimensions", "rotys", "locations", "offset_3D", "extra_kpts_2d",
                        "extra_kpts_3d", "reg_mask", "Calib_P")

    def prepare_targets(self, targets, test):
        pad_size = stack_field(targets, "pad_size")
        out = {"calib": [t.get_field("calib") for t in targets], "pad_size": pad_size,
               "size": torch.stack([torch.as_tensor(t.size) for t in targets]).to(pad_size.device)}
        if not test:
            for name in self._OPTIONAL_FIELDS:
                if all(t.has_field(name) for t in targets):
                    out[name] = stack_field(targets, name)
        return out

    # ---- stage 1: which cells are detections ------------------------------------------------------------------------
    def _detections(self, heat, reg, predictions=None):
        """3x3 max-pool NMS + per-image top-K in one launch (nms_hm -> select_topk, detector_infer.py:101-106), score
        threshold (the one host sync of the decode), regression vectors of the kept cells.  Any batch: `image_of` is the image
        each kept row came from (rows stay in (image, rank) order).  A predictor with `sparse_eval_heads` has done the top-K
        itself and hands over the heads at those cells (`reg_pois`, `topk`)."""
        if predictions is not None and predictions.get('topk') is not None:
            scores, cell_idx, classes, ys, xs = predictions['topk']
            vectors = predictions['reg_pois'].reshape(-1, predictions['reg_pois'].shape[-1])
            if vectors.shape[0] != scores.numel():
                raise ValueError("predictions['topk'] holds %d cells, predictions['reg_pois'] %d rows" % (scores.numel(), vectors.shape[0]))
        else:
            scores, cell_idx, classes, ys, xs = select_topk(heat, K=self.max_detection, fuse_nms=True)
            vectors = select_point_of_interest(heat.shape[0], cell_idx, reg).view(-1, reg.shape[1])
        per_image = scores.shape[1]
        scores = scores.view(-1)
        keep = (scores >= self.det_threshold).nonzero(as_tuple=True)[0]
        if keep.numel() == 0:
            return None
        centres = torch.stack((xs.view(-1), ys.view(-1)), dim=1)
        return {"scores": scores[keep], "classes": classes.view(-1)[keep], "centres": centres[keep], "vectors": vectors[keep],
                "image_of": torch.div(keep, per_image, rounding_mode="floor")}

    @staticmethod
    def _nothing_detected(like, vis):
        empty = like.new_zeros
        vis['keypoints'], vis['proj_center'] = empty(0, 20), empty(0, 2)
        info = {'dis_ious': None, 'depth_errors': None, 'vis_scores': empty(0), 'uncertainty_conf': empty(0),
                'estimated_depth_error': empty(0)}
        return empty(0, 14), info, vis

    # ---- stage 2: per-head decodes -----------------------------------------------------------------------------------
    def _decode_heads(self, det, ctx, reg, vis, image_of=None):
        """image_of = None: the reference's one-image decode (first image's padding / size / calibration for every row);
        image_of given (forward_batch): every row with its own image's."""
        enc, sl, vec = self.anno_encoder, self.key2channel, det["vectors"]
        out = {"offset_3d": vec[:, sl('3d_offset')]}
        vis['proj_center'] = det["centres"] + out["offset_3d"]
        if image_of is None:
            out["box2d"] = enc.decode_box2d_fcos(det["centres"], F.relu(vec[:, sl('2d_dim')]), ctx["pad_size"], ctx["size"])
        else:                                                  # the same arithmetic with per-row padding and image size (:82-89)
            off2d = F.relu(vec[:, sl('2d_dim')])
            c2 = det["centres"].view(-1, 2)
            box2d = torch.cat((c2 - off2d[:, :2], c2 + off2d[:, 2:]), dim=1) * enc.down_ratio - ctx["pad_size"][image_of].repeat(1, 2)
            lim = ctx["size"][image_of].to(box2d).repeat(1, 2) - 1
            out["box2d"] = torch.min(box2d.clamp(min=0), lim)
        out["dims"] = enc.decode_dimension(det["classes"], vec[:, sl('3d_dim')])
        out["orientation"] = torch.cat((vec[:, sl('ori_cls')], vec[:, sl('ori_offset')]), dim=1)
        if self.pred_direct_depth:
            out["direct_depth"] = enc.decode_depth(vec[:, sl('depth')].squeeze(-1))
        if self.depth_with_uncertainty:
            out["direct_sigma"] = vec[:, sl('depth_uncertainty')].exp()
            vis['depth_uncertainty'] = reg[:, sl('depth_uncertainty'), ...].squeeze(1) if reg is not None else None   # (dense map: visualiser only)
        if self.regress_keypoints:
            out["corner_offsets"] = vec[:, sl('corner_offset')].view(-1, 10, 2)
            if image_of is None:
                out["corner_depths"] = enc.decode_depth_from_keypoints_batch(out["corner_offsets"], out["dims"], ctx["calib"])
            else:                                              # each row with its own image's focal length (no rank quirk: :206-207
                f_u = enc._calib_table(ctx["calib"], vec.device)[image_of, 2]        # only bites when one call holds several images)
                out["corner_depths"] = enc.decode_depth_from_keypoints_batch(out["corner_offsets"], out["dims"], ctx["calib"], f_u=f_u)
            vis['keypoints'] = out["corner_offsets"]
        if self.keypoint_depth_with_uncertainty:
            out["corner_sigma"] = vec[:, sl('corner_uncertainty')].exp()
        return out

    # ---- stage 3: inverse-uncertainty weighted depth (detector_infer.py:150-170) -------------------------------------
    def _fuse_depths(self, dec, vis):
        if self.pred_direct_depth and self.depth_with_uncertainty:
            depths = torch.cat((dec["direct_depth"].unsqueeze(1), dec["corner_depths"]), dim=1)
            sigma = torch.cat((dec["direct_sigma"], dec["corner_sigma"]), dim=1)
        else:
            depths, sigma = dec["corner_depths"].clone(), dec["corner_sigma"].clone()
        weights = 1 / sigma
        vis['min_uncertainty'] = weights.argmax(dim=1)
        weights = weights / weights.sum(dim=1, keepdim=True)
        return (depths * weights).sum(dim=1), (weights * sigma).sum(dim=1)

    def forward(self, predictions, targets, features=None, test=False, refine_module=None):
        return self._decode(predictions, targets, test, batched=False)

    def forward_batch(self, predictions, targets, features=None, test=False):
        """The decode for a whole batch at once (round 4; BASELINE config 4 runs it on 16 images): ONE NMS + top-K launch, one
        POI gather, one threshold (one host sync), one edge-solver call, every row decoded with ITS OWN image's padding, size and
        intrinsics -- i.e. exactly what `forward` gives when called image by image (the reference's loop,
        DGDE/engine/inference.py:59-84, TEST.IMS_PER_BATCH = 1), row for row, in (image, rank) order.  Returns
        (rows (N, 14), info, vis, image_of (N,)); `engine.gen_data.infer_records_batch` splits the records per image after ONE
        device-to-host copy."""
        return self._decode(predictions, targets, test, batched=True)

    def _decode(self, predictions, targets, test, batched):
        if self.eval_dis_iou or self.eval_depth:
            raise NotImplementedError("TEST.EVAL_DIS_IOUS / TEST.EVAL_DEPTH call functions the reference never defines "
                                      "(detector_infer.py:95,98)")
        heat, reg = predictions['cls'], predictions['reg']
        enc = self.anno_encoder
        ctx = self.prepare_targets(targets, test=test)
        vis = {'heat_map': heat.clone()}
        det = self._detections(heat, reg, predictions)
        if det is None:
            out = self._nothing_detected(heat, vis)
            return out + (heat.new_zeros(0).long(),) if batched else out
        rows_image = det["image_of"] if batched else None
        dec = self._decode_heads(det, ctx, reg, vis, rows_image)
        fused_depth, depth_error = self._fuse_depths(dec, vis)

        centres, offset = det["centres"], dec["offset_3d"]
        # one-image form: batch size 1, like the reference (:173)
        image_of = rows_image if batched else fused_depth.new_zeros(fused_depth.shape[0]).long()
        ctx["rows_image"] = rows_image
        coarse = enc.decode_location_flatten(centres, offset, fused_depth, ctx["calib"], ctx["pad_size"], image_of)
        rotys, alphas = enc.decode_axes_orientation(dec["orientation"], coarse)
        rotys, alphas = rotys.view(-1, 1), alphas.view(-1, 1)
        scores = det["scores"].view(-1, 1)

        # the reported depth comes from the dense edge constraints: mean over all keypoint pairs (detector_infer.py:183-184)
        edge_depth = self.compute_pairs_kpts_depth(ctx, det["vectors"], centres, offset, rotys, vis)
        bottom = enc.decode_location_flatten(centres, offset, edge_depth, ctx["calib"], ctx["pad_size"], image_of)
        locations = bottom.clone()
        locations[:, 1] += dec["dims"][:, 1] / 2                               # object centre -> bottom-face centre (KITTI)
        if self.generate_data:
            self.generate_infer_data(ctx, det["vectors"], centres, offset, dec.get("corner_offsets"), dec["dims"], vis,
                                     dec["box2d"], rotys, locations, scores)

        dims_hwl = dec["dims"].roll(shifts=-1, dims=1)                          # (l, h, w) -> (h, w, l)
        raw_scores = scores.clone()
        if self.uncertainty_as_conf and depth_error is not None:
            confidence = 1 - torch.clamp(depth_error, min=0.01, max=1)
            scores = torch.nan_to_num(scores * confidence.view(-1, 1), nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        else:
            confidence = depth_error = None
        rows = torch.cat([det["classes"].view(-1, 1), alphas, dec["box2d"], dims_hwl, locations, rotys, scores], dim=1)
        info = {'dis_ious': None, 'depth_errors': None, 'uncertainty_conf': confidence, 'estimated_depth_error': depth_error,
                'vis_scores': raw_scores}
        return (rows, info, vis, rows_image) if batched else (rows, info, vis)

    def _image_kpts(self, targets, pois, pred_bbox_points, pred_offset_3D):
        k2c = self.key2channel
        kp2d = pois[:, k2c('extra_kpts_2d')].reshape((-1, self.extra_kpts_num + 10, 2))
        rows_image = targets.get("rows_image")
        pad = targets["pad_size"] if rows_image is None else targets["pad_size"][rows_image].unsqueeze(1)
        real_2d = (kp2d + (pred_bbox_points + pred_offset_3D).unsqueeze(1).expand_as(kp2d)) * 4 - pad
        kp3d = pois[:, k2c('extra_kpts_3d')].reshape((kp2d.shape[0], -1, 3))
        return real_2d, kp3d

    def _rows_P(self, targets, like):
        """(n, 3, 4) projection matrices of the rows: image 0's for all (one-image decode, detector_infer.py:186,237) or each
        row's own image's (forward_batch).  One host-to-device copy per distinct set of intrinsics."""
        import numpy as np
        rows_image = targets.get("rows_image")
        if rows_image is None:
            P = torch.as_tensor(targets['calib'][0].P, dtype=torch.float32, device=like.device)
            return P.unsqueeze(0).expand(like.shape[0], -1, -1)
        key = tuple(np.asarray(c.P, dtype=np.float32).tobytes() for c in targets['calib'])
        if getattr(self, "_P_cache", (None, None))[0] != (key, str(like.device)):
            tab = torch.as_tensor(np.stack([np.asarray(c.P, dtype=np.float32) for c in targets['calib']]), device=like.device)
            self._P_cache = ((key, str(like.device)), tab)
        return self._P_cache[1][rows_image]

    def compute_pairs_kpts_depth(self, targets, pois, pred_bbox_points, pred_offset_3D, pred_rots, vis_pred):
        real_2d, kp3d = self._image_kpts(targets, pois, pred_bbox_points, pred_offset_3D)
        P = self._rows_P(targets, real_2d)
        pairs_depths, _ = self.anno_encoder.decode_pairs_kpts_depth(real_2d, kp3d, pred_rots, P)
        vis_pred['pred_extra_kpts_2d'] = real_2d
        vis_pred['pred_extra_kpts_3d'] = kp3d
        return pairs_depths.mean(1)

    def generate_infer_data(self, targets, pois, pred_bbox_points, pred_offset_3D, pred_keypoint_offset, pred_dim,
                            vis_pred, pred_box2d=None, pred_rotys=None, pred_locations=None, scores=None):
        """K-normalised keypoints for GMW inference (detector_infer.py:227-243); image 0's intrinsics for every
        object, as in the reference (:237).  The records (SURVEY.md section 8f-2 schema) are kept on device in
        `vis_pred['gen_*']`; the caller serialises them."""
        real_2d, kp3d = self._image_kpts(targets, pois, pred_bbox_points, pred_offset_3D)
        if targets.get("rows_image") is None:
            K = torch.as_tensor(targets['calib'][0].P[:, :3], dtype=real_2d.dtype, device=real_2d.device)
            kn = torch.stack(((real_2d[:, :, 0] - K[0, 2]) / K[0, 0], (real_2d[:, :, 1] - K[1, 2]) / K[1, 1]), dim=-1)
        else:                                                  # every row with its own image's intrinsics
            P = self._rows_P(targets, real_2d)
            kn = torch.stack(((real_2d[:, :, 0] - P[:, 0, 2:3]) / P[:, 0, 0:1], (real_2d[:, :, 1] - P[:, 1, 2:3]) / P[:, 1, 1:2]), dim=-1)
        vis_pred['gen_pred_extra_kpts_2d'] = kn
        vis_pred['gen_pred_extra_kpts_3d'] = kp3d
        if pred_box2d is not None:
            vis_pred['gen_box'] = pred_box2d
            vis_pred['gen_dim'] = pred_dim.roll(shifts=-1, dims=1)
            vis_pred['gen_pred_rot'] = pred_rotys
            vis_pred['gen_pred_location'] = pred_locations
            vis_pred['gen_score'] = scores
