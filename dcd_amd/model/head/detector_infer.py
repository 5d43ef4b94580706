"""Eval-time decode (mirrors `PostProcessor`, DGDE/model/head/detector_infer.py:27-243).

heat-map NMS + top-K (one fused HIP launch) -> POI gather -> threshold -> decode -> edge-constraint depth
(mean over all 2628 pairs, one launch) -> KITTI rows [cls, alpha, x1,y1,x2,y2, h,w,l, x,y,z, roty, score].
Like the reference the decode assumes batch size 1 (calib[0], zero batch_idxs: detector_infer.py:173,186,221).
"""
import torch
from torch import nn
from torch.nn import functional as F

from dcd_amd.model.anno_encoder import Anno_Encoder
from dcd_amd.model.layers.utils import Converter_key2channel, select_point_of_interest, select_topk


def make_post_processor(cfg):
    anno_encoder = Anno_Encoder(cfg)
    key2channel = Converter_key2channel(keys=cfg.MODEL.HEAD.REGRESSION_HEADS, channels=cfg.MODEL.HEAD.REGRESSION_CHANNELS)
    return PostProcessor(cfg=cfg, anno_encoder=anno_encoder, key2channel=key2channel)


class PostProcessor(nn.Module):
    def __init__(self, cfg, anno_encoder, key2channel):
        super().__init__()
        self.anno_encoder = anno_encoder
        self.key2channel = key2channel
        self.det_threshold = cfg.TEST.DETECTIONS_THRESHOLD
        self.max_detection = cfg.TEST.DETECTIONS_PER_IMG
        self.eval_dis_iou = cfg.TEST.EVAL_DIS_IOUS
        self.eval_depth = cfg.TEST.EVAL_DEPTH
        self.extra_kpts_num = cfg.MODEL.HEAD.EXTRA_KPTS_NUM
        self.output_depth = cfg.MODEL.HEAD.OUTPUT_DEPTH
        self.pred_2d = cfg.TEST.PRED_2D
        self.pred_direct_depth = 'depth' in self.key2channel.keys
        self.depth_with_uncertainty = 'depth_uncertainty' in self.key2channel.keys
        self.regress_keypoints = 'corner_offset' in self.key2channel.keys
        self.keypoint_depth_with_uncertainty = 'corner_uncertainty' in self.key2channel.keys
        self.use_extra_kpts = 'extra_kpts_2d' in self.key2channel.keys
        self.use_only_extra_kpts = cfg.TEST.USE_ONLY_EXTRA_KPTS
        self.uncertainty_as_conf = cfg.TEST.UNCERTAINTY_AS_CONFIDENCE
        self.generate_data = cfg.TEST.GENERATE_GMW
        self.img_width = cfg.INPUT.WIDTH_TRAIN
        self.gen_infer_records = []

    def prepare_targets(self, targets, test):
        pad_size = torch.stack([t.get_field("pad_size") for t in targets])
        calibs = [t.get_field("calib") for t in targets]
        size = torch.stack([torch.as_tensor(t.size) for t in targets]).to(pad_size.device)
        out = dict(calib=calibs, size=size, pad_size=pad_size)
        if test:
            return out
        for name in ("cls_ids", "target_centers", "dimensions", "rotys", "locations", "offset_3D", "extra_kpts_2d",
                     "extra_kpts_3d", "reg_mask", "Calib_P"):
            if all(t.has_field(name) for t in targets):
                out[name] = torch.stack([t.get_field(name) for t in targets])
        return out

    def forward(self, predictions, targets, features=None, test=False, refine_module=None):
        pred_heatmap, pred_regression = predictions['cls'], predictions['reg']
        batch = pred_heatmap.shape[0]
        enc, k2c = self.anno_encoder, self.key2channel
        tv = self.prepare_targets(targets, test=test)
        calib, pad_size, img_size = tv['calib'], tv['pad_size'], tv['size']
        if self.eval_dis_iou or self.eval_depth:
            raise NotImplementedError("TEST.EVAL_DIS_IOUS / TEST.EVAL_DEPTH call functions the reference never defines "
                                      "(detector_infer.py:95,98)")
        dis_ious = depth_errors = None
        visualize_preds = {'heat_map': pred_heatmap.clone()}

        # 3x3 max-pool NMS + per-image top-K in one launch (nms_hm -> select_topk, detector_infer.py:101-106)
        scores, indexs, clses, ys, xs = select_topk(pred_heatmap, K=self.max_detection, fuse_nms=True)
        pred_bbox_points = torch.cat([xs.view(-1, 1), ys.view(-1, 1)], dim=1)
        pois = select_point_of_interest(batch, indexs, pred_regression).view(-1, pred_regression.shape[1])

        scores = scores.view(-1)
        keep = (scores >= self.det_threshold).nonzero(as_tuple=True)[0]     # the host sync of the decode
        if keep.numel() == 0:
            z = scores.new_zeros
            visualize_preds['keypoints'] = z(0, 20)
            visualize_preds['proj_center'] = z(0, 2)
            eval_utils = {'dis_ious': dis_ious, 'depth_errors': depth_errors, 'vis_scores': z(0),
                          'uncertainty_conf': z(0), 'estimated_depth_error': z(0)}
            return z(0, 14), eval_utils, visualize_preds

        scores = scores.index_select(0, keep)
        clses = clses.view(-1).index_select(0, keep)
        pred_bbox_points = pred_bbox_points.index_select(0, keep)
        pois = pois.index_select(0, keep)

        pred_2d_reg = F.relu(pois[:, k2c('2d_dim')])
        pred_offset_3D = pois[:, k2c('3d_offset')]
        pred_dimensions_offsets = pois[:, k2c('3d_dim')]
        pred_orientation = torch.cat((pois[:, k2c('ori_cls')], pois[:, k2c('ori_offset')]), dim=1)
        visualize_preds['proj_center'] = pred_bbox_points + pred_offset_3D
        pred_box2d = enc.decode_box2d_fcos(pred_bbox_points, pred_2d_reg, pad_size, img_size)
        pred_dimensions = enc.decode_dimension(clses, pred_dimensions_offsets)

        if self.pred_direct_depth:
            pred_direct_depths = enc.decode_depth(pois[:, k2c('depth')].squeeze(-1))
        if self.depth_with_uncertainty:
            pred_direct_uncertainty = pois[:, k2c('depth_uncertainty')].exp()
            visualize_preds['depth_uncertainty'] = pred_regression[:, k2c('depth_uncertainty'), ...].squeeze(1)
        if self.regress_keypoints:
            pred_keypoint_offset = pois[:, k2c('corner_offset')].view(-1, 10, 2)
            pred_keypoints_depths = enc.decode_depth_from_keypoints_batch(pred_keypoint_offset, pred_dimensions, calib)
            visualize_preds['keypoints'] = pred_keypoint_offset
        if self.keypoint_depth_with_uncertainty:
            pred_keypoint_uncertainty = pois[:, k2c('corner_uncertainty')].exp()

        if self.pred_direct_depth and self.depth_with_uncertainty:
            combined_depths = torch.cat((pred_direct_depths.unsqueeze(1), pred_keypoints_depths), dim=1)
            combined_uncertainty = torch.cat((pred_direct_uncertainty, pred_keypoint_uncertainty), dim=1)
        else:
            combined_depths = pred_keypoints_depths.clone()
            combined_uncertainty = pred_keypoint_uncertainty.clone()
        depth_weights = 1 / combined_uncertainty
        visualize_preds['min_uncertainty'] = depth_weights.argmax(dim=1)
        depth_weights = depth_weights / depth_weights.sum(dim=1, keepdim=True)
        pred_depths = torch.sum(combined_depths * depth_weights, dim=1)
        estimated_depth_error = torch.sum(depth_weights * combined_uncertainty, dim=1)

        batch_idxs = pred_depths.new_zeros(pred_depths.shape[0]).long()
        coarse_loc = enc.decode_location_flatten(pred_bbox_points, pred_offset_3D, pred_depths, calib, pad_size, batch_idxs)
        pred_rotys, pred_alphas = enc.decode_axes_orientation(pred_orientation, coarse_loc)
        clses = clses.view(-1, 1)
        pred_alphas = pred_alphas.view(-1, 1)
        pred_rotys = pred_rotys.view(-1, 1)
        scores = scores.view(-1, 1)

        # depth from the dense edge constraints: mean over all keypoint pairs (detector_infer.py:183-184, :215-225)
        pred_depths = self.compute_pairs_kpts_depth(tv, pois, pred_bbox_points, pred_offset_3D, pred_rotys, visualize_preds)
        pred_locations = enc.decode_location_flatten(pred_bbox_points, pred_offset_3D, pred_depths, calib, pad_size, batch_idxs)
        pred_locations = torch.cat((pred_locations[:, :1], pred_locations[:, 1:2] + pred_dimensions[:, 1:2] / 2,
                                    pred_locations[:, 2:]), dim=1)
        if self.generate_data:
            self.generate_infer_data(tv, pois, pred_bbox_points, pred_offset_3D, pred_keypoint_offset, pred_dimensions,
                                     visualize_preds, pred_box2d, pred_rotys, pred_locations, scores)

        pred_dimensions = pred_dimensions.roll(shifts=-1, dims=1)     # (l,h,w) -> (h,w,l)
        vis_scores = scores.clone()
        if self.uncertainty_as_conf and estimated_depth_error is not None:
            uncertainty_conf = 1 - torch.clamp(estimated_depth_error, min=0.01, max=1)
            scores = scores * uncertainty_conf.view(-1, 1)
            scores = torch.where(torch.isnan(scores), torch.zeros_like(scores), scores)
        else:
            uncertainty_conf, estimated_depth_error = None, None

        result = torch.cat([clses, pred_alphas, pred_box2d, pred_dimensions, pred_locations, pred_rotys, scores], dim=1)
        eval_utils = {'dis_ious': dis_ious, 'depth_errors': depth_errors, 'uncertainty_conf': uncertainty_conf,
                      'estimated_depth_error': estimated_depth_error, 'vis_scores': vis_scores}
        return result, eval_utils, visualize_preds

    def _image_kpts(self, targets, pois, pred_bbox_points, pred_offset_3D):
        k2c = self.key2channel
        kp2d = pois[:, k2c('extra_kpts_2d')].reshape((-1, self.extra_kpts_num + 10, 2))
        real_2d = (kp2d + (pred_bbox_points + pred_offset_3D).unsqueeze(1).expand_as(kp2d)) * 4 - targets["pad_size"]
        kp3d = pois[:, k2c('extra_kpts_3d')].reshape((kp2d.shape[0], -1, 3))
        return real_2d, kp3d

    def compute_pairs_kpts_depth(self, targets, pois, pred_bbox_points, pred_offset_3D, pred_rots, vis_pred):
        real_2d, kp3d = self._image_kpts(targets, pois, pred_bbox_points, pred_offset_3D)
        P = torch.as_tensor(targets['calib'][0].P, dtype=torch.float32, device=real_2d.device)
        P = P.unsqueeze(0).expand(real_2d.shape[0], -1, -1)
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
        K = torch.as_tensor(targets['calib'][0].P[:, :3], dtype=real_2d.dtype, device=real_2d.device)
        kn = torch.stack(((real_2d[:, :, 0] - K[0, 2]) / K[0, 0], (real_2d[:, :, 1] - K[1, 2]) / K[1, 1]), dim=-1)
        vis_pred['gen_pred_extra_kpts_2d'] = kn
        vis_pred['gen_pred_extra_kpts_3d'] = kp3d
        if pred_box2d is not None:
            vis_pred['gen_box'] = pred_box2d
            vis_pred['gen_dim'] = pred_dim.roll(shifts=-1, dims=1)
            vis_pred['gen_pred_rot'] = pred_rotys
            vis_pred['gen_pred_location'] = pred_locations
            vis_pred['gen_score'] = scores
