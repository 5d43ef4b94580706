"""Training loss of the DGDE head (mirrors `Loss_Computation`, DGDE/model/head/detector_loss.py:23-666).

Same inputs ({'cls','reg'} + list of ParamsList), same 13 loss keys and log keys, same arithmetic per term.
What changed is how it executes on the GPU:
  * no host synchronisation at all in training: all B*M object slots are processed, empty ones masked out of every sum
    (the reference compacts with ~40 boolean-mask gathers, each a sync); data generation keeps the compact lists;
    every later "loss[mask].sum()" is a masked sum, so no further sync happens inside the step;
  * focal / GIoU / POI gather / pair-depth solve / 3-D IoU are single HIP launches (dcd_amd.ops);
  * the log dict is materialised with ONE device->host copy instead of ~20 `.item()` calls (:589-630);
  * the three pair-depth decodes whose results the reference never reads (`pairs_kpt_depths_gt/_2d/_3d`,
    :378-380) and the unused ensemble-MAE block (:559-578) are not computed.
Reference quirk kept on purpose: the pair mask comes from the top-1500 ordering of the TARGET 2-D keypoints while
`pairs_kpt_depths_all` is ordered by the PREDICTED ones (:378 vs :381), and they are combined elementwise (:188-204).
"""
import os

import torch
from torch.nn import functional as F

from dcd_amd.structures.params_3d import stack_field
from dcd_amd import ops
from dcd_amd.model.anno_encoder import Anno_Encoder
from dcd_amd.model.head.depth_losses import RegWeightedL1Loss, Berhu_Loss, Inverse_Sigmoid_Loss, Log_L1_Loss
from dcd_amd.model.layers.focal_loss import FocalLoss
from dcd_amd.model.layers.iou_loss import IOULoss, get_iou_3d
from dcd_amd.model.layers.utils import Converter_key2channel, select_point_of_interest
from dcd_amd.utils.comm import get_world_size


def _total(x):
    """Sum of every element, as a row reduction followed by a sum of the rows.  A flat `.sum()` of a large tensor is a
    multi-block reduction in ATen: partials in a scratch buffer plus a semaphore cleared by hipMemsetAsync.  Inside a captured
    HIP graph that memset node is not reliably ordered before the kernel on this ROCm stack (csrc/zero_fill.h): from the second
    replay on, the flat sums of the (slots, 1500) pair-depth tensors came back as 0, 2x, or another reduction's value while all
    inputs were bit-identical to the eager evaluation (tools/debug_step_graph5.py).  Row reductions and single-block sums use
    neither the scratch buffer nor the semaphore."""
    return x.reshape(x.shape[0], -1).sum(dim=1).sum() if x.dim() > 1 else x.sum()


class _MaskedSums:
    """Collects per-object columns and reduces them together: one stack, one product with the stacked masks, one column sum
    instead of (mask product, sum, weight, division) per loss term -- the terms are ~4 tiny launches each in the forward and
    as many in the backward, and a dependent launch costs ~2 us however small it is.  2-D terms enter as their
    row sums (the first stage of `_total`); the column sum over the B*M slots is its second stage."""

    def __init__(self):
        self.cols, self.masks = [], []

    def add(self, vec, mask):
        self.cols.append(vec.reshape(-1))
        self.masks.append(mask.reshape(-1))
        return len(self.cols) - 1

    def reduce(self):
        return (torch.stack(self.cols, dim=1) * torch.stack(self.masks, dim=1)).sum(dim=0)


def make_loss_evaluator(cfg):
    return Loss_Computation(cfg=cfg)


# Everything `Loss_Computation` reads from the config, as tables (the reference assigns them one by one,
# detector_loss.py:24-98): attribute <- cfg path.
_HEAD = "MODEL.HEAD."
_CFG_ATTRS = {
    "max_objs": "DATASETS.MAX_OBJECTS", "orien_bin_size": "INPUT.ORIENTATION_BIN_SIZE", "fp16": "MODEL.FP16",
    "batch_weight_factor": "MODEL.BATCH_WEIGHT_FACTOR", "is_gen": "TEST.GENERATE_GMW",
    "center_sample": _HEAD + "CENTER_SAMPLE", "regress_area": _HEAD + "REGRESSION_AREA", "heatmap_type": _HEAD + "HEATMAP_TYPE",
    "corner_depth_sp": _HEAD + "SUPERVISE_CORNER_DEPTH", "loss_keys": _HEAD + "LOSS_NAMES",
    "uncertainty_range": _HEAD + "UNCERTAINTY_RANGE", "trunc_offset_loss_type": _HEAD + "TRUNCATION_OFFSET_LOSS",
    "uncertainty_weight": _HEAD + "UNCERTAINTY_WEIGHT", "keypoint_xy_weights": _HEAD + "KEYPOINT_XY_WEIGHT",
    "keypoint_norm_factor": _HEAD + "KEYPOINT_NORM_FACTOR", "extra_kpts_num": _HEAD + "EXTRA_KPTS_NUM",
    "modify_invalid_keypoint_depths": _HEAD + "MODIFY_INVALID_KEYPOINT_DEPTH", "corner_loss_depth": _HEAD + "CORNER_LOSS_DEPTH",
}
# flag <- name that must appear in MODEL.HEAD.LOSS_NAMES / in the regression-head channel map
_LOSS_FLAGS = {"compute_direct_depth_loss": "depth_loss", "compute_keypoint_depth_loss": "keypoint_depth_loss",
               "compute_pairs_kpts_depth_loss": "pairs_kpts_depth_loss", "compute_weighted_depth_loss": "weighted_avg_depth_loss",
               "compute_corner_loss": "corner_loss", "separate_trunc_offset": "trunc_offset_loss"}
_HEAD_FLAGS = {"pred_direct_depth": "depth", "depth_with_uncertainty": "depth_uncertainty", "compute_keypoint_corner": "corner_offset",
               "compute_extra_kpts_corner": "extra_kpts_2d", "corner_with_uncertainty": "corner_uncertainty"}
_DEPTH_LOSSES = {"berhu": Berhu_Loss, "inv_sig": Inverse_Sigmoid_Loss, "log": Log_L1_Loss, "L1": lambda: F.l1_loss}


class Loss_Computation():
    eps = 1e-5

    def __init__(self, cfg):
        head = cfg.MODEL.HEAD
        for attr, path in _CFG_ATTRS.items():
            node = cfg
            for part in path.split("."):
                node = getattr(node, part)
            setattr(self, attr, node)
        self.anno_encoder = Anno_Encoder(cfg)
        self.key2channel = Converter_key2channel(keys=head.REGRESSION_HEADS, channels=head.REGRESSION_CHANNELS)
        for flag, name in _LOSS_FLAGS.items():
            setattr(self, flag, name in self.loss_keys)
        for flag, name in _HEAD_FLAGS.items():
            setattr(self, flag, name in self.key2channel.keys)
        self.world_size = get_world_size()
        self.multibin = cfg.INPUT.ORIENTATION == 'multi-bin'
        self.dim_weight = torch.as_tensor(head.DIMENSION_WEIGHT).view(1, 3)
        self.loss_weights = dict(zip(head.LOSS_NAMES, head.INIT_LOSS_WEIGHT))

        # LOSS_TYPE = [heat-map, regression, 2-D box, depth]
        _, self.reg_loss, box_loss, depth_loss = head.LOSS_TYPE[:4]
        if depth_loss not in _DEPTH_LOSSES:
            raise ValueError("MODEL.HEAD.LOSS_TYPE[3] = %r" % (depth_loss,))
        self.cls_loss_fnc = FocalLoss(head.LOSS_PENALTY_ALPHA, head.LOSS_BETA, cfg=cfg)
        self.iou_loss = IOULoss(loss_type=box_loss)
        self.depth_loss = _DEPTH_LOSSES[depth_loss]()
        self.reg_loss_fnc = F.l1_loss if self.reg_loss == 'L1' else F.smooth_l1_loss
        self.keypoint_loss_fnc = self.extra_kpts_3d_loss_fnc = F.l1_loss
        self.extra_kpts_2d_loss_fnc = RegWeightedL1Loss()

        self.use_graph = os.environ.get("DCD_LOSS_GRAPH", "1") != "0"
        # the per-object rows as one kernel (csrc/loss_rows.hip) when the configuration is the one it implements (the DGDE
        # run); any other configuration, and data generation, evaluate the same terms op by op below
        enc = self.anno_encoder
        self.fused_rows = (os.environ.get("DCD_LOSS_ROWS", "1") != "0" and not self.is_gen and self.multibin
                           and self.orien_bin_size == 4 and self.reg_loss == 'L1' and box_loss == 'giou' and depth_loss == 'L1'
                           and all(getattr(self, f) for f in list(_LOSS_FLAGS) + list(_HEAD_FLAGS) if f != "compute_weighted_depth_loss")
                           and self.corner_loss_depth == 'edges' and self.uncertainty_range is not None
                           and enc.depth_mode == 'inv_sigmoid' and enc.depth_range is not None
                           and enc.dim_modes[0] == 'exp' and not enc.dim_modes[2]
                           and self.trunc_offset_loss_type in ('L1', 'log'))
        self._rows_spec = None
        self._graphs = {}                      # input-shape key -> (graphed callable, {'loss_keys', 'log_names'})
        self._wcache = (None, None)
        self._mcache = (None, None)
        self.gen_data = {k: [] for k in ('kpts_2d', 'kpts_3d', 'pred_rot', 'gt_location', 'pred_location', 'weight_img', 'img_idx')}

    # ------------------------------------------------------------------------------------------
    def prepare_targets(self, targets):
        """Stack the per-image fields into batch tensors (detector_loss.py:106-146)."""
        def stack(name):
            return stack_field(targets, name)
        out = {k: stack(f) for k, f in (
            ('cls_ids', 'cls_ids'), ('target_centers', 'target_centers'), ('bboxes', '2d_bboxes'),
            ('keypoints', 'keypoints'), ('extra_kpts_2d', 'extra_kpts_2d'), ('extra_kpts_3d', 'extra_kpts_3d'),
            ('Calib_P', 'Calib_P'), ('dimensions', 'dimensions'), ('locations', 'locations'), ('rotys', 'rotys'),
            ('alphas', 'alphas'), ('pad_size', 'pad_size'), ('reg_mask', 'reg_mask'), ('reg_weight', 'reg_weight'),
            ('offset_3D', 'offset_3D'), ('trunc_mask', 'trunc_mask'), ('orientations', 'orientations'),
            ('keypoints_depth_mask', 'keypoints_depth_mask'), ('extra_kpts_depth_mask', 'extra_kpts_depth_mask'),
            ('find_pcl', 'find_pcl'), ('ori_mask', 'ori_mask'))}
        out['calib'] = [t.get_field("calib") for t in targets]
        out['img_idx'] = [t.get_field("img_idx") for t in targets]
        if all(t.has_field("ori_img") for t in targets):   # carried by the reference, unused by the loss
            out['ori_imgs'] = stack("ori_img")
        return stack("hm"), out

    def generate_data(self, targets_variables, pred_extra_kpts_2D_img, pred_extra_kpts_3D_real, reg_mask_gt,
                      pred_rotys_3D, target_locations_3D, pred_locations_3D):
        """Collect the per-object records GMW trains on (detector_loss.py:148-173).  Image 0's intrinsics
        normalise every object, as in the reference (:150)."""
        K = torch.as_tensor(targets_variables['calib'][0].P[:, :3], dtype=pred_extra_kpts_2D_img.dtype,
                            device=pred_extra_kpts_2D_img.device)
        kps = pred_extra_kpts_2D_img
        kn = torch.stack(((kps[:, :, 0] - K[0, 2]) / K[0, 0], (kps[:, :, 1] - K[1, 2]) / K[1, 1]), dim=-1)
        # one device buffer per iteration, ONE device-to-host copy: [k2 (K*2) | k3 (K*3) | roty | gt xyz | pred xyz | count row]
        n, nk = kn.shape[0], kn.shape[1]
        f32 = torch.float32
        rows = torch.cat((kn.detach().reshape(n, nk * 2).to(f32), pred_extra_kpts_3D_real.detach().reshape(n, nk * 3).to(f32),
                          pred_rotys_3D.detach().reshape(n, 1).to(f32), target_locations_3D.detach().reshape(n, 3).to(f32),
                          pred_locations_3D.detach().reshape(n, 3).to(f32)), dim=1)
        counts_row = torch.zeros((1, rows.shape[1]), dtype=f32, device=rows.device)
        cnt = reg_mask_gt.sum(-1).to(f32)
        counts_row[0, :cnt.numel()] = cnt
        host = torch.cat((rows, counts_row)).cpu().numpy()
        body, counts = host[:n], host[n, :cnt.numel()]
        self.gen_data['kpts_2d'].append(body[:, :nk * 2].reshape(n, nk, 2).tolist())
        self.gen_data['kpts_3d'].append(body[:, nk * 2:nk * 5].reshape(n, nk, 3).tolist())
        ids = []
        for i, num in enumerate(counts.tolist()):
            ids += [targets_variables['img_idx'][i]] * int(num)
        self.gen_data['img_idx'].append(ids)
        self.gen_data['pred_rot'].append(body[:, nk * 5].tolist())
        self.gen_data['gt_location'].append(body[:, nk * 5 + 1:nk * 5 + 4].tolist())
        self.gen_data['pred_location'].append(body[:, nk * 5 + 4:nk * 5 + 7].tolist())

    # ------------------------------------------------------------------------------------------
    def compute_pairs_kpts_loss(self, preds, pred_targets, acc, one):
        """Dense-keypoint L1 terms and the pair-depth term (detector_loss.py:176-215) as row sums in `acc` (all masks are
        already inside the rows, so their column mask is `one`); `_core` forms the reference's ratios from the column sums."""
        m2d = pred_targets['extra_kpts_2d_mask'].float()
        m3d = pred_targets['extra_kpts_3d_mask'].float()
        l2d = self.extra_kpts_2d_loss_fnc(preds['extra_kpts_2d'], pred_targets['extra_kpts_2d'], pred_targets['depth_3D']) * m2d
        l3d = self.extra_kpts_3d_loss_fnc(preds['extra_kpts_3d'], pred_targets['extra_kpts_3d'], reduction='none').sum(dim=2) * m3d
        pred = preds['pairs_kpt_depths_all']
        pmask = preds['pairs_kpt_depths_mask'] > 0
        found = pred_targets['find_pcl'].bool().unsqueeze(-1)
        valid = (pmask & found).float()
        invalid = ((~pmask) & found).float()
        target = pred_targets['depth_3D'].unsqueeze(-1).expand_as(pred)
        reg = self.reg_loss_fnc(pred, target, reduction='none')
        mae = ((pred.detach() - target).abs() / target) * valid
        cols = {'l2d': l2d, 'm2d': m2d, 'l3d': l3d, 'm3d': m3d, 'valid_l': reg * valid, 'invalid_l': reg.detach() * invalid,
                'n_valid': valid, 'n_invalid': invalid, 'mae': mae}
        return {k: acc.add(v.sum(dim=1), one) for k, v in cols.items()}, mae

    # ------------------------------------------------------------------------------------------
    def prepare_predictions(self, targets_variables, predictions):
        """Select the annotated objects, gather their predictions and decode them (detector_loss.py:217-403)."""
        pred_regression = predictions['reg']
        reg_pois = predictions.get('reg_pois')                 # (B, M, C): heads already evaluated at the object centres
        if reg_pois is not None:
            batch, channel = reg_pois.shape[0], reg_pois.shape[2]
        else:
            batch, channel, feat_h, feat_w = pred_regression.shape
        enc = self.anno_encoder
        tv = targets_variables

        reg_mask_gt = tv["reg_mask"]
        flat_mask = reg_mask_gt.reshape(-1).bool()
        M = reg_mask_gt.shape[1]
        if self.is_gen:
            sel = flat_mask.nonzero(as_tuple=True)[0]            # data generation needs the compact lists (host sync)
            obj_valid = torch.ones_like(sel, dtype=torch.bool)
        else:
            # No host sync: every one of the B*M slots is processed.  Empty slots point at the first annotated object, so
            # all arithmetic below stays finite, and `obj_valid` zeroes their contribution in every reduction -- the sums
            # equal the reference's sums over the compacted object list (detector_loss.py:217-230).
            slots = torch.arange(batch * M, device=flat_mask.device)
            first = torch.argmax(flat_mask.to(torch.uint8))
            sel = torch.where(flat_mask, slots, first)
            obj_valid = flat_mask
        n_obj = sel.numel()

        def pick(t, *shape):
            return t.reshape(batch * M, *shape).index_select(0, sel)

        batch_idxs = torch.div(sel, M, rounding_mode='floor')
        points = pick(tv["target_centers"], 2).float()
        # 2-D box targets in FCOS (l,t,r,b) form
        boxes = pick(tv['bboxes'], 4)
        box_h, box_w = boxes[:, 3] - boxes[:, 1], boxes[:, 2] - boxes[:, 0]
        target_regression_2D = torch.cat((points - boxes[:, :2], boxes[:, 2:] - points), dim=1).float()
        mask_regression_2D = (box_h > 0) & (box_w > 0) & obj_valid

        target_clses = pick(tv["cls_ids"])
        target_depths_3D = pick(tv['locations'][..., -1])
        target_rotys_3D = pick(tv['rotys'])
        target_offset_3D = pick(tv["offset_3D"], 2)
        target_dimensions_3D = pick(tv['dimensions'], 3)
        target_center = pick(tv['target_centers'], 2)
        target_pad_size = tv['pad_size'].index_select(0, batch_idxs)
        target_ori_mask = pick(tv['ori_mask'])
        target_orientation_3D = pick(tv['orientations'], tv['orientations'].shape[-1])
        target_locations_3D = enc.decode_location_flatten(points, target_offset_3D, target_depths_3D, tv['calib'],
                                                          tv['pad_size'], batch_idxs)
        target_corners_3D = enc.encode_box3d(target_rotys_3D, target_dimensions_3D, target_locations_3D).float()
        target_bboxes_3D = torch.cat((target_locations_3D, target_dimensions_3D, target_rotys_3D[:, None]), dim=1)

        # predictions at the object centres: direct strided gather, no NHWC copy (utils.py:120-145)
        if reg_pois is not None:
            pois = reg_pois.reshape(-1, channel).index_select(0, sel)
        else:
            pois = select_point_of_interest(batch, tv["target_centers"], pred_regression).view(-1, channel).index_select(0, sel)
        k2c = self.key2channel
        pred_regression_2D = F.relu(pois[:, k2c('2d_dim')]).float()
        pred_offset_3D = pois[:, k2c('3d_offset')].float()
        pred_dimensions_offsets_3D = pois[:, k2c('3d_dim')].float()
        pred_orientation_3D = torch.cat((pois[:, k2c('ori_cls')], pois[:, k2c('ori_offset')]), dim=1)
        pred_dimensions_3D = enc.decode_dimension(target_clses, pred_dimensions_offsets_3D)

        targets = {'reg_2D': target_regression_2D, 'reg_2D_mask': mask_regression_2D, 'offset_3D': target_offset_3D,
                   'depth_3D': target_depths_3D, 'orien_3D': target_orientation_3D, 'dims_3D': target_dimensions_3D,
                   'corners_3D': target_corners_3D, 'width_2D': box_w, 'rotys_3D': target_rotys_3D,
                   'cat_3D': target_bboxes_3D, 'trunc_mask_3D': pick(tv['trunc_mask']), 'height_2D': box_h,
                   'location_3D': target_locations_3D, 'find_pcl': pick(tv["find_pcl"]), 'ori_mask': target_ori_mask,
                   'Calib_P': pick(tv["Calib_P"], 3, 4)}
        preds = {'reg_2D': pred_regression_2D, 'offset_3D': pred_offset_3D, 'orien_3D': pred_orientation_3D,
                 'dims_3D': pred_dimensions_3D}
        reg_nums = {'reg_2D': mask_regression_2D.sum(), 'reg_3D': obj_valid.sum(), 'reg_obj': obj_valid.sum()}
        weights = {'object_weights': pick(tv["reg_weight"])}
        targets['obj_valid'] = obj_valid

        if self.pred_direct_depth:
            preds['depth_3D'] = enc.decode_depth(pois[:, k2c('depth')].squeeze(-1), targets['Calib_P'])
        if self.depth_with_uncertainty:
            u = pois[:, k2c('depth_uncertainty')].squeeze(-1)
            if self.uncertainty_range is not None:
                u = torch.clamp(u, min=self.uncertainty_range[0], max=self.uncertainty_range[1])
            preds['depth_uncertainty'] = u

        if self.compute_keypoint_corner:
            kp = pick(tv["keypoints"], tv["keypoints"].shape[2], 3)
            targets['keypoints'] = kp[..., :2]
            targets['keypoints_mask'] = kp[..., -1]
            reg_nums['keypoints'] = _total(targets['keypoints_mask'])
            targets['keypoints_depth_mask'] = pick(tv["keypoints_depth_mask"], 3)
            pred_keypoints_3D = pois[:, k2c('corner_offset')].reshape(max(n_obj, 1), -1, 2)
            preds['keypoints'] = pred_keypoints_3D
            preds['keypoints_depths'] = enc.decode_depth_from_keypoints_batch(pred_keypoints_3D, pred_dimensions_3D,
                                                                              tv['calib'], batch_idxs)
            if self.corner_with_uncertainty:
                cu = pois[:, k2c('corner_uncertainty')]
                if self.uncertainty_range is not None:
                    cu = torch.clamp(cu, min=self.uncertainty_range[0], max=self.uncertainty_range[1])
                preds['corner_offset_uncertainty'] = cu

        if self.compute_extra_kpts_corner:
            ek = pick(tv["extra_kpts_2d"], tv["extra_kpts_2d"].shape[2], 3)
            targets['extra_kpts_2d'] = ek[..., :2]
            targets['extra_kpts_3d'] = pick(tv["extra_kpts_3d"], tv["extra_kpts_3d"].shape[2], 3)
            targets['find_pcl'] = targets['find_pcl'].bool() & obj_valid
            found = targets['find_pcl'].unsqueeze(-1).expand_as(ek[..., 2])
            targets['extra_kpts_2d_mask'] = (ek[..., 2] != 0) & found
            targets['extra_kpts_3d_mask'] = found
            reg_nums['extra_kpts_2d'] = _total(targets['extra_kpts_2d_mask'])
            reg_nums['extra_kpts_3d'] = _total(targets['extra_kpts_3d_mask'])

            pred_extra_kpts_2D = pois[:, k2c('extra_kpts_2d')].reshape(n_obj, -1, 2)
            pred_extra_kpts_3D = pois[:, k2c('extra_kpts_3d')].reshape(n_obj, -1, 3)
            preds['extra_kpts_2d'] = pred_extra_kpts_2D
            preds['extra_kpts_3d'] = pred_extra_kpts_3D
            pad = target_pad_size.unsqueeze(1).expand_as(pred_extra_kpts_2D)
            pred_extra_kpts_2D_img = enc.decode_kpts_2d_img(pred_extra_kpts_2D, target_center, target_offset_3D, pad)
            target_extra_kpts_2D_img = enc.decode_kpts_2d_img(targets['extra_kpts_2d'], target_center, target_offset_3D, pad)
            rot = target_rotys_3D.unsqueeze(-1)
            kmask = targets['extra_kpts_2d_mask']
            # (gt 2-D, gt 3-D): only its pair mask is consumed downstream (detector_loss.py:378, :188)
            with torch.no_grad():
                _, preds['pairs_kpt_depths_mask'] = enc.decode_pairs_kpts_depth(
                    target_extra_kpts_2D_img, targets['extra_kpts_3d'], rot, targets['Calib_P'], True, kmask,
                    targets['depth_3D'])
            # (pred 2-D, pred 3-D): the depth candidates that are trained (:381)
            preds['pairs_kpt_depths_all'], _ = enc.decode_pairs_kpts_depth(
                pred_extra_kpts_2D_img, pred_extra_kpts_3D, rot, targets['Calib_P'], True, kmask, targets['depth_3D'])

        if self.corner_loss_depth == 'edges':
            pred_corner_depth_3D = preds['pairs_kpt_depths_all'].mean(1)
        elif self.corner_loss_depth == 'direct':
            pred_corner_depth_3D = preds['depth_3D']
        else:
            raise ValueError("MODEL.HEAD.CORNER_LOSS_DEPTH must be 'edges' or 'direct'")

        pred_locations_3D = enc.decode_location_flatten(points, pred_offset_3D, pred_corner_depth_3D, tv['calib'],
                                                        tv['pad_size'], batch_idxs)
        pred_rotys_3D, _ = enc.decode_axes_orientation(pred_orientation_3D, pred_locations_3D)
        pred_corners_3D = enc.encode_box3d(pred_rotys_3D, pred_dimensions_3D, pred_locations_3D).float()
        pred_bboxes_3D = torch.cat((pred_locations_3D, pred_dimensions_3D, pred_rotys_3D[:, None]), dim=1)
        preds.update({'corners_3D': pred_corners_3D, 'rotys_3D': pred_rotys_3D, 'cat_3D': pred_bboxes_3D})
        if self.is_gen:
            self.generate_data(tv, pred_extra_kpts_2D_img, pred_extra_kpts_3D, reg_mask_gt, pred_rotys_3D,
                               target_locations_3D, pred_locations_3D)
        return targets, preds, reg_nums, weights

    # ------------------------------------------------------------------------------------------
    def __call__(self, predictions, targets):
        targets_heatmap, targets_variables = self.prepare_targets(targets)
        pred_heatmap = predictions['cls']
        reg_pois = predictions.get('reg_pois')
        if (self.use_graph and reg_pois is not None and not self.is_gen and pred_heatmap.is_cuda and torch.is_grad_enabled()
                and pred_heatmap.requires_grad and reg_pois.requires_grad and not torch.cuda.is_current_stream_capturing()):
            return self._call_graphed(pred_heatmap, reg_pois, targets_heatmap, targets_variables)
        loss_dict, names, packed = self._core(predictions, targets_heatmap, targets_variables)
        return loss_dict, LazyLogDict(names, packed, list(loss_dict))

    # ------------------------------------------------------------------------------------------
    # HIP-graph path.  With the object slots static and no host synchronisation, the ~500 small kernels of the loss (and the
    # ~500 of its backward) are captured once per input shape and replayed with two graph launches per step; eagerly the
    # host needs ~15 us per op and the GPU idles through most of this section (tools/prof_gaps.py).
    def _call_graphed(self, pred_heatmap, reg_pois, targets_heatmap, tv):
        tv = dict(tv)
        tv['calib'] = self.anno_encoder._calib_table(tv['calib'], pred_heatmap.device)
        names = sorted(k for k, v in tv.items() if torch.is_tensor(v) and k != 'ori_imgs')
        # The graphed callable copies every tensor argument into its static buffer before it replays: one launch per argument,
        # ~22 target fields.  They are packed per dtype first (one `cat` each: a single launch for any number of pieces) and
        # unpacked as views inside the graph: 3-4 arguments instead of 22.
        dtypes = sorted({tv[k].dtype for k in names}, key=str)
        groups = [[k for k in names if tv[k].dtype == dt] for dt in dtypes]
        packs = [torch.cat([tv[k].reshape(-1) for k in g]) if len(g) > 1 else tv[g[0]].reshape(-1) for g in groups]
        key = (tuple(pred_heatmap.shape), tuple(reg_pois.shape), tuple((k, tuple(tv[k].shape), tv[k].dtype) for k in names))
        entry = self._graphs.get(key)
        if entry is None:
            meta = {}
            layout = [[(k, tuple(tv[k].shape), tv[k].numel()) for k in g] for g in groups]

            def core_flat(cls, pois, hm, *pk):
                fields = {}
                for pack, items in zip(pk, layout):
                    o = 0
                    for k, shape, n in items:
                        fields[k] = pack[o:o + n].view(shape)
                        o += n
                loss_dict, log_names, packed = self._core({'cls': cls, 'reg': None, 'reg_pois': pois}, hm, fields)
                meta['loss_keys'], meta['log_names'] = list(loss_dict), log_names
                # three outputs instead of fifteen: the graphed backward copies one gradient per output into its static
                # buffers before it replays, and those launches sit in the one gap where the GPU waits for the host
                return loss_dict.stacked, loss_dict.total, packed
            sample = (pred_heatmap.detach().clone().requires_grad_(True), reg_pois.detach().clone().requires_grad_(True),
                      targets_heatmap.detach().clone()) + tuple(t.detach().clone() for t in packs)
            if len(self._graphs) >= 4:                       # a few input shapes at most (e.g. the last, smaller batch)
                self._graphs.pop(next(iter(self._graphs)))
            entry = (torch.cuda.make_graphed_callables(core_flat, sample), meta)
            self._graphs[key] = entry
        graphed, meta = entry
        stacked, total, packed = graphed(pred_heatmap, reg_pois, targets_heatmap, *packs)
        loss_dict = LossDict(zip(meta['loss_keys'], stacked.unbind(0)))      # views: summing them back-propagates as well
        loss_dict.total = total                  # the sum, formed inside the graph: `total.backward()` needs no eager adds
        return loss_dict, LazyLogDict(meta['log_names'], packed.clone(), list(loss_dict))

    def _column_weights(self, W, n, device):
        """(n,) constants `weight / batch_weight` of the columns; built on the host once per (values, device) -- inside a
        graph capture a host->device copy is not allowed, and the eager warm-up calls have filled the cache by then."""
        key = (tuple(float(W.get(i, 1.0)) for i in range(n)), str(device))
        if self._wcache[0] != key:
            self._wcache = (key, torch.tensor(key[0], dtype=torch.float32, device=device))
        return self._wcache[1]

    def _loss_matrix(self, spec, n, ratio_cols, device):
        """(M (len(spec), n) with M[l, c] = 1 for the columns c of loss l; ones (n,); ratio column indices) -- constants,
        built once per structure (see _column_weights for why not per call)."""
        key = (tuple((k, tuple(c)) for k, c in spec), n, tuple(ratio_cols), str(device))
        if self._mcache[0] != key:
            M = torch.zeros((len(spec), n), dtype=torch.float32)
            for r, (_, cols) in enumerate(spec):
                for c in cols:
                    M[r, c] = 1.0
            self._mcache = (key, (M.to(device), torch.ones(n, dtype=torch.float32, device=device),
                                  torch.tensor(list(ratio_cols), dtype=torch.long, device=device),
                                  (M.sum(dim=0) == 0).to(device)))          # columns no loss reads (logging-only metrics)
        return self._mcache[1]

    def _fused_rows(self, predictions, tv, batch_weight):
        """The per-object columns from one kernel (ops.loss_rows): (column sums, column indices as `_rows` returns them)."""
        pois = predictions['reg_pois']
        k2c, enc, lw = self.key2channel, self.anno_encoder, self.loss_weights
        K = tv['extra_kpts_2d'].shape[2]
        if self._rows_spec is None or self._rows_spec["K"] != K:
            heads = {'ch_box2d': '2d_dim', 'ch_offset': '3d_offset', 'ch_corner': 'corner_offset',
                     'ch_corner_unc': 'corner_uncertainty', 'ch_dims': '3d_dim', 'ch_ori_cls': 'ori_cls',
                     'ch_ori_off': 'ori_offset', 'ch_depth': 'depth', 'ch_depth_unc': 'depth_uncertainty',
                     'ch_kpts2d': 'extra_kpts_2d', 'ch_kpts3d': 'extra_kpts_3d'}
            spec = {f: k2c(name).start for f, name in heads.items()}
            spec.update(K=K, NP=1500, trunc_log=int(self.trunc_offset_loss_type != 'L1'),
                        depth_lo=float(enc.depth_range[0]), depth_hi=float(enc.depth_range[1]),
                        unc_lo=float(self.uncertainty_range[0]), unc_hi=float(self.uncertainty_range[1]),
                        depth_weight=float(lw['depth_loss']), dim_weight=[float(v) for v in self.dim_weight.reshape(-1).tolist()],
                        down_ratio=float(enc.down_ratio), kd_eps=float(enc.EPS))
            self._rows_spec = spec
        if enc.dim_mean.device != pois.device:
            enc.dim_mean = enc.dim_mean.to(pois.device)
        tv = dict(tv)
        tv['calib'] = enc._calib_table(tv['calib'], pois.device)
        S = ops.loss_rows(pois, self._rows_spec, enc.dim_mean.float(), tv)
        names = ('ov', 'giou', 'iou', 'm2', 'depth_real', 'depth', 'trunc', 'off', 'ori', 'dims', 'iou3d', 'corner', 'kp',
                 'l2d', 'm2d', 'l3d', 'm3d', 'valid_l', 'invalid_l', 'n_valid', 'n_invalid', 'mae', 'kd_log', 'kd_v', 'kd_i')
        return S, {n: i for i, n in enumerate(names)}

    def _core(self, predictions, targets_heatmap, targets_variables):
        pred_heatmap = predictions['cls']
        lw = self.loss_weights
        batch_weight = pred_heatmap.shape[0] * self.batch_weight_factor

        if self.heatmap_type != 'centernet':
            raise ValueError
        hm_loss, num_hm_pos = self.cls_loss_fnc(pred_heatmap, targets_heatmap)
        hm_loss = lw['hm_loss'] * hm_loss / batch_weight

        reg_pois = predictions.get('reg_pois')
        # the row kernel is HIP only: host tensors (MODEL.DEVICE=cpu, the oracle tools, the gloo tests) and other dtypes take
        # the op-by-op rows, like detector_predictor's `features.is_cuda` test for the head rows (advisor r3)
        if (self.fused_rows and reg_pois is not None and reg_pois.is_cuda and reg_pois.dtype == torch.float32
                and reg_pois.dim() == 3 and targets_variables['keypoints'].shape[2] == 10
                and targets_variables['orientations'].shape[-1] == 8):
            S_raw, ix = self._fused_rows(predictions, targets_variables, batch_weight)
        else:
            S_raw, ix = self._rows(predictions, targets_variables)
        return self._losses_from_columns(S_raw, ix, hm_loss, batch_weight)

    def _rows(self, predictions, targets_variables):
        """The per-object columns op by op: (column sums, {name: column index})."""
        pt, preds, reg_nums, weights = self.prepare_predictions(targets_variables, predictions)
        lw = self.loss_weights

        # Every term is a masked sum over the B*M object slots: the per-object columns are collected in `acc` and reduced
        # together (see _MaskedSums); `_losses_from_columns` applies weight / batch_weight per column.
        acc = _MaskedSums()
        ix = {}

        def term(name, vec, mask):
            ix[name] = acc.add(vec, mask)

        ov = pt['obj_valid'].float()                             # 1 for annotated objects, 0 for the padded slots
        one = torch.ones_like(ov)
        trunc = pt['trunc_mask_3D'].bool().float()
        term('ov', one, ov)

        # 2-D box: GIoU on the objects with a non-degenerate box
        m2 = pt['reg_2D_mask']
        safe_target = torch.where(m2.unsqueeze(1), pt['reg_2D'], torch.ones_like(pt['reg_2D']))
        giou_l, iou = self.iou_loss(preds['reg_2D'], safe_target)
        m2f = m2.float()
        term('giou', giou_l, m2f)
        term('iou', iou.detach(), m2f)
        term('m2', one, m2f)

        # direct depth (+ aleatoric uncertainty)
        if self.pred_direct_depth:
            depth_3D_loss = lw['depth_loss'] * self.depth_loss(preds['depth_3D'], pt['depth_3D'], reduction='none')
            term('depth_real', depth_3D_loss.detach(), ov)
            if self.depth_with_uncertainty:
                depth_3D_loss = depth_3D_loss * torch.exp(-preds['depth_uncertainty']) + preds['depth_uncertainty'] * lw['depth_loss']
            term('depth', depth_3D_loss, ov)

        # projected-centre offset; truncated objects use the log form
        off_l = self.reg_loss_fnc(preds['offset_3D'], pt['offset_3D'], reduction='none').sum(dim=1)
        if self.separate_trunc_offset:
            t_l = off_l if self.trunc_offset_loss_type == 'L1' else torch.log(1 + off_l)
            tv = trunc * ov
            term('trunc', t_l, tv)
            term('off', off_l, ov - tv)
        else:
            term('off', off_l, ov)

        if self.multibin:
            ori_rows = Real_MultiBin_loss(preds['orien_3D'], pt['orien_3D'], num_bin=self.orien_bin_size, per_row_only=True)
            term('ori', ori_rows, (pt['ori_mask'].bool() & pt['obj_valid']).float())
        else:
            raise NotImplementedError("only INPUT.ORIENTATION == 'multi-bin' is on the DGDE path")

        if self.dim_weight.device != preds['dims_3D'].device:      # one host->device copy, ever (a per-step copy from
            self.dim_weight = self.dim_weight.to(preds['dims_3D'])  # pageable memory synchronises the stream)
        dims_rows = (self.reg_loss_fnc(preds['dims_3D'], pt['dims_3D'], reduction='none') * self.dim_weight).sum(dim=1)
        term('dims', dims_rows, ov)

        with torch.no_grad():
            term('iou3d', get_iou_3d(preds['corners_3D'], pt['corners_3D']), ov)

        if self.compute_corner_loss:
            term('corner', self.reg_loss_fnc(preds['corners_3D'], pt['corners_3D'], reduction='none').sum(dim=(1, 2)), ov)
        if self.compute_keypoint_corner:
            kl = self.keypoint_loss_fnc(preds['keypoints'], pt['keypoints'], reduction='none').sum(dim=2) * pt['keypoints_mask']
            term('kp', kl.sum(dim=1), ov)
        if self.compute_extra_kpts_corner:
            ix_pairs, _mae = self.compute_pairs_kpts_loss(preds, pt, acc, one)
            ix.update(ix_pairs)
        if self.compute_keypoint_corner and self.compute_keypoint_depth_loss:
            kd = preds['keypoints_depths']
            km = pt['keypoints_depth_mask'].bool().float()
            tgt = pt['depth_3D'].unsqueeze(-1).expand_as(kd)
            v_l = self.reg_loss_fnc(kd, tgt, reduction='none')
            term('kd_log', (v_l.detach() * km).sum(dim=1), ov)
            i_l = v_l.detach()
            if self.corner_with_uncertainty:
                cu = preds['corner_offset_uncertainty']
                ecu = torch.exp(-cu)
                v_l = v_l * ecu + cu
                i_l = i_l * ecu
            term('kd_v', (v_l * km).sum(dim=1), ov)
            term('kd_i', (i_l * (1 - km)).sum(dim=1), ov)
        return acc.reduce(), ix

    def _losses_from_columns(self, S_raw, ix, hm_loss, batch_weight):
        """Column sums -> the 13 losses as ONE vector: `stacked = M (rc * [S, hm_loss])`, M a constant 0/1 matrix (which
        columns make up which loss), rc = 1 except for the four columns the reference divides by a mask count (no gradient
        through the counts).  Selecting the losses one by one out of S would cost a zero-filled vector + a copy + an
        accumulation per loss in the backward."""
        lw = self.loss_weights
        n_cols = S_raw.shape[0]
        # weight / batch_weight per column (the pair terms are means over their own mask counts: no batch weight)
        per_batch = {'giou': lw['bbox_loss'], 'depth_real': 1.0, 'depth': 1.0, 'trunc': lw.get('trunc_offset_loss', 1.0),
                     'off': lw['offset_loss'], 'ori': lw['orien_loss'], 'dims': lw['dims_loss'], 'corner': lw.get('corner_loss', 1.0),
                     'kp': lw.get('keypoint_loss', 1.0), 'kd_log': lw.get('keypoint_depth_loss', 1.0),
                     'kd_v': lw.get('keypoint_depth_loss', 1.0), 'kd_i': lw.get('keypoint_depth_loss', 1.0)}
        plain = {'l2d': lw.get('extra_kpts_2d_loss', 1.0), 'l3d': lw.get('extra_kpts_3d_loss', 1.0),
                 'valid_l': lw.get('pairs_kpts_depth_loss', 1.0), 'invalid_l': lw.get('pairs_kpts_depth_loss', 1.0)}
        W = {ix[k]: v / batch_weight for k, v in per_batch.items() if k in ix}
        W.update({ix[k]: v for k, v in plain.items() if k in ix})
        S = S_raw * self._column_weights(W, n_cols, S_raw.device)
        Sd = S.detach()
        i_ov = ix['ov']
        spec = [('hm_loss', [n_cols]), ('bbox_loss', [ix['giou']]), ('dims_loss', [ix['dims']]), ('orien_loss', [ix['ori']]),
                ('offset_loss', [ix['off']])]
        log_tensors = {'2D_IoU': Sd[ix['iou']] / torch.clamp(Sd[ix['m2']], min=1), '3D_IoU': Sd[ix['iou3d']] / Sd[i_ov]}   # means over the objects
        if self.separate_trunc_offset:
            spec.append(('trunc_offset_loss', [ix['trunc']]))
        if self.compute_corner_loss:
            spec.append(('corner_loss', [ix['corner']]))
        if self.pred_direct_depth:
            spec.append(('depth_loss', [ix['depth']]))
            log_tensors['depth_loss'] = Sd[ix['depth_real']]
        if self.compute_keypoint_corner:
            spec.append(('keypoint_loss', [ix['kp']]))
        ratio_cols, ratios = [], []
        if self.compute_extra_kpts_corner:
            scale = Sd[i_ov] / batch_weight                       # number of annotated objects / batch weight
            n_valid = Sd[ix['n_valid']]
            den = torch.clamp(torch.stack((Sd[ix['m2d']], Sd[ix['m3d']], n_valid, Sd[ix['n_invalid']])), min=1)
            ratio_cols, ratios = [ix['l2d'], ix['l3d'], ix['valid_l'], ix['invalid_l']], scale / den
            spec.append(('extra_kpts_2d_loss', [ix['l2d']]))
            spec.append(('extra_kpts_3d_loss', [ix['l3d']]))
            spec.append(('extra_kpts_depth_loss', [ix['valid_l'], ix['invalid_l']] if self.modify_invalid_keypoint_depths
                         else [ix['valid_l']]))
            log_tensors['extra_kpts_depth_loss'] = Sd[ix['valid_l']] / n_valid   # mean over the valid set (nan if empty, as the reference)
            all_mae = Sd[ix['mae']] / den[2]
        if self.compute_keypoint_corner and self.compute_keypoint_depth_loss:
            log_tensors['keypoint_depth_loss'] = Sd[ix['kd_log']]
            spec.append(('keypoint_depth_loss', [ix['kd_v'], ix['kd_i']] if self.modify_invalid_keypoint_depths else [ix['kd_v']]))
        M, rc_one, rc_idx, unused = self._loss_matrix(spec, n_cols + 1, ratio_cols, S_raw.device)
        cols = torch.cat((S, hm_loss.reshape(1).to(S.dtype)))
        if ratio_cols:
            cols = cols * rc_one.index_copy(0, rc_idx, ratios.to(S.dtype))
        # the metric-only columns (3-D IoU, 2-D IoU, MAE, ...) have zero rows in M, but 0 * NaN is NaN: a non-finite logging
        # value (3-D IoU of a degenerate box) must not turn every loss into NaN (advisor r2) -- they are zeroed out of the product
        stacked = torch.mv(M, cols.masked_fill(unused, 0.0))
        loss_dict = LossDict(zip([k for k, _ in spec], stacked.unbind(0)))
        loss_dict.stacked, loss_dict.total = stacked, stacked.sum()

        # ---- logging: one device->host copy for every scalar, made only when somebody reads the log dict (the reference
        # calls .item() twenty times inside the forward, detector_loss.py:589-630).  Reading forces the sync and the
        # reference's NaN/Inf check (:633-639: it drops into pdb; we raise).
        names = list(log_tensors) + [k for k in loss_dict if k not in log_tensors]
        vals = [log_tensors[k] if k in log_tensors else loss_dict[k].detach() for k in names]
        if self.compute_extra_kpts_corner:
            names.append('extra_all_MAE')
            vals.append(all_mae)
        packed = torch.stack([v.float().reshape(()) for v in vals])
        return loss_dict, names, packed


class LossDict(dict):
    """The 13-entry loss dict of the reference (detector_loss.py:582-623) plus `total`, the sum of its values when that sum
    was already formed on the device (graph path); `engine.trainer.train_step` back-propagates `total` when present."""
    total = None
    stacked = None         # the values as one vector, in key order (what `total` sums)


class LazyLogDict(dict):
    """dict of Python floats that is filled from ONE device tensor on first access, so that building it does not
    synchronise the host with the GPU in the middle of a train step."""

    def __init__(self, names, packed, loss_keys):
        super().__init__()
        self._names, self._packed, self._loss_keys = names, packed, loss_keys

    def _fill(self):
        if self._packed is not None:
            host = self._packed.tolist()
            self._packed = None
            super().update(zip(self._names, host))
            for k in self._loss_keys:
                v = super().__getitem__(k)
                if v != v or v in (float('inf'), float('-inf')):
                    raise FloatingPointError("non-finite loss %s: %s" % (k, dict(self)))

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        return len(self._names)

    def __contains__(self, k):
        return k in self._names

    def keys(self):
        self._fill()
        return super().keys()

    def values(self):
        self._fill()
        return super().values()

    def items(self):
        self._fill()
        return super().items()

    def get(self, k, default=None):
        self._fill()
        return super().get(k, default)

    def __repr__(self):
        self._fill()
        return super().__repr__()


def Real_MultiBin_loss(vector_ori, gt_ori, num_bin=4, row_mask=None, per_row_only=False):
    """Multi-bin orientation loss (detector_loss.py:644-666): per-bin 2-way cross entropy + L1 on the
    normalised (sin, cos) offsets of the bins that contain the angle.  `row_mask` replaces the
    boolean row selection of the caller (:473-477) with a masked sum.
    All bins at once on (N, num_bin, 2) views -- the reference's loop over the bins is ~22 launches per bin and as many in
    the backward; the sums are the same sums in a different order."""
    gt_ori = gt_ori.view(-1, gt_ori.shape[-1])
    n = gt_ori.shape[0]
    bins = gt_ori[:, :num_bin]
    logp = F.log_softmax(vector_ori[:, :2 * num_bin].reshape(n, num_bin, 2), dim=2)
    ce = -logp.gather(2, bins.long().unsqueeze(-1)).squeeze(-1)                       # (N, num_bin)
    off = F.normalize(vector_ori[:, 2 * num_bin:4 * num_bin].reshape(n, num_bin, 2), dim=2)
    ang = gt_ori[:, num_bin:2 * num_bin]
    target = torch.stack((torch.sin(ang), torch.cos(ang)), dim=2)
    reg = (off - target).abs().sum(dim=2)
    per_row = (ce * (1.0 / num_bin) + reg * (bins == 1).to(gt_ori.dtype)).sum(dim=1)
    if per_row_only:
        return per_row
    return per_row.sum() if row_mask is None else (per_row * row_mask.to(gt_ori.dtype)).sum()
