"""One GMW optimisation step (GMW/main.py:447-466): edge depth candidates from the solver kernel, GMW forward,
correspondence loss on the transport plan, weighted-depth regression loss, backward, AdamW step."""
import torch


def correspondence_loss(P, C_gt=None):
    """mean over the batch of sum((1 - 2 C) P)  (GMW/lib/losses.py:22-26,115-119).  C_gt = None means the identity the
    train loop uses (main.py:456): sum(P) - 2 trace(P), without materialising a (B, 2628, 2628) identity."""
    # sums over the (2628, 2628) plans in two stages (rows, then the row sums): reducing both axes in one call makes ATen sum
    # 6.9 M contiguous values per output with a handful of workgroups -- 1.6 ms for the 221 MB of eight plans (0.14 TB/s)
    if C_gt is None:
        return (P.sum(dim=-1).sum(dim=-1) - 2.0 * P.diagonal(dim1=-2, dim2=-1).sum(-1)).mean()
    return ((1.0 - 2.0 * C_gt) * P).sum(dim=-1).sum(dim=-1).mean()


def compute_reg_loss(pre_depths, edge_weight, gt_depth, good_idx):
    """softmax of the matching weights over the 1500 best-conditioned edges, weighted mean of their depths, L1 to the
    ground truth (main.py:364-371)."""
    d = pre_depths.gather(-1, good_idx)
    w = edge_weight.gather(-1, good_idx).softmax(dim=-1)
    z = (d * w).sum(-1)
    return (z - gt_depth).abs().mean(), z


def gmw_losses(model, kpts_2d, kpts_3d, pred_rot, gt_location, cls_weight=1.0, reg_weight=0.0, compute_z=None):
    """(loss, cls_loss, reg_loss, pred_depth).  `compute_z` defaults to the HIP solver kernel (dcd_amd.ops.compute_z)."""
    if compute_z is None:
        from dcd_amd import ops
        compute_z = ops.compute_z
    pre_depths, good_idx = compute_z(kpts_2d, kpts_3d, pred_rot)
    reg_weights, edge_P = model(kpts_2d, kpts_3d, pred_rot)
    cls_loss = correspondence_loss(edge_P)
    reg_loss, pred_depth = compute_reg_loss(pre_depths, reg_weights, gt_location[:, -1], good_idx)
    return cls_weight * cls_loss + reg_weight * reg_loss, cls_loss, reg_loss, pred_depth


def gmw_train_step(model, optimizer, kpts_2d, kpts_3d, pred_rot, gt_location, cls_weight=1.0, reg_weight=0.0, compute_z=None):
    loss, cls_loss, reg_loss, pred_depth = gmw_losses(model, kpts_2d, kpts_3d, pred_rot, gt_location, cls_weight, reg_weight,
                                                      compute_z)
    optimizer.zero_grad()
    if not torch.isnan(loss).any():                 # main.py:464-465
        loss.backward()
    optimizer.step()
    return loss.detach(), cls_loss.detach(), reg_loss.detach(), pred_depth.detach()


def gmw_val_step(model, kpts_2d, kpts_3d, pred_rot, raw_location, dim, cls_weight=1.0, reg_weight=0.0, compute_z=None):
    """Validation step (GMW/main.py:524-548): the losses of the train step without gradients, and the detector's location
    rescaled along its viewing ray to the weighted edge depth -- about the object CENTRE: y is moved up by h/2 (KITTI
    locations are the bottom-face centre), scaled by pred_depth / raw_depth, and moved back.
    Returns (loss, cls_loss, reg_loss, pred_depth, pred_location)."""
    with torch.no_grad():
        loss, cls_loss, reg_loss, pred_depth = gmw_losses(model, kpts_2d, kpts_3d, pred_rot, raw_location, cls_weight, reg_weight,
                                                          compute_z)
        loc = raw_location.clone()
        scale = pred_depth / loc[:, 2]
        h = dim[:, 0]
        loc[:, 1] -= h / 2
        pred_location = scale.unsqueeze(-1) * loc
        pred_location[:, 1] += h / 2
    return loss, cls_loss, reg_loss, pred_depth, pred_location
