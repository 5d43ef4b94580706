"""DGDE -> GMW wire format.

  gen_data_train.json : Loss_Computation.gen_data dumped as is  (DGDE/engine/trainer.py:207-215):
      {'kpts_2d': [it][n][73][2] (K-normalised), 'kpts_3d': [it][n][73][3], 'pred_rot': [it][n], 'gt_location': [it][n][3],
       'pred_location': [it][n][3], 'weight_img': [], 'img_idx': [it][n] str}
  gen_data_infer.json : {img_id: [{'kpts_2d','kpts_3d','pred_rot','box','dim','pred_location','score','cat'}]}
      built from the PostProcessor rows [cls, alpha, x1,y1,x2,y2, h,w,l, x,y,z, roty, score] (DGDE/engine/inference.py:59-84).
Same keys, nesting and `json.dump(indent=4)` as the reference, so GMW's reader (GMW/utilities/dataset_utilities.py:11-56)
takes the files unchanged.
"""
import json
import os

import torch


def dump_gen_data_train(loss_evaluator, out_dir="gen_data"):
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "gen_data_train.json")
    with open(path, "w") as f:
        json.dump(loss_evaluator.gen_data, f, indent=4)
    return path


def infer_records(output, visualize_preds, cat="Car"):
    """One image's detections (N,14) + the PostProcessor's `gen_*` tensors -> list of GMW records."""
    n = output.shape[0]
    if n == 0:                                                  # nothing above the score threshold: no `gen_*` entries either
        return []
    k2, k3 = visualize_preds['gen_pred_extra_kpts_2d'], visualize_preds['gen_pred_extra_kpts_3d']
    nk = k2.shape[1]
    # one packed device buffer, one device-to-host copy for the whole image
    f32 = output.dtype
    packed = torch.cat((output.detach(), k2.detach().reshape(n, nk * 2).to(f32), k3.detach().reshape(n, nk * 3).to(f32)), dim=1).cpu().numpy()
    out, k2h, k3h = packed[:, :14], packed[:, 14:14 + nk * 2].reshape(n, nk, 2), packed[:, 14 + nk * 2:].reshape(n, nk, 3)
    # three conversions for the whole image, then list slices (a `.tolist()` per field and detection was 350 calls, 3 ms per image)
    k2l, k3l, rows = k2h.tolist(), k3h.tolist(), out.tolist()
    return [{'kpts_2d': k2l[i], 'kpts_3d': k3l[i], 'pred_rot': rows[i][12:13], 'box': rows[i][2:6], 'dim': rows[i][6:9],
             'pred_location': rows[i][9:12], 'score': rows[i][13:14], 'cat': cat} for i in range(n)]


def infer_records_batch(output, visualize_preds, image_of, n_images, cat="Car", as_lists=False):
    """`PostProcessor.forward_batch`'s rows for a whole batch -> per-image lists of GMW records ([[records of image 0], ...]): ONE
    packed device buffer, ONE device-to-host copy for the batch (BASELINE config 4: 16 images x 50 detections).
    The records' values are numpy VIEWS of that host buffer (kpts_2d (73, 2), kpts_3d (73, 3), box (4,), ...): turning 800 x 73 x 5
    floats into nested Python lists costs more host time than the GPU needs for the whole batch (94 of 180 ms per 16 images,
    tools/probes/gen_phases.py), and the only consumer, `dump_gen_data_infer`, converts while it writes.  as_lists=True gives the
    reference's lists right away (DGDE/engine/inference.py:66-81)."""
    per_image = [[] for _ in range(n_images)]
    n = output.shape[0]
    if n == 0:
        return per_image
    k2, k3 = visualize_preds['gen_pred_extra_kpts_2d'], visualize_preds['gen_pred_extra_kpts_3d']
    nk = k2.shape[1]
    f32 = output.dtype
    packed = torch.cat((output.detach(), k2.detach().reshape(n, nk * 2).to(f32), k3.detach().reshape(n, nk * 3).to(f32),
                        image_of.detach().view(n, 1).to(f32)), dim=1).cpu().numpy()
    k2h = packed[:, 14:14 + nk * 2].reshape(n, nk, 2)
    k3h = packed[:, 14 + nk * 2:14 + nk * 5].reshape(n, nk, 3)
    rows = packed[:, :14]
    owner = packed[:, -1].astype(int).tolist()
    if as_lists:
        k2h, k3h, rows = k2h.tolist(), k3h.tolist(), rows.tolist()
    for i in range(n):
        r = rows[i]
        per_image[owner[i]].append({'kpts_2d': k2h[i], 'kpts_3d': k3h[i], 'pred_rot': r[12:13], 'box': r[2:6],
                                    'dim': r[6:9], 'pred_location': r[9:12], 'score': r[13:14], 'cat': cat})
    return per_image


def _json_default(o):
    """numpy-backed record values (infer_records_batch) -> the lists the wire format holds."""
    import numpy as np
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, np.generic):
        return o.item()
    raise TypeError("not JSON serialisable: %r" % type(o))


def dump_gen_data_infer(infer_data, out_dir="gen_data"):
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "gen_data_infer.json")
    with open(path, "w") as f:
        json.dump(infer_data, f, indent=4, default=_json_default)
    return path
