"""oracle/target_oracle.py (numpy restatement of the reference's target encoding) against the fixture produced by the
reference's own `KITTIDataset.__getitem__` (tests/golden/make_golden_targets.py): every ParamsList field of every image."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import target_oracle as TO  # noqa: E402

INT_FIELDS = ("cls_ids", "target_centers", "reg_mask", "trunc_mask", "find_pcl", "ori_mask", "pad_size", "edge_indices", "edge_len",
              "final_output_w", "final_output_h")


def load():
    return np.load(os.path.join(ROOT, "tests", "golden", "target_encoding.npz"))


def raw_inputs(g, i):
    return dict(image_size=g["in%d_image_size" % i], P=g["in%d_P" % i], trunc_occ=g["in%d_trunc_occ" % i], box2d=g["in%d_box2d" % i],
                hwl=g["in%d_hwl" % i], t=g["in%d_t" % i], ry=g["in%d_ry" % i], alpha=g["in%d_alpha" % i],
                find_pcl=g["in%d_find_pcl" % i], kpts3d=g["in%d_kpts3d" % i])


def compare(got, g, i, float_tol):
    names = [k[len("out%d_" % i):] for k in g.files if k.startswith("out%d_" % i) and not k.endswith("_size")]
    assert set(names) <= set(got), sorted(set(names) - set(got))
    for name in names:
        ref, val = g["out%d_%s" % (i, name)], np.asarray(got[name])
        assert val.shape == ref.shape, (name, val.shape, ref.shape)
        if name in INT_FIELDS:
            np.testing.assert_array_equal(val.astype(np.int64), ref.astype(np.int64), err_msg="image %d field %s" % (i, name))
        else:
            scale = max(np.abs(ref).max(), 1.0)
            assert np.abs(val.astype(np.float64) - ref.astype(np.float64)).max() <= float_tol * scale, (i, name)


def test_target_oracle_matches_reference_fixture():
    g = load()
    assert int(g["n_images"]) == 3
    kept = trunc = 0
    for i in range(int(g["n_images"])):
        got = TO.encode_image(**raw_inputs(g, i))
        compare(got, g, i, 1e-7)
        kept += int(got["reg_mask"].sum())
        trunc += int(got["trunc_mask"].sum())
    assert kept >= 12 and trunc >= 3            # the fixture does exercise the truncated-object branch
