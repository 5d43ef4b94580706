"""Fixture for SURVEY section 8(f)-4 (target encoding): runs the REFERENCE's own `KITTIDataset.__getitem__`
(DGDE/data/datasets/kitti.py:283-606, heat-map helpers DGDE/model/heatmap_coder.py:37-124) on a tiny KITTI-format data set that
this script fabricates in a temporary directory, and stores its inputs (the raw label / calibration / key-point annotation
values, as arrays) and outputs (every ParamsList field) in tests/golden/target_encoding.npz.

Build container only (needs /root/reference).  What has to be supplied for the reference to run here, none of which touches
the arithmetic: the stand-in modules of make_golden.install_stubs (cv2, ...), `np.bool` / `np.int` / `np.bool8` aliases that
numpy 2 removed (kitti.py:367,386,505), and the files themselves -- blank PNG images (only their size is read on this path),
label_2 / calib text files and kpts_ann/kpts_ann_train.json written from the seeded values below.  Augmentation is off
(`augment=False`), as for a deterministic fixture.

The scenes are built to reach every branch of the encoder: objects inside the image, objects whose projected centre falls
outside (truncated: `approx_proj_center`, 1-D edge heat map), a box that fails FILTER_ANNOS, an object behind the camera,
classes that are filtered out ('Van', 'DontCare'), an object without point-cloud key points (find_pcl = 0), key points
outside the image / behind the camera, two image sizes (different paddings)."""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]])
P2_B = np.array([[707.0493, 0.0, 604.0814, 45.75831], [0.0, 707.0493, 180.5066, -0.3454157], [0.0, 0.0, 1.0, 0.004981016]])
N_EXTRA = 63


def scenes():
    """[(image (w,h), P, [object dict])]; every value is what the label / json files will hold (rounded like KITTI's text)."""
    rng = np.random.RandomState(7)
    out = []

    def car(x, z, ry, trunc=0.0, occ=0, typ="Car", dims=None, box=None, pcl=True, y=None):
        h, w, l = dims if dims is not None else (round(rng.normal(1.5, 0.1), 2), round(rng.normal(1.6, 0.1), 2), round(rng.normal(3.9, 0.3), 2))
        k3 = np.round(rng.uniform(-0.5, 0.5, (N_EXTRA, 3)) * np.array([l, h, w]) + np.array([0, h / 2, 0]), 4)
        k2 = np.round(rng.uniform(0, 1, (N_EXTRA, 3)), 3)
        return dict(type=typ, trunc=trunc, occ=occ, h=h, w=w, l=l, t=(round(x, 2), round(1.65 if y is None else y, 2), round(z, 2)),
                    ry=round(ry, 2), box=box, pcl=pcl, k3=k3, k2=k2)
    out.append(((1242, 375), P2, [car(-3.0, 15.0, 1.2), car(2.5, 30.0, -0.4), car(8.0, 22.0, 2.9, occ=1), car(-1.0, 45.0, 0.1, pcl=False),
                                  car(1.0, 9.0, -1.7, typ="Van"), car(0.0, 20.0, 0.0, typ="DontCare"), car(-6.5, 9.5, 0.3, trunc=0.4)]))
    out.append(((1224, 370), P2_B, [car(5.6, 6.0, 1.5, trunc=0.6), car(-6.0, 6.5, -1.3, trunc=0.5), car(0.5, 12.0, -3.0),
                                    car(3.0, -5.0, 0.2), car(-9.0, 14.0, 0.7, trunc=0.95, box=(0.0, 170.0, 14.0, 200.0)),
                                    car(12.5, 18.0, -2.2, trunc=0.3)]))
    out.append(((1242, 375), P2, [car(0.3, 2.6, 0.05, trunc=0.7), car(-17.5, 20.0, 1.0, trunc=0.95, box=(0.0, 160.0, 12.0, 190.0)), car(5.0, 60.0, 0.8),
                                  car(-2.0, 11.0, -0.9, y=1.2)]))
    return out


def box_from_projection(o, P, img):
    """Annotated 2-D box: KITTI's is the clipped projection of the 3-D box (two decimals), unless given."""
    if o["box"] is not None:
        return o["box"]
    h, w, l = o["h"], o["w"], o["l"]
    xc = np.array([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2])
    yc = np.array([0, 0, 0, 0, -h, -h, -h, -h])
    zc = np.array([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2])
    R = np.array([[np.cos(o["ry"]), 0, np.sin(o["ry"])], [0, 1, 0], [-np.sin(o["ry"]), 0, np.cos(o["ry"])]])
    c = (R @ np.stack([xc, yc, zc])).T + np.array(o["t"])
    c = c[c[:, 2] > 0.1] if (c[:, 2] > 0.1).any() else c
    hom = np.concatenate([c, np.ones((c.shape[0], 1))], 1) @ P.T
    uv = hom[:, :2] / hom[:, 2:3]
    x1, y1 = np.clip(uv[:, 0].min(), 0, img[0] - 1), np.clip(uv[:, 1].min(), 0, img[1] - 1)
    x2, y2 = np.clip(uv[:, 0].max(), 0, img[0] - 1), np.clip(uv[:, 1].max(), 0, img[1] - 1)
    return tuple(round(float(v), 2) for v in (x1, y1, x2, y2))


def write_dataset(root, sc):
    from PIL import Image
    for d in ("image_2", "label_2", "calib", "ImageSets"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    os.makedirs(os.path.join(root, "kpts_ann"), exist_ok=True)
    ann, ids = {}, []
    for i, (img, P, objs) in enumerate(sc):
        name = "%06d" % i
        ids.append(name)
        Image.new("RGB", img).save(os.path.join(root, "image_2", name + ".png"))
        with open(os.path.join(root, "calib", name + ".txt"), "w") as f:
            flat = " ".join("%.12e" % v for v in P.reshape(-1))
            f.write("P0: %s\nP1: %s\nP2: %s\nP3: %s\n" % (flat, flat, flat, flat))
            f.write("R0_rect: 1 0 0 0 1 0 0 0 1\nTr_velo_to_cam: 1 0 0 0 0 1 0 0 0 0 1 0\nTr_imu_to_velo: 1 0 0 0 0 1 0 0 0 0 1 0\n")
        lines, recs = [], []
        for o in objs:
            o["box"] = box_from_projection(o, P, img)
            alpha = o["ry"] - np.arctan2(o["t"][0], o["t"][2])
            lines.append("%s %.2f %d %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f" % (
                o["type"], o["trunc"], o["occ"], alpha, *o["box"], o["h"], o["w"], o["l"], *o["t"], o["ry"]))
            recs.append({"dim": [o["h"], o["w"], o["l"]], "find_pcl": bool(o["pcl"]), "3dkeypoints": o["k3"].reshape(-1).tolist(),
                         "2dkeypoints": o["k2"].reshape(-1).tolist()})
        with open(os.path.join(root, "label_2", name + ".txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
        ann[str(i)] = recs
    with open(os.path.join(root, "ImageSets", "train.txt"), "w") as f:
        f.write("\n".join(ids) + "\n")
    with open(os.path.join(root, "kpts_ann", "kpts_ann_train.json"), "w") as f:
        json.dump(ann, f)


def main():
    mg.install_stubs()
    import types
    for name in ("matplotlib", "matplotlib.pyplot"):            # imported at module level by kitti.py, unused on this path
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    np.bool, np.int, np.bool8, np.float = bool, int, np.bool_, float        # aliases numpy 2 removed (kitti.py:367,386,505)
    sys.path.insert(0, mg.REF)
    sc = scenes()
    with tempfile.TemporaryDirectory() as tmp:
        write_dataset(tmp, sc)
        os.chdir(tmp)                                            # kitti.py opens 'kpts_ann/kpts_ann_train.json' relative to the cwd
        cfg = mg.ref_cfg()
        from data.datasets.kitti import KITTIDataset
        ds = KITTIDataset(cfg, tmp, is_train=True, transforms=None, augment=False)
        assert len(ds) == len(sc), (len(ds), len(sc))
        out = {"n_images": np.array(len(ds))}
        for i in range(len(ds)):
            img, target, idx = ds[i]
            assert idx == "%06d" % i
            objs = ds.filtrate_objects(ds.get_label_objects(i))             # the reference's own parse of the files = the encoder's input
            out["in%d_image_size" % i] = np.array(sc[i][0])
            out["in%d_P" % i] = np.asarray(target.get_field("calib").P, np.float64)
            out["in%d_n" % i] = np.array(len(objs))
            out["in%d_trunc_occ" % i] = np.array([[o.truncation, float(o.occlusion)] for o in objs], np.float64)
            out["in%d_box2d" % i] = np.stack([o.box2d for o in objs]).astype(np.float32)
            out["in%d_hwl" % i] = np.array([[o.h, o.w, o.l] for o in objs], np.float64)
            out["in%d_t" % i] = np.stack([o.t for o in objs]).astype(np.float32)
            out["in%d_ry" % i] = np.array([o.ry for o in objs], np.float64)
            out["in%d_alpha" % i] = np.array([o.alpha for o in objs], np.float64)
            out["in%d_find_pcl" % i] = np.array([o.find_pcl for o in objs], np.int32)
            out["in%d_kpts3d" % i] = np.stack([o.extra_kpts_3D for o in objs]).astype(np.float64)      # already shifted by -h/2 (kitti_utils.py:112)
            for name in target.fields():
                v = target.get_field(name)
                if name in ("calib", "ori_img", "img_idx"):
                    continue
                out["out%d_%s" % (i, name)] = np.asarray(v)
            out["out%d_size" % i] = np.array(target.size)
        os.chdir(HERE)
    path = os.path.join(HERE, "target_encoding.npz")
    np.savez_compressed(path, **out)
    print("wrote target_encoding.npz %.1f KB; objects kept per image:" % (os.path.getsize(path) / 1024),
          [int(out["out%d_reg_mask" % i].sum()) for i in range(len(sc))], "truncated:", [int(out["out%d_trunc_mask" % i].sum()) for i in range(len(sc))])


if __name__ == "__main__":
    main()
