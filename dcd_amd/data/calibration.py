"""Camera calibration holder with the attributes the hot path reads from the reference's `Calibration`
(DGDE/data/datasets/kitti_utils.py:186-248: P, c_u, c_v, f_u, f_v, b_x, b_y) and its image->camera
back-projection (`project_image_to_rect`, kitti_utils.py:399-418).  Built from a 3x4 matrix instead of a
KITTI calib file (inputs are synthetic here; file parsing is data-pipeline code, out of scope)."""
import numpy as np
import torch

KITTI_P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728],
                     [0.0, 721.5377, 172.854, 0.2163791],
                     [0.0, 0.0, 1.0, 0.002745884]], dtype=np.float64)


class Calibration:
    def __init__(self, P=KITTI_P2):
        self.P = np.asarray(P, dtype=np.float64).reshape(3, 4)  # float64 like read_calib_file (kitti_utils.py)
        self.c_u = self.P[0, 2]
        self.c_v = self.P[1, 2]
        self.f_u = self.P[0, 0]
        self.f_v = self.P[1, 1]
        self.b_x = self.P[0, 3] / (-self.f_u)
        self.b_y = self.P[1, 3] / (-self.f_v)

    def project_image_to_rect(self, uv_depth):
        """(n,3) [u, v, depth] -> (n,3) camera-frame points; numpy or torch."""
        x = ((uv_depth[:, 0] - self.c_u) * uv_depth[:, 2]) / self.f_u + self.b_x
        y = ((uv_depth[:, 1] - self.c_v) * uv_depth[:, 2]) / self.f_v + self.b_y
        if isinstance(uv_depth, np.ndarray):
            return np.stack([x, y, uv_depth[:, 2]], axis=1)
        return torch.stack([x, y, uv_depth[:, 2]], dim=1)

    def project_rect_to_image(self, pts_3d_rect):
        """(n,3) camera-frame points -> (n,2) pixels and (n,) depth (kitti_utils.py project_rect_to_image)."""
        if isinstance(pts_3d_rect, np.ndarray):
            hom = np.concatenate([pts_3d_rect, np.ones((pts_3d_rect.shape[0], 1), pts_3d_rect.dtype)], axis=1)
            proj = hom @ self.P.T.astype(pts_3d_rect.dtype)
        else:
            hom = torch.cat([pts_3d_rect, pts_3d_rect.new_ones(pts_3d_rect.shape[0], 1)], dim=1)
            proj = hom @ torch.as_tensor(self.P, dtype=pts_3d_rect.dtype, device=pts_3d_rect.device).t()
        return proj[:, :2] / proj[:, 2:3], proj[:, 2]
