"""Camera tensors in the reference's conventions (host logic, CPU tensors).

Restates scene/cameras.py:56-79 and utils/graphics_utils.py:38-126 of the reference: the
rasterizer receives `world_view_transform` = W2C^T and `full_proj_transform` = W2C^T @ P^T, both
row-major float32, so that p_view = [p,1] @ viewmatrix.  znear 0.01 / zfar 100 as
scene/cameras.py:56-57.  Checked against the reference's own functions by
tests/test_golden_helpers.py (fixtures in tests/golden/).
"""
import math

import numpy as np
import torch

ZNEAR, ZFAR = 0.01, 100.0


def world_to_view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """getWorld2View2 (utils/graphics_utils.py:37-101, numpy branch). R is the COLMAP
    camera-to-world rotation (its transpose goes into the matrix), t the world-to-camera translation."""
    R = np.asarray(R, np.float64)
    t = np.asarray(t, np.float64).reshape(3)
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, np.float64)) * scale
    return np.linalg.inv(C2W).astype(np.float32)


def projection(znear, zfar, fovx, fovy, cx=0.0, cy=0.0):
    """getProjectionMatrix (utils/graphics_utils.py:106-126). cx, cy are principal-point offsets in
    NDC units (scene/dataset_readers.py:553-554)."""
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    top, right = ty * znear, tx * znear
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - (-right))
    P[1, 1] = 2.0 * znear / (top - (-top))
    P[0, 2] = cx
    P[1, 2] = cy
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_frame(R, t, fovx, fovy, W, H, cx=0.0, cy=0.0, kernel_size=0.1, scale_modifier=1.0, sh_degree=0,
               bg=(0.0, 0.0, 0.0), subpix=None, depth_mode=0):
    """Everything GaussianRasterizationSettings needs (gaussian_renderer/__init__.py:40-55), CPU."""
    view = torch.tensor(world_to_view(R, t)).transpose(0, 1).contiguous()
    proj = projection(ZNEAR, ZFAR, fovx, fovy, cx, cy).transpose(0, 1)
    full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    campos = view.inverse()[3, :3].contiguous()
    return dict(H=int(H), W=int(W), tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                kernel_size=float(kernel_size), scale_modifier=float(scale_modifier), sh_degree=int(sh_degree),
                view=view, proj=full, campos=campos, bg=torch.tensor(bg, dtype=torch.float32), subpix=subpix,
                depth_mode=int(depth_mode))


def fovy_from_fovx(fovx, W, H):
    """Equal focal length in x and y (render_video.py:105-107)."""
    focal = W / (2.0 * math.tan(fovx / 2))
    return 2.0 * math.atan(H / (2.0 * focal))
