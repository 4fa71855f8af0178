"""Oracle of compute_3D_filter (TEST INFRASTRUCTURE): scene/gaussian_model.py:255-308 restated in numpy float64.
Pinned against a golden vector produced by the REAL GaussianModel.compute_3D_filter (tests/golden/make_golden.py)."""
import numpy as np


def compute_3D_filter(xyz, cameras):
    """cameras: objects/dicts with R, T, cx, cy, image_width, image_height, focal_x, focal_y."""
    get = lambda c, k: c[k] if isinstance(c, dict) else getattr(c, k)
    xyz = np.asarray(xyz, np.float64)
    distance = np.full(xyz.shape[0], 1e8)
    valid_points = np.zeros(xyz.shape[0], bool)
    focal_length = 0.0
    for cam in cameras:
        R = np.asarray(get(cam, "R"), np.float64)
        T = np.asarray(get(cam, "T"), np.float64)
        W, H = get(cam, "image_width"), get(cam, "image_height")
        xyz_cam = xyz @ R + T[None, :]
        valid_depth = xyz_cam[:, 2] > 0.2
        x, y, z = xyz_cam[:, 0], xyz_cam[:, 1], np.maximum(xyz_cam[:, 2], 0.001)
        cx_ori = get(cam, "cx") / 2 * W + W / 2
        cy_ori = get(cam, "cy") / 2 * H + H / 2
        x = x / z * get(cam, "focal_x") + cx_ori
        y = y / z * get(cam, "focal_y") + cy_ori
        in_screen = (x >= -0.15 * W) & (x <= W * 1.15) & (y >= -0.15 * H) & (y <= 1.15 * H)
        valid = valid_depth & in_screen
        distance[valid] = np.minimum(distance[valid], z[valid])
        valid_points |= valid
        focal_length = max(focal_length, get(cam, "focal_x"))
    distance[~valid_points] = distance[valid_points].max()
    return (distance / focal_length * (0.2 ** 0.5))[:, None]
