"""GaussianModel.compute_3D_filter as one fused HIP pass.

Reference: scene/gaussian_model.py:255-308, called from train.py at start-up and after every densification: a Python
loop over all training cameras with ~12 float64 torch kernels each. `compute_3D_filter(xyz, cameras)` returns the
same [N,1] float64 tensor; `install(GaussianModel)` swaps the method on the reference's class (no source edit)."""
import numpy as np
import torch

from . import _lib as L

__all__ = ["compute_3D_filter", "install", "uninstall"]


def pack_cameras(cameras):
    """[C,18] float64: R (as the reference uses it: xyz @ R), T, focal_x, focal_y, cx_ori, cy_ori, W, H
    (scene/gaussian_model.py:268-286)."""
    rows, focal = [], 0.0
    for cam in cameras:
        W, H = float(cam.image_width), float(cam.image_height)
        cx_ori = cam.cx / 2 * cam.image_width + cam.image_width / 2
        cy_ori = cam.cy / 2 * cam.image_height + cam.image_height / 2
        rows.append(np.concatenate([np.asarray(cam.R, np.float64).reshape(9), np.asarray(cam.T, np.float64).reshape(3),
                                    [float(cam.focal_x), float(cam.focal_y), float(cx_ori), float(cy_ori), W, H]]))
        if focal < cam.focal_x:
            focal = float(cam.focal_x)
    return np.stack(rows) if rows else np.zeros((0, 18)), focal


@torch.no_grad()
def compute_3D_filter(xyz, cameras):
    if xyz.dtype != torch.float32 or not xyz.is_cuda or xyz.dim() != 2 or xyz.shape[1] != 3:
        raise ValueError("xyz must be a float32 GPU tensor of shape [N,3]")
    lib = L.load()
    cams, focal = pack_cameras(cameras)
    if len(cams) == 0:
        raise ValueError("compute_3D_filter needs at least one camera")
    dev = xyz.device
    N = int(xyz.shape[0])
    xyz = xyz.detach().contiguous()
    cams_d = torch.tensor(cams, dtype=torch.float64, device=dev)
    out = torch.empty(N, 1, dtype=torch.float64, device=dev)
    scratch = torch.empty(max(lib.sfgs_filter3d_scratch_bytes(N), 1), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sfgs_filter3d(L.ptr(xyz), N, L.ptr(cams_d), len(cams), focal, L.ptr(out), L.ptr(scratch),
                                  scratch.numel(), stream))
    return out


_ORIG = {}


def install(gaussian_model_cls):
    """Replace GaussianModel.compute_3D_filter (same signature, same result) by the fused pass."""
    if gaussian_model_cls in _ORIG:
        return
    _ORIG[gaussian_model_cls] = gaussian_model_cls.compute_3D_filter

    def compute(self, cameras):
        self.filter_3D = compute_3D_filter(self.get_xyz, cameras)
    gaussian_model_cls.compute_3D_filter = compute


def uninstall(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        gaussian_model_cls.compute_3D_filter = _ORIG.pop(gaussian_model_cls)
