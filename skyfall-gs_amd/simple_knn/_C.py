"""simple_knn._C.distCUDA2(points[N,3] float32 GPU) -> [N] float32: mean squared distance to the three
nearest other points (reference call site scene/gaussian_model.py:324). Kernels: csrc/knn.hip (exact; brute force for
small clouds, Z-curve counting sort + box-pruned search above 32 768 points)."""
import torch

from sfgs import _lib as L

__all__ = ["distCUDA2"]


def distCUDA2(points):
    if points.dim() != 2 or points.shape[1] != 3:
        raise ValueError("distCUDA2 expects an [N,3] tensor")
    if points.dtype != torch.float32 or not points.is_cuda:
        raise ValueError("distCUDA2 expects a float32 GPU tensor")
    lib = L.load()
    pts = points.contiguous()
    N = int(pts.shape[0])
    dev = pts.device
    out = torch.empty(N, dtype=torch.float32, device=dev)
    nbytes = int(lib.sfgs_knn_scratch_bytes(N))     # 0 for small clouds (brute force); the spatial path sorts into scratch
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    with torch.cuda.device(dev):
        stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(lib.sfgs_knn_dist2(L.ptr(pts), N, L.ptr(out), L.ptr(scratch), nbytes, stream))
    return out
