"""GaussianModel.add_densification_stats as one in-place HIP kernel without host syncs.

Reference: scene/gaussian_model.py:744-749 (called every training step below densify_until_iter, train.py:315).
`install(GaussianModel)` swaps the method on the reference's class; semantics and in-place behaviour are identical."""
import torch

from . import _lib as L

__all__ = ["add_densification_stats", "install", "uninstall"]


@torch.no_grad()
def add_densification_stats(model, viewspace_point_tensor, update_filter):
    grad = viewspace_point_tensor.grad
    N = int(grad.shape[0])
    bufs = [model.xyz_gradient_accum, model.xyz_gradient_accum_abs, getattr(model, "xyz_gradient_accum_abs_max", None),
            model.denom]
    for b in bufs:
        if b is not None and (b.dtype != torch.float32 or not b.is_contiguous() or b.numel() != N or not b.is_cuda):
            raise ValueError("densification buffers must be contiguous float32 GPU tensors of N elements")
    if grad.dtype != torch.float32 or tuple(grad.shape) != (N, 3) or not grad.is_cuda:
        raise ValueError("viewspace_point_tensor.grad must be a float32 GPU tensor [N,3]")
    f = update_filter
    if f.dtype == torch.bool:
        f = f.view(torch.uint8)
    if f.dtype != torch.uint8 or f.numel() != N:
        raise ValueError("update_filter must be a bool tensor with one entry per Gaussian")
    grad, f = grad.contiguous(), f.contiguous()
    dev = grad.device
    with torch.cuda.device(dev):
        stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(L.load().sfgs_densify_stats(N, L.ptr(grad), L.ptr(f), L.ptr(bufs[0]), L.ptr(bufs[1]), L.ptr(bufs[2]),
                                            L.ptr(bufs[3]), stream))
    for b in bufs:  # written through raw pointers: bump the autograd version counters like an in-place torch op
        if b is not None:
            torch.autograd.graph.increment_version(b)


_ORIG = {}


def install(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        return
    _ORIG[gaussian_model_cls] = gaussian_model_cls.add_densification_stats
    gaussian_model_cls.add_densification_stats = add_densification_stats


def uninstall(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        gaussian_model_cls.add_densification_stats = _ORIG.pop(gaussian_model_cls)
