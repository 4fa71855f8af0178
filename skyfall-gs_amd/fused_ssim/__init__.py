"""fused_ssim -- drop-in for rahul-goel/fused-ssim as the reference uses it (train.py:42,222,778):
fused_ssim(img1, img2, padding="same", train=True) -> scalar mean SSIM, differentiable w.r.t. img1.
Semantics == utils/loss_utils.py:23-63. Kernels: skyfall-gs_amd/csrc/ssim.hip via libsfgs.so."""
import torch

from sfgs import _lib as L

__all__ = ["fused_ssim"]


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, train):
        lib = L.load()
        B, Cc, H, W = (int(v) for v in img1.shape)
        dev = img1.device
        with_grad = bool(train and ctx.needs_input_grad[0])
        with torch.cuda.device(dev):
            stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            nbytes = lib.sfgs_ssim_scratch_bytes(B, Cc, H, W, int(with_grad))
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            mean = torch.empty((), dtype=torch.float32, device=dev)
            L.check(lib.sfgs_ssim_forward(L.ptr(img1), L.ptr(img2), B, Cc, H, W, None, L.ptr(mean), L.ptr(scratch),
                                          scratch.numel(), int(with_grad), stream))
        if with_grad:
            ctx.save_for_backward(img1, img2, scratch)
        return mean

    @staticmethod
    def backward(ctx, g_mean):
        lib = L.load()
        img1, img2, scratch = ctx.saved_tensors
        B, Cc, H, W = (int(v) for v in img1.shape)
        dev = img1.device
        with torch.cuda.device(dev):
            stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            g = g_mean.contiguous().float()
            out = torch.empty_like(img1)
            L.check(lib.sfgs_ssim_backward(L.ptr(img1), L.ptr(img2), B, Cc, H, W, L.ptr(scratch), L.ptr(g), L.ptr(out),
                                           stream))
        return out, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    if padding != "same":
        raise ValueError("only padding='same' is supported (the reference never passes anything else)")
    if img1.dim() != 4 or img1.shape != img2.shape:
        raise ValueError("fused_ssim expects two [B,C,H,W] tensors of equal shape")
    for name, t in (("img1", img1), ("img2", img2)):
        if t.dtype != torch.float32 or not t.is_cuda:
            raise ValueError(f"{name} must be a float32 GPU tensor")
    return _FusedSSIM.apply(img1.contiguous(), img2.contiguous().detach(), train)
