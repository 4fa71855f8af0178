"""Fused `eval_sh` (utils/sh_utils.py:57-112) with autograd, SURVEY 8f row 1.

render() evaluates SH in Python on two of its colour paths: the appearance path (`colors_toned = eval_sh(
pc.active_sh_degree, colors_toned, dir_pp_normalized)`, gaussian_renderer/__init__.py:115) every training step, and
`pipe.convert_SHs_python` (:124). `install(gaussian_renderer)` rebinds the module-level name `eval_sh` that render()
looks up, so the reference's source stays untouched; semantics are the reference function's (channel-major
sh[..., 3, K], only the first (deg+1)^2 coefficients used, no normalisation / offset / clamp)."""
import torch

from . import _lib as L

__all__ = ["eval_sh", "install", "uninstall"]


def _stream(dev):
    return L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _EvalSH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, deg, sh, dirs):
        n, k = sh.shape[0], sh.shape[2]
        out = torch.empty(n, 3, dtype=torch.float32, device=sh.device)
        with torch.cuda.device(sh.device):
            L.check(L.load().sfgs_sh_eval_forward(n, deg, k, L.ptr(sh), L.ptr(dirs), L.ptr(out), _stream(sh.device)))
        ctx.deg = deg
        ctx.save_for_backward(sh, dirs)
        return out

    @staticmethod
    def backward(ctx, g_out):
        sh, dirs = ctx.saved_tensors
        n, k = sh.shape[0], sh.shape[2]
        g_out = g_out.contiguous()
        g_sh = torch.empty_like(sh)
        g_dirs = torch.empty_like(dirs) if ctx.needs_input_grad[2] else None
        with torch.cuda.device(sh.device):
            L.check(L.load().sfgs_sh_eval_backward(n, ctx.deg, k, L.ptr(sh), L.ptr(dirs), L.ptr(g_out), L.ptr(g_sh),
                                                   L.ptr(g_dirs), _stream(sh.device)))
        return None, g_sh, g_dirs


def eval_sh(deg, sh, dirs):
    """Same contract as the reference's eval_sh: sh [..., 3, K] (K >= (deg+1)^2), dirs [..., 3] -> [..., 3]."""
    if not (0 <= deg <= 4):
        raise AssertionError  # the reference asserts `deg <= 4 and deg >= 0`
    if sh.shape[-1] < (deg + 1) ** 2:
        raise AssertionError  # reference: `assert sh.shape[-1] >= coeff`
    if sh.shape[-2] != 3 or dirs.shape[-1] != 3 or sh.shape[:-2] != dirs.shape[:-1]:
        raise ValueError(f"eval_sh expects sh [..., 3, K] and dirs [..., 3]; got {tuple(sh.shape)}, {tuple(dirs.shape)}")
    if not sh.is_cuda or sh.dtype != torch.float32 or dirs.dtype != torch.float32 or dirs.device != sh.device:
        raise ValueError("eval_sh: float32 GPU tensors on one device required (there is no CPU fallback)")
    lead = sh.shape[:-2]
    out = _EvalSH.apply(int(deg), sh.reshape(-1, 3, sh.shape[-1]).contiguous(), dirs.reshape(-1, 3).contiguous())
    return out.reshape(*lead, 3)


_ORIG = {}


def install(renderer_module):
    """Rebind `eval_sh` in the module that defines render() (gaussian_renderer/__init__.py:17 imports it by name)."""
    if renderer_module in _ORIG:
        return
    _ORIG[renderer_module] = renderer_module.eval_sh
    renderer_module.eval_sh = eval_sh


def uninstall(renderer_module):
    if renderer_module in _ORIG:
        renderer_module.eval_sh = _ORIG.pop(renderer_module)
