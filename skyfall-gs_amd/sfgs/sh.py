"""Fused `eval_sh` (utils/sh_utils.py:57-112) with autograd, SURVEY 8f row 1.

render() evaluates SH in Python on two of its colour paths: the appearance path (`colors_toned = eval_sh(
pc.active_sh_degree, colors_toned, dir_pp_normalized)`, gaussian_renderer/__init__.py:115) every training step, and
`pipe.convert_SHs_python` (:124). `install(gaussian_renderer)` rebinds the module-level name `eval_sh` that render()
looks up, so the reference's source stays untouched; semantics are the reference function's (channel-major
sh[..., 3, K], only the first (deg+1)^2 coefficients used, no normalisation / offset / clamp)."""
import torch

from . import _handles

from . import _lib as L

__all__ = ["eval_sh", "eval_sh_deferred", "DeferredColor", "materialise", "install", "uninstall"]


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


# ---- deferred result: eval_sh + 0.5 + clamp_min folded into the rasterizer ---------------------------------------------
_METADATA = frozenset(("dim", "ndimension", "numel", "nelement", "size", "__len__", "is_contiguous", "element_size",
                       "is_floating_point", "is_complex", "stride", "storage_offset"))
FOLD_MAX_DEGREE = 4          # the rasterizer's in-kernel SH goes as far as eval_sh does (round 4; the upstream one stops at 3)
FOLD_COEFFS = (1, 4, 9, 16, 25)


def _is_scalar(x, value=None):
    if isinstance(x, bool) or not isinstance(x, (int, float)):
        return False
    return value is None or float(x) == value


class DeferredColor(torch.Tensor):
    """What the patched `eval_sh` returns with `install(..., fold=True)`: a storage-less [N,3] float32 handle standing for
    `eval_sh(deg, sh, dirs)`. render() does exactly two things with that value before it hands it to the rasterizer as
    `colors_precomp` (gaussian_renderer/__init__.py:116-117 and :124-125):

        colors = torch.clamp_min(eval_sh(...) + 0.5, 0.0)

    Both statements are RECORDED on the handle (`+ c` with a Python scalar, then `clamp_min(c)`), and GaussianRasterizer,
    when it receives a handle that recorded exactly `+ 0.5` and `clamp_min(0.0)`, evaluates the whole expression inside
    its preprocess kernels from the channel-major coefficients and the directions (SfgsGaussians.sh_dirs): no eval_sh
    launch, no N x 3 intermediates, no autograd nodes for the three steps. ANY other use of the handle -- other
    arithmetic, indexing, printing, a different constant, an unusual coefficient count -- materialises the real tensor with the ordinary
    fused eval_sh (plus the recorded steps as torch operations, with their autograd graph) and proceeds on it."""

    @staticmethod
    def __new__(cls, deg, sh, dirs, offset=0.0, clamp=None):
        t = torch.Tensor._make_wrapper_subclass(
            cls, (sh.shape[0], 3), dtype=torch.float32, device=sh.device,
            requires_grad=torch.is_grad_enabled() and (sh.requires_grad or dirs.requires_grad))
        t._sfgs_expr = (int(deg), sh, dirs, float(offset), clamp)
        t._sfgs_real = None
        return t

    def materialise(self):
        if self._sfgs_real is None:
            deg, sh, dirs, offset, clamp = self._sfgs_expr
            from . import viewdirs
            v = _EvalSH.apply(deg, sh.contiguous(), viewdirs.materialise(dirs).contiguous())
            if offset != 0.0:
                v = v + offset
            if clamp is not None:
                v = torch.clamp_min(v, clamp)
            self._sfgs_real = v
        return self._sfgs_real

    def folded_inputs(self):
        """(deg, coefficients, dirs[N,3], channel_major) when the handle stands for render()'s exact expression and the
        rasterizer can evaluate it (degree <= 4; 1 / 4 / 9 / 16 / 25 stored coefficients); None otherwise. `coefficients` is
        the [N,3,K] tensor itself when it is contiguous (channel_major = True: the appearance path's `.contiguous()`
        result) or, when it is the transposed view of a contiguous [N,K,3] tensor (convert_SHs_python:
        `pc.get_features.transpose(1, 2).view(-1, 3, K)`), that [N,K,3] tensor (channel_major = False) -- no copy
        either way; any other striding is made contiguous. With sfgs.features installed the convert_SHs_python view is a
        DeferredFeatures handle and `coefficients` the PAIR (features_dc [N,1,3], features_rest [N,K-1,3]). `dirs` is
        what eval_sh was given: a tensor, or the sfgs.viewdirs handle standing for render()'s normalised `dir_pp` (the
        rasterizer asks it for the centres: viewdirs.centers_of)."""
        deg, sh, dirs, offset, clamp = self._sfgs_expr
        if self._sfgs_real is not None or offset != 0.5 or clamp != 0.0:
            return None
        if deg > FOLD_MAX_DEGREE or sh.shape[2] not in FOLD_COEFFS:
            return None
        from . import features
        if isinstance(sh, features.DeferredFeatures):
            # convert_SHs_python with sfgs.features installed: `pc.get_features.transpose(1, 2).view(-1, 3, K)` is still a
            # handle on the model's two parameters -- the rasterizer reads them as they are (SfgsGaussians.shs_rest)
            parts = features.split_parts(sh)
            if parts is not None and parts[2]:
                return deg, (parts[0], parts[1]), dirs, False
            sh = sh.materialise()
        if sh.is_contiguous():
            return deg, sh, dirs, True
        t = sh.transpose(1, 2)
        if t.is_contiguous():
            return deg, t, dirs, False
        return deg, sh.contiguous(), dirs, True

    @classmethod
    def _unwrap(cls, x):
        if isinstance(x, DeferredColor):
            return x.materialise()
        if isinstance(x, (list, tuple)):
            return type(x)(cls._unwrap(y) for y in x)
        if isinstance(x, dict):
            return {k: cls._unwrap(v) for k, v in x.items()}
        return x

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name == "__get__" and not _handles.answered_by_wrapper(func):
            # .grad, .grad_fn, ._version, .data ...: properties of the tensor the handle stands for, read from it
            return getattr(args[0].materialise(), _handles.property_name(func))
        if name == "__get__" or name in _METADATA:   # shape, dtype, device, ...: answered by the wrapper's metadata
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if func is torch.Tensor.float and len(args) == 1 and not kwargs:
            return args[0]
        h = args[0] if args and isinstance(args[0], DeferredColor) else None
        if h is not None and h._sfgs_real is None:
            deg, sh, dirs, offset, clamp = h._sfgs_expr
            # `h + c` (Tensor.__add__ / Tensor.add / torch.add with a Python scalar, no alpha), before any clamp
            if name in ("add", "__add__") and len(args) == 2 and not kwargs and _is_scalar(args[1]) and clamp is None:
                return DeferredColor(deg, sh, dirs, offset + float(args[1]), None)
            # torch.clamp_min(h, c) / h.clamp_min(c)
            if name == "clamp_min" and clamp is None and not kwargs.get("out"):
                c = args[1] if len(args) == 2 else kwargs.get("min")
                if _is_scalar(c) and len(args) + len(kwargs) == 2:
                    return DeferredColor(deg, sh, dirs, offset, float(c))
        return func(*cls._unwrap(args), **cls._unwrap(kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):   # backstop: nothing should get here unmaterialised
        return func(*cls._unwrap(args), **cls._unwrap(kwargs or {}))


def _checked(deg, sh, dirs):
    """eval_sh's argument checks (the reference's two assertions, shapes, float32 GPU tensors on one device)."""
    if not (0 <= deg <= 4):
        raise AssertionError
    if sh.shape[-1] < (deg + 1) ** 2:
        raise AssertionError
    if sh.shape[-2] != 3 or dirs.shape[-1] != 3 or sh.shape[:-2] != dirs.shape[:-1]:
        raise ValueError(f"eval_sh expects sh [..., 3, K] and dirs [..., 3]; got {tuple(sh.shape)}, {tuple(dirs.shape)}")
    if not sh.is_cuda or sh.dtype != torch.float32 or dirs.dtype != torch.float32 or dirs.device != sh.device:
        raise ValueError("eval_sh: float32 GPU tensors on one device required (there is no CPU fallback)")


def eval_sh_deferred(deg, sh, dirs):
    """`eval_sh` with the same contract and the same argument checks, returning a DeferredColor handle."""
    _checked(deg, sh, dirs)
    if sh.dim() != 3:     # render() always passes [N,3,K]; other shapes take the ordinary route
        return eval_sh(deg, sh, dirs)
    return DeferredColor(int(deg), sh, dirs.contiguous())


def materialise(t):
    return t.materialise() if isinstance(t, DeferredColor) else t


_ORIG = {}


def install(renderer_module, fold=None):
    """Rebind `eval_sh` in the module that defines render() (gaussian_renderer/__init__.py:17 imports it by name).
    fold=True (default; SFGS_SH_FOLD=0 in the environment turns it off): the function returns a DeferredColor handle and
    render()'s `clamp_min(eval_sh(...) + 0.5, 0.0)` is evaluated inside the rasterizer's preprocess kernels; fold=False:
    one fused eval_sh launch returning an ordinary tensor."""
    if fold is None:
        import os
        fold = os.environ.get("SFGS_SH_FOLD", "1") != "0"
    if renderer_module not in _ORIG:
        _ORIG[renderer_module] = renderer_module.eval_sh
    renderer_module.eval_sh = eval_sh_deferred if fold else eval_sh


def uninstall(renderer_module):
    if renderer_module in _ORIG:
        renderer_module.eval_sh = _ORIG.pop(renderer_module)
