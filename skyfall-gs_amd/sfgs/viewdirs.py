"""render()'s view directions without their torch kernels (the per-Gaussian pre-pass of the colour paths).

Both Python colour paths of render() build eval_sh's `dirs` argument with the same two statements
(gaussian_renderer/__init__.py:114-115 appearance path, :122-123 convert_SHs_python):

    dir_pp = (pc.get_xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1))
    dir_pp_normalized = dir_pp/dir_pp.norm(dim=1, keepdim=True)

-- a subtraction, a norm (three launches), a broadcast division, and eight more launches when autograd walks back through
them into `xyz.grad`: ~0.13 ms of a 1.9 ms training iteration at 2 M Gaussians.

`install(GaussianModel)` (called by `sfgs.prepass.install`) makes `get_xyz` return a storage-less handle on `_xyz`. The
handle RECORDS exactly those statements -- `handle - tensor[N,3]`, `.norm(dim=1, keepdim=True)` of that, their quotient --
and the result, handed to the folding `eval_sh` (sfgs.sh) and on to the rasterizer inside the colour handle, makes the
rasterizer compute normalize(means3D - centres) inside preprocess / preprocess_bwd (SfgsGaussians.sh_centers), with the
direction's gradient added to means3D's there. Anything else done with any of the handles -- other arithmetic, indexing,
another norm, printing -- runs the recorded statements as ordinary torch operations (with their autograd graph) and
proceeds on the result; the handle on `_xyz` itself simply stands for the parameter."""
import torch

from . import _handles

__all__ = ["LazyDirs", "centers_of", "materialise", "install", "uninstall"]

_METADATA = frozenset(("dim", "ndimension", "numel", "nelement", "size", "__len__", "is_contiguous", "element_size",
                       "is_floating_point", "is_complex", "stride", "storage_offset"))
XYZ, DIRPP, NORM, DIRS = "xyz", "dir_pp", "norm", "dirs"


class LazyDirs(torch.Tensor):
    """kind XYZ: the model's `_xyz` (src = the parameter); DIRPP: `xyz - centres` (src = (xyz handle, centres)); NORM:
    `dir_pp.norm(dim=1, keepdim=True)` (src = the DIRPP handle); DIRS: `dir_pp / norm` (src = (DIRPP handle, NORM handle))."""

    @staticmethod
    def __new__(cls, kind, src, shape, like):
        # (the handle on the parameter mirrors the parameter, a leaf: requires_grad whatever the grad mode; the others are
        # results of operations)
        t = torch.Tensor._make_wrapper_subclass(
            cls, shape, dtype=like.dtype, device=like.device,
            requires_grad=like.requires_grad and (kind == XYZ or torch.is_grad_enabled()))
        t._sfgs_kind, t._sfgs_src, t._sfgs_real = kind, src, None
        return t

    def materialise(self):
        if self._sfgs_real is None:
            k, s = self._sfgs_kind, self._sfgs_src
            if k == XYZ:
                v = s
            elif k == DIRPP:
                v = s[0].materialise() - s[1]
            elif k == NORM:
                v = s.materialise().norm(dim=1, keepdim=True)
            else:
                v = s[0].materialise() / s[1].materialise()
            self._sfgs_real = v
        return self._sfgs_real

    @classmethod
    def _unwrap(cls, x):
        if isinstance(x, LazyDirs):
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
            h0 = args[0]   # (the handle on `_xyz` reads its parameter without counting as "looked into")
            return getattr(h0._sfgs_src if h0._sfgs_kind == XYZ else h0.materialise(), _handles.property_name(func))
        if name == "__get__" or name in _METADATA:   # shape, dtype, device, ...: answered by the wrapper's metadata
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        h = args[0] if args and isinstance(args[0], LazyDirs) else None
        if h is not None and h._sfgs_real is None:
            kind = h._sfgs_kind
            if kind == XYZ and name in ("sub", "__sub__") and len(args) == 2 and not kwargs:
                c = args[1]
                if (type(c) is torch.Tensor and tuple(c.shape) == tuple(h.shape) and c.dtype == h.dtype == torch.float32
                        and c.device == h.device and not c.requires_grad and h.dim() == 2 and h.shape[1] == 3):
                    return LazyDirs(DIRPP, (h, c), tuple(h.shape), h)
            elif kind == DIRPP and name == "norm" and len(args) == 1 and set(kwargs) <= {"p", "dim", "keepdim", "dtype"} \
                    and kwargs.get("dim") in (1, -1) and kwargs.get("keepdim") is True \
                    and kwargs.get("p", "fro") in ("fro", 2, 2.0) and kwargs.get("dtype") is None:
                # Tensor.norm(dim=1, keepdim=True): torch's Python wrapper passes p = "fro" and dtype = None along
                return LazyDirs(NORM, h, (h.shape[0], 1), h)
            elif kind == DIRPP and name in ("div", "__truediv__", "true_divide") and len(args) == 2 and not kwargs:
                n = args[1]
                if isinstance(n, LazyDirs) and n._sfgs_kind == NORM and n._sfgs_src is h and n._sfgs_real is None:
                    return LazyDirs(DIRS, (h, n), tuple(h.shape), h)
            elif kind == DIRS and name in ("contiguous", "float") and len(args) == 1 and not kwargs:
                return h
        return func(*cls._unwrap(args), **cls._unwrap(kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):   # backstop: nothing should get here unmaterialised
        return func(*cls._unwrap(args), **cls._unwrap(kwargs or {}))


def centers_of(dirs, means3D):
    """The centres tensor [N,3] when `dirs` is an untouched DIRS handle whose positions are `means3D` (the same tensor
    object: the rasterizer normalises ITS means3D), contiguous; else None (the caller materialises the handle)."""
    if not isinstance(dirs, LazyDirs) or dirs._sfgs_kind != DIRS or dirs._sfgs_real is not None:
        return None
    dirpp, norm = dirs._sfgs_src
    if dirpp._sfgs_real is not None or norm._sfgs_real is not None:
        return None            # someone looked at an intermediate: its values (and graph) are in use, keep everything in torch
    xyz, centers = dirpp._sfgs_src
    if xyz._sfgs_src is not means3D:
        return None
    return centers.contiguous()


def materialise(t):
    return t.materialise() if isinstance(t, LazyDirs) else t


_ORIG = {}


def install(gaussian_model_cls):
    """Patch `get_xyz` on the reference's GaussianModel (a no-op for a class without that property)."""
    if gaussian_model_cls in _ORIG or "get_xyz" not in gaussian_model_cls.__dict__:
        return
    _ORIG[gaussian_model_cls] = gaussian_model_cls.__dict__["get_xyz"]
    gaussian_model_cls.get_xyz = property(lambda self: LazyDirs(XYZ, self._xyz, tuple(self._xyz.shape), self._xyz))


def uninstall(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        gaussian_model_cls.get_xyz = _ORIG.pop(gaussian_model_cls)
