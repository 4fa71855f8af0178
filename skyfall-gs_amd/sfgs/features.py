"""`GaussianModel.get_features` without the concatenation (the getters feed the op directly).

The reference stores the SH coefficients as two parameters -- `_features_dc` [N,1,3] and `_features_rest` [N,K-1,3] -- and
its getter concatenates them on EVERY call (scene/gaussian_model.py:227-231). render() calls it twice per frame, once for
the values and once only for `.shape[0]` (gaussian_renderer/__init__.py:110+114, 121+122, 127): at 2 M Gaussians and SH
degree 1 that is 2 x 0.08 ms of copy kernels per iteration plus the split of the gradient on the way back; at degree 3,
four times that.

`install(GaussianModel)` (called by `sfgs.prepass.install`) makes the getter return a `DeferredFeatures` handle: shape,
dtype and device of the concatenated tensor, no storage. What render() does with it stays cheap:

  * `.shape[0]`                                       -> answered from the metadata, nothing runs;
  * `shs = pc.get_features` -> GaussianRasterizer     -> the rasterizer reads the two parameters themselves
                                                         (SfgsGaussians.shs_rest) and writes their two gradients;
  * `.transpose(1, 2).view(-1, 3, K)` -> eval_sh      -> stays a handle; the patched eval_sh (sfgs.sh) folds it into the
                                                         rasterizer the same way;
  * anything else (the appearance MLP, indexing ...)  -> ONE torch.cat, with its autograd graph, exactly the reference's value.
"""
import torch

from . import _handles

__all__ = ["DeferredFeatures", "split_parts", "materialise", "install", "uninstall"]

_METADATA = frozenset(("dim", "ndimension", "numel", "nelement", "size", "__len__", "is_contiguous", "element_size",
                       "is_floating_point", "is_complex", "stride", "storage_offset"))


class DeferredFeatures(torch.Tensor):
    """Storage-less stand-in for `torch.cat((_features_dc, _features_rest), dim=1)` ([N,K,3]) or, after
    `.transpose(1, 2)`, for its [N,3,K] transposed VIEW (same memory order: coefficient-major)."""

    @staticmethod
    def __new__(cls, dc, rest, transposed=False):
        n, k = int(dc.shape[0]), int(dc.shape[1] + rest.shape[1])
        shape, strides = ((n, 3, k), (3 * k, 1, 3)) if transposed else ((n, k, 3), (3 * k, 3, 1))
        t = torch.Tensor._make_wrapper_subclass(
            cls, shape, strides=strides, dtype=dc.dtype, device=dc.device,
            requires_grad=torch.is_grad_enabled() and (dc.requires_grad or rest.requires_grad))
        t._sfgs_parts = (dc, rest, bool(transposed))
        t._sfgs_real = None
        return t

    def materialise(self):
        if self._sfgs_real is None:
            dc, rest, transposed = self._sfgs_parts
            v = torch.cat((dc, rest), dim=1)
            self._sfgs_real = v.transpose(1, 2) if transposed else v
        return self._sfgs_real

    @classmethod
    def _unwrap(cls, x):
        if isinstance(x, DeferredFeatures):
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
        h = args[0] if args and isinstance(args[0], DeferredFeatures) else None
        if h is not None and h._sfgs_real is None and not kwargs:
            dc, rest, transposed = h._sfgs_parts
            if name == "float" and len(args) == 1 and h.dtype == torch.float32:
                return h
            if name == "transpose" and len(args) == 3 and {int(args[1]) % 3, int(args[2]) % 3} == {1, 2}:
                return DeferredFeatures(dc, rest, not transposed)
            if name in ("view", "reshape"):         # render(): `.view(-1, 3, K)` of the transposed handle -- the same shape
                shape = args[1] if len(args) == 2 and isinstance(args[1], (tuple, list, torch.Size)) else args[1:]
                if len(shape) == 3 and all(isinstance(s, int) for s in shape):
                    own = tuple(h.shape)
                    if all(s == o or (s == -1 and i == 0) for i, (s, o) in enumerate(zip(shape, own))):
                        return h
        return func(*cls._unwrap(args), **cls._unwrap(kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):   # backstop: nothing should get here unmaterialised
        return func(*cls._unwrap(args), **cls._unwrap(kwargs or {}))


def split_parts(t):
    """(features_dc [N,1,3], features_rest [N,K-1,3], transposed) of a handle nothing has looked into yet -- float32,
    contiguous, K > 1 -- else None (the caller then materialises it)."""
    if not isinstance(t, DeferredFeatures) or t._sfgs_real is not None:
        return None
    dc, rest, transposed = t._sfgs_parts
    if dc.dtype != torch.float32 or rest.dtype != torch.float32 or rest.shape[1] == 0 or rest.device != dc.device:
        return None
    if tuple(dc.shape[1:]) != (1, 3) or rest.dim() != 3 or rest.shape[2] != 3 or rest.shape[0] != dc.shape[0]:
        return None
    return dc.contiguous(), rest.contiguous(), transposed


def materialise(t):
    return t.materialise() if isinstance(t, DeferredFeatures) else t


_ORIG = {}


def install(gaussian_model_cls):
    """Patch `get_features` on the reference's GaussianModel (a no-op for a class without that property)."""
    if gaussian_model_cls in _ORIG or "get_features" not in gaussian_model_cls.__dict__:
        return
    _ORIG[gaussian_model_cls] = gaussian_model_cls.__dict__["get_features"]
    gaussian_model_cls.get_features = property(lambda self: DeferredFeatures(self._features_dc, self._features_rest))


def uninstall(gaussian_model_cls):
    if gaussian_model_cls in _ORIG:
        gaussian_model_cls.get_features = _ORIG.pop(gaussian_model_cls)
