"""train.py:314 without its three host synchronisations ("fuse `max_radii2D`, ...").

Every training iteration below densify_until_iter the reference runs, in training() itself (train.py:311-315),

    gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter])

with `radii` the rasterizer's output and `visibility_filter = radii > 0` built inside render()
(gaussian_renderer/__init__.py:160-162). Three boolean-mask index operations = three `nonzero` kernels, each followed by a
device-to-host copy of the count the HOST waits for, two gathers, a max, a scatter: seven kernels and three stalls of the
launching thread per iteration for what is `m = where(v, max(m, r), m)` -- one pass over N elements, no synchronisation.

The statement lives in a function body, so there is no method to swap. Instead the two tensors it is built from carry the
information: `install()` makes `diff_gauss.GaussianRasterizer` hand `radii` out as a `RadiiTensor` -- a `torch.Tensor`
subclass over the SAME storage -- whose `radii > 0` is a `VisMask` (again a real bool tensor). Indexing anything with a
`VisMask` as the ONLY index yields a `MaskedSelect` note instead of launching `nonzero`; `torch.max` of two notes on the same
mask is a `MaskedMax` note; assigning that note back through the same mask runs the three element-wise kernels. Everything
else anybody does with these objects -- `.sum()`, arithmetic, printing, indexing with a tuple, passing the mask to
add_densification_stats -- sees ordinary tensors: a note that is used in any other way first becomes the tensor it stands
for, by the reference's own operation (`base[mask]`), so no semantics change; only the one statement gets cheaper.

Host logic in plain torch (no kernel of ours: three element-wise torch kernels are already at the floor); `uninstall()`
restores the plain tensors. tests/test_max_radii.py pins the statement, the fall-backs and the no-sync property."""
import torch

__all__ = ["install", "uninstall", "RadiiTensor", "VisMask", "wrap_radii"]

_ON = False


def _plain(t):
    return t.as_subclass(torch.Tensor) if isinstance(t, torch.Tensor) and type(t) is not torch.Tensor else t


class _Note:
    """A deferred result. Any use other than the ones the statement makes turns it into the tensor it stands for."""

    def materialise(self):
        raise NotImplementedError

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.max and len(args) == 2 and not kwargs:
            a, b = args
            if isinstance(a, MaskedSelect) and isinstance(b, MaskedSelect) and a.mask is b.mask:
                return MaskedMax(a, b)
        args = tuple(x.materialise() if isinstance(x, _Note) else x for x in args)
        kwargs = {k: (v.materialise() if isinstance(v, _Note) else v) for k, v in kwargs.items()}
        return func(*args, **kwargs)

    def __getattr__(self, name):            # .sum(), .shape, .dtype, ...: whatever the real tensor answers
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialise(), name)


def _delegate(name):                        # operators and other dunder methods bypass __getattr__
    def op(self, *a):
        return getattr(self.materialise(), name)(*(x.materialise() if isinstance(x, _Note) else x for x in a))
    op.__name__ = name
    return op


for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__neg__", "__lt__",
           "__le__", "__gt__", "__ge__", "__eq__", "__ne__", "__getitem__", "__len__", "__iter__", "__bool__", "__float__", "__int__",
           "__repr__", "__and__", "__or__", "__invert__", "__abs__"):
    setattr(_Note, _n, _delegate(_n))
del _n


class MaskedSelect(_Note):
    """`base[mask]` not yet evaluated."""

    def __init__(self, base, mask):
        self.base, self.mask, self._val = base, mask, None

    def materialise(self):
        if self._val is None:
            self._val = _plain(self.base)[_plain(self.mask)]
        return self._val


class MaskedMax(_Note):
    """`torch.max(a.base[mask], b.base[mask])` not yet evaluated."""

    def __init__(self, a, b):
        self.a, self.b, self.mask = a, b, a.mask

    def materialise(self):
        return torch.max(self.a.materialise(), self.b.materialise())


class _PlainCopies:
    """Copies and pickles of the two marked tensors are ordinary tensors (the marks describe one frame's objects, not data)."""

    def __deepcopy__(self, memo):
        return _plain(self).clone()

    def __reduce_ex__(self, proto):
        return _plain(self).__reduce_ex__(proto)


class VisMask(_PlainCopies, torch.Tensor):
    """`radii > 0`: a real bool tensor that additionally recognises being used as the sole index of a tensor."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if _ON and func is torch.Tensor.__getitem__ and len(args) == 2 and type(args[1]) is VisMask \
                and isinstance(args[0], torch.Tensor) and args[0].dim() >= 1 and args[0].shape[0] == args[1].shape[0] \
                and args[1].dim() == 1:
            return MaskedSelect(args[0], args[1])
        if func is torch.Tensor.__setitem__ and len(args) == 3 and type(args[1]) is VisMask:
            target, mask, value = args
            if isinstance(value, MaskedMax) and value.mask is mask and (value.a.base is target or value.b.base is target):
                other = value.b.base if value.a.base is target else value.a.base
                tp, op, mp = _plain(target), _plain(other), _plain(mask)
                if tp.shape == op.shape == mp.shape:
                    # m[v] = max(m[v], r[v])  ==  m = where(v, max(m, r), m): same values, same type promotion
                    # (int32 radii against the float32 buffer -> float32), no nonzero, no host wait
                    new = torch.where(mp, torch.maximum(tp, op.to(tp.dtype)), tp)
                    with torch._C.DisableTorchFunctionSubclass():
                        tp.copy_(new)
                    return None
            if isinstance(value, _Note):
                value = value.materialise()
            with torch._C.DisableTorchFunctionSubclass():
                return func(_plain(target), _plain(mask), value)
        args = tuple(x.materialise() if isinstance(x, _Note) else x for x in args)
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return _strip(out)


class RadiiTensor(_PlainCopies, torch.Tensor):
    """The rasterizer's `radii` [N] int32: an ordinary tensor whose `> 0` is a VisMask."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if _ON and func in (torch.Tensor.__gt__, torch.gt, torch.Tensor.gt) and len(args) == 2 and type(args[0]) is RadiiTensor \
                and isinstance(args[1], (int, float)) and args[1] == 0 and not kwargs:
            with torch._C.DisableTorchFunctionSubclass():
                m = func(_plain(args[0]), args[1])
            return m.as_subclass(VisMask)
        if _ON and func is torch.Tensor.__getitem__ and len(args) == 2 and type(args[1]) is VisMask:
            return VisMask.__torch_function__(func, types, args, kwargs)
        args = tuple(x.materialise() if isinstance(x, _Note) else x for x in args)
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        return _strip(out)


def _strip(out):
    """Results of any other operation are plain tensors (the subclasses mark two specific objects, not a tensor family)."""
    if isinstance(out, (RadiiTensor, VisMask)):
        return out.as_subclass(torch.Tensor)
    if isinstance(out, (tuple, list)):
        return type(out)(_strip(x) for x in out)
    return out


def wrap_radii(radii):
    """Called by diff_gauss.GaussianRasterizer.forward on its `radii` output (a view change, no copy, no kernel)."""
    return radii.as_subclass(RadiiTensor) if _ON and type(radii) is torch.Tensor else radii


def install(*_ignored):
    """From now on `radii` leaves the rasterizer as a RadiiTensor. Accepts (and ignores) a GaussianModel class so that it can
    be listed with the other hooks (tools/launch_scenes.py: install_hooks)."""
    global _ON
    _ON = True


def uninstall(*_ignored):
    global _ON
    _ON = False
