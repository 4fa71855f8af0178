"""Fused per-Gaussian pre-pass of render(): activations + Mip-Splatting 3D filter in one HIP
kernel each way instead of ~40 elementwise torch kernels per training step.

Reference semantics (scene/gaussian_model.py): get_scaling_with_3D_filter :207-213, get_opacity_with_3D_filter
:237-249, get_rotation :216-217 (F.normalize), consumed by gaussian_renderer/__init__.py:61,71-72,137-138.

Two ways to use it:
  * `fused_activations(_scaling, _opacity, _rotation, filter_3D)` -> (scales, opacities, rotations), differentiable;
  * `install(GaussianModel)`: replaces the three properties on the reference's class so that render() and
    train.py run UNCHANGED. By default the getters return `Deferred` handles that the rasterizer resolves INSIDE its
    preprocess kernels (the row as SURVEY 8f words it: "folded into preprocess fwd/bwd" -- no pre-pass launch at all
    on the render() path); any other consumer of a getter gets the tensors of one fused launch per distinct parameter
    version, shared by the three getters (`fold=False`: always that).
"""
import torch

from . import _handles

from . import _lib as L

__all__ = ["fused_activations", "install", "uninstall", "Deferred", "raw_parameters", "materialise", "f64_mask"]


class _FusedActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling_raw, opacity_raw, rotation_raw, filter3d, state=None):
        ctx.sfgs_state = state
        lib = L.load()
        N = int(scaling_raw.shape[0])
        dev = scaling_raw.device
        scales = torch.empty(N, 3, dtype=torch.float32, device=dev)
        opac = torch.empty(N, 1, dtype=torch.float32, device=dev)
        rot = torch.empty(N, 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            L.check(lib.sfgs_prepass_forward(N, L.ptr(scaling_raw), L.ptr(opacity_raw), L.ptr(rotation_raw),
                                             L.ptr(filter3d), f64_mask(filter3d, opacity_raw), L.ptr(scales),
                                             L.ptr(opac), L.ptr(rot), stream))
        ctx.save_for_backward(scaling_raw, opacity_raw, rotation_raw, filter3d)
        return scales, opac, rot

    @staticmethod
    def backward(ctx, g_scales, g_opac, g_rot):
        lib = L.load()
        scaling_raw, opacity_raw, rotation_raw, filter3d = ctx.saved_tensors
        if ctx.sfgs_state is not None:
            ctx.sfgs_state["consumed"] = True   # the getters' cache must not hand this graph out again (see _cached)
        N = int(scaling_raw.shape[0])
        dev = scaling_raw.device
        f32 = dict(dtype=torch.float32, device=dev)
        gs, gr = torch.empty(N, 3, **f32), torch.empty(N, 4, **f32)
        go = torch.empty(N, 1, dtype=opacity_raw.dtype, device=dev)      # float64 after the reference's reset_opacity
        c = lambda t: None if t is None else t.contiguous().float()
        g_scales, g_opac, g_rot = c(g_scales), c(g_opac), c(g_rot)
        with torch.cuda.device(dev):
            stream = L.C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            L.check(lib.sfgs_prepass_backward(N, L.ptr(scaling_raw), L.ptr(opacity_raw), L.ptr(rotation_raw),
                                              L.ptr(filter3d), f64_mask(filter3d, opacity_raw), L.ptr(g_scales),
                                              L.ptr(g_opac), L.ptr(g_rot), L.ptr(gs), L.ptr(go), L.ptr(gr), stream))
        return gs, go, gr, None, None


def f64_mask(filter3d, opacity_raw):
    """SfgsGaussians.raw_f64_mask / the f64_mask argument of sfgs_prepass_*: bit 0 filter float64, bit 1 opacity float64."""
    return int(filter3d.dtype == torch.float64) | (int(opacity_raw.dtype == torch.float64) << 1)


def _checked(scaling_raw, opacity_raw, rotation_raw, filter_3D):
    """Validated, contiguous (raw scaling, raw opacity, raw rotation, detached filter)."""
    N = scaling_raw.shape[0]
    for name, t, shape in (("_scaling", scaling_raw, (N, 3)), ("_opacity", opacity_raw, (N, 1)),
                           ("_rotation", rotation_raw, (N, 4))):
        # `_opacity` is float64 from the reference's first reset_opacity on (scene/gaussian_model.py:483-501)
        ok = (torch.float32, torch.float64) if name == "_opacity" else (torch.float32,)
        if t.dtype not in ok or not t.is_cuda or tuple(t.shape) != shape:
            raise ValueError(f"{name} must be a {'/'.join(str(d) for d in ok)} GPU tensor of shape {shape}")
    if filter_3D.dtype not in (torch.float32, torch.float64) or filter_3D.numel() != N or not filter_3D.is_cuda:
        raise ValueError("filter_3D must be a float32/float64 GPU tensor with one value per Gaussian")
    return scaling_raw.contiguous(), opacity_raw.contiguous(), rotation_raw.contiguous(), filter_3D.detach().contiguous()


def fused_activations(scaling_raw, opacity_raw, rotation_raw, filter_3D, _state=None):
    """(_scaling[N,3], _opacity[N,1], _rotation[N,4], filter_3D[N,1] f32|f64) -> (scales[N,3], opacities[N,1],
    rotations[N,4]) float32, identical to the three reference getters followed by render()'s .float() casts."""
    return _FusedActivations.apply(*_checked(scaling_raw, opacity_raw, rotation_raw, filter_3D), _state)


# ---- deferred getter results: the pre-pass folded into the rasterizer ------------------------------------------
class _Shared:
    """What the three getters of one parameter version share: the raw parameters and, once anything other than the
    rasterizer looked at a value, the materialised activations (one fused launch, with its autograd graph)."""
    __slots__ = ("raw", "real", "state")

    def __init__(self, raw):
        self.raw, self.real, self.state = raw, None, {"consumed": False}

    def materialise(self):
        # once: a materialised handle keeps its values like any tensor would (a backward through them is noted in `state`,
        # and the getters then hand out fresh handles: _cached)
        if self.real is None:
            self.real = fused_activations(*self.raw, self.state)
        return self.real


_METADATA = frozenset(("dim", "ndimension", "numel", "nelement", "size", "__len__", "is_contiguous", "element_size",
                       "is_floating_point", "is_complex", "stride", "storage_offset"))


class Deferred(torch.Tensor):
    """Result of a patched getter (`install(cls, fold=True)`): has the shape / dtype / device of the activated tensor
    but no storage yet. render() only casts it (`.float()`, a no-op on float32: gaussian_renderer/__init__.py:137-138)
    and hands it to GaussianRasterizer, which recognises it and runs the rasterizer in RAW-PARAMETER MODE (include/sfgs.h:
    preprocess / preprocess_bwd apply the activations themselves, no pre-pass kernel, no N-sized intermediates, no
    autograd nodes). ANY other use -- arithmetic, indexing, printing, saving -- materialises the real tensor first (one
    fused_activations launch shared by the three getters) and proceeds on it, so every other caller of the getters
    (save_fused_ply, get_covariance ...) sees ordinary values. Until then the handle stands for the activations of the
    parameters' CURRENT values (render() consumes it at once; a handle kept across an in-place parameter update and only
    then looked at shows the updated values); from its materialisation on it keeps its values like any tensor."""

    @staticmethod
    def __new__(cls, shared, index, shape):
        t = torch.Tensor._make_wrapper_subclass(
            cls, shape, dtype=torch.float32, device=shared.raw[0].device,
            requires_grad=torch.is_grad_enabled() and any(p.requires_grad for p in shared.raw[:3]))
        t._sfgs_shared, t._sfgs_index = shared, index
        return t

    def materialise(self):
        return self._sfgs_shared.materialise()[self._sfgs_index]

    @classmethod
    def _unwrap(cls, x):
        if isinstance(x, Deferred):
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
        if func is torch.Tensor.float and len(args) == 1 and not kwargs:
            return args[0]                       # already float32: .float() returns the tensor itself, like torch
        if name == "__get__" and not _handles.answered_by_wrapper(func):
            # .grad, .grad_fn, ._version, .data ...: properties of the tensor the handle stands for, read from it
            return getattr(args[0].materialise(), _handles.property_name(func))
        if name == "__get__" or name in _METADATA:   # shape, dtype, device, ...: answered by the wrapper's metadata
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        return func(*cls._unwrap(args), **cls._unwrap(kwargs))

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):   # backstop: nothing should get here unmaterialised
        return func(*cls._unwrap(args), **cls._unwrap(kwargs or {}))


def raw_parameters(scales, opacities, rotations):
    """(raw scaling, raw opacity, raw rotation, filter_3D) when the three tensors are the Deferred results of ONE getter
    version; None otherwise (the caller then materialises whichever of them is Deferred)."""
    if not (isinstance(scales, Deferred) and isinstance(opacities, Deferred) and isinstance(rotations, Deferred)):
        return None
    sh = scales._sfgs_shared
    if opacities._sfgs_shared is not sh or rotations._sfgs_shared is not sh:
        return None
    if (scales._sfgs_index, opacities._sfgs_index, rotations._sfgs_index) != (0, 1, 2):
        return None
    return sh.raw


def materialise(t):
    return t.materialise() if isinstance(t, Deferred) else t


# ---- drop-in for the reference's GaussianModel ----------------------------------------------------------------
_ORIG = {}
_FOLD = {}


def _cached(self):
    key = tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype)
                for t in (self._scaling, self._opacity, self._rotation, self.filter_3D))
    key += (torch.is_grad_enabled(),)
    hit = getattr(self, "_sfgs_prepass_cache", None)
    if _FOLD.get(type(self), False):
        if hit is None or hit[0] != key or hit[2]["consumed"] or not isinstance(hit[1][0], Deferred):
            N = int(self._scaling.shape[0])
            raw = _checked(self._scaling, self._opacity, self._rotation, self.filter_3D)
            shared = _Shared(raw)
            hit = (key, tuple(Deferred(shared, i, shp) for i, shp in enumerate(((N, 3), (N, 1), (N, 4)))), shared.state)
            self._sfgs_prepass_cache = hit
        return hit[1]
    # one fused launch serves the three getters of one render() call; once a backward has run through it (its graph
    # is freed) the next getter call recomputes, so two render + backward cycles without an optimizer step in between
    # (gradient accumulation, a skipped step) work like they do with the reference's own getters
    if hit is None or hit[0] != key or hit[2]["consumed"] or isinstance(hit[1][0], Deferred):
        state = {"consumed": False}
        hit = (key, fused_activations(self._scaling, self._opacity, self._rotation, self.filter_3D, state), state)
        self._sfgs_prepass_cache = hit
    return hit[1]


def install(gaussian_model_cls, fold=None):
    """Monkey-patch the reference's GaussianModel: the three getters render() reads become views of ONE fused op.
    fold=True (default; SFGS_PREPASS_FOLD=0 in the environment turns it off): the getters return Deferred handles and the
    rasterizer applies the activations inside its own preprocess kernels (see Deferred); fold=False: they return the
    tensors of one fused_activations launch. With fold, `get_features` is patched as well (sfgs.features: a handle on
    _features_dc / _features_rest instead of their concatenation; SFGS_FEATURES_FOLD=0 keeps the reference's getter) and
    `get_xyz` (sfgs.viewdirs: render()'s view-direction statements recorded and evaluated by the rasterizer; SFGS_DIRS_FOLD=0)."""
    import os
    if fold is None:
        fold = os.environ.get("SFGS_PREPASS_FOLD", "1") != "0"
    _FOLD[gaussian_model_cls] = bool(fold)
    from . import features, viewdirs
    if fold and os.environ.get("SFGS_FEATURES_FOLD", "1") != "0":
        features.install(gaussian_model_cls)      # get_features without the concatenation (sfgs.features)
    else:
        features.uninstall(gaussian_model_cls)
    if fold and os.environ.get("SFGS_DIRS_FOLD", "1") != "0":
        viewdirs.install(gaussian_model_cls)      # render()'s dir_pp normalisation inside the rasterizer (sfgs.viewdirs)
    else:
        viewdirs.uninstall(gaussian_model_cls)
    if gaussian_model_cls in _ORIG:
        return
    _ORIG[gaussian_model_cls] = {n: gaussian_model_cls.__dict__[n] for n in
                                 ("get_scaling_with_3D_filter", "get_opacity_with_3D_filter", "get_rotation")}
    gaussian_model_cls.get_scaling_with_3D_filter = property(lambda self: _cached(self)[0])
    gaussian_model_cls.get_opacity_with_3D_filter = property(lambda self: _cached(self)[1])
    gaussian_model_cls.get_rotation = property(lambda self: _cached(self)[2])


def uninstall(gaussian_model_cls):
    _FOLD.pop(gaussian_model_cls, None)
    from . import features, viewdirs
    features.uninstall(gaussian_model_cls)
    viewdirs.uninstall(gaussian_model_cls)
    for n, v in _ORIG.pop(gaussian_model_cls, {}).items():
        setattr(gaussian_model_cls, n, v)
