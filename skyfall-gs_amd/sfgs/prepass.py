"""Fused per-Gaussian pre-pass of render(): activations + Mip-Splatting 3D filter in one HIP
kernel each way instead of ~40 elementwise torch kernels per training step.

Reference semantics (scene/gaussian_model.py): get_scaling_with_3D_filter :207-213, get_opacity_with_3D_filter
:237-249, get_rotation :216-217 (F.normalize), consumed by gaussian_renderer/__init__.py:61,71-72,137-138.

Two ways to use it:
  * `fused_activations(_scaling, _opacity, _rotation, filter_3D)` -> (scales, opacities, rotations), differentiable;
  * `install(GaussianModel)`: replaces the three properties on the reference's class so that render() and
    train.py run UNCHANGED (one fused launch per distinct parameter version, shared by the three getters).
"""
import torch

from . import _lib as L

__all__ = ["fused_activations", "install", "uninstall"]


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
                                             L.ptr(filter3d), _f64_mask(filter3d, opacity_raw), L.ptr(scales),
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
                                              L.ptr(filter3d), _f64_mask(filter3d, opacity_raw), L.ptr(g_scales),
                                              L.ptr(g_opac), L.ptr(g_rot), L.ptr(gs), L.ptr(go), L.ptr(gr), stream))
        return gs, go, gr, None, None


def _f64_mask(filter3d, opacity_raw):
    return int(filter3d.dtype == torch.float64) | (int(opacity_raw.dtype == torch.float64) << 1)


def fused_activations(scaling_raw, opacity_raw, rotation_raw, filter_3D, _state=None):
    """(_scaling[N,3], _opacity[N,1], _rotation[N,4], filter_3D[N,1] f32|f64) -> (scales[N,3], opacities[N,1],
    rotations[N,4]) float32, identical to the three reference getters followed by render()'s .float() casts."""
    N = scaling_raw.shape[0]
    for name, t, shape in (("_scaling", scaling_raw, (N, 3)), ("_opacity", opacity_raw, (N, 1)),
                           ("_rotation", rotation_raw, (N, 4))):
        # `_opacity` is float64 from the reference's first reset_opacity on (scene/gaussian_model.py:483-501)
        ok = (torch.float32, torch.float64) if name == "_opacity" else (torch.float32,)
        if t.dtype not in ok or not t.is_cuda or tuple(t.shape) != shape:
            raise ValueError(f"{name} must be a {'/'.join(str(d) for d in ok)} GPU tensor of shape {shape}")
    if filter_3D.dtype not in (torch.float32, torch.float64) or filter_3D.numel() != N or not filter_3D.is_cuda:
        raise ValueError("filter_3D must be a float32/float64 GPU tensor with one value per Gaussian")
    return _FusedActivations.apply(scaling_raw.contiguous(), opacity_raw.contiguous(), rotation_raw.contiguous(),
                                   filter_3D.detach().contiguous(), _state)


# ---- drop-in for the reference's GaussianModel ----------------------------------------------------------------
_ORIG = {}


def _cached(self):
    key = tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype)
                for t in (self._scaling, self._opacity, self._rotation, self.filter_3D))
    key += (torch.is_grad_enabled(),)
    hit = getattr(self, "_sfgs_prepass_cache", None)
    # one fused launch serves the three getters of one render() call; once a backward has run through it (its graph
    # is freed) the next getter call recomputes, so two render + backward cycles without an optimizer step in between
    # (gradient accumulation, a skipped step) work like they do with the reference's own getters
    if hit is None or hit[0] != key or hit[2]["consumed"]:
        state = {"consumed": False}
        hit = (key, fused_activations(self._scaling, self._opacity, self._rotation, self.filter_3D, state), state)
        self._sfgs_prepass_cache = hit
    return hit[1]


def install(gaussian_model_cls):
    """Monkey-patch the reference's GaussianModel: the three getters render() reads become views of ONE fused op."""
    if gaussian_model_cls in _ORIG:
        return
    _ORIG[gaussian_model_cls] = {n: gaussian_model_cls.__dict__[n] for n in
                                 ("get_scaling_with_3D_filter", "get_opacity_with_3D_filter", "get_rotation")}
    gaussian_model_cls.get_scaling_with_3D_filter = property(lambda self: _cached(self)[0])
    gaussian_model_cls.get_opacity_with_3D_filter = property(lambda self: _cached(self)[1])
    gaussian_model_cls.get_rotation = property(lambda self: _cached(self)[2])


def uninstall(gaussian_model_cls):
    for n, v in _ORIG.pop(gaussian_model_cls, {}).items():
        setattr(gaussian_model_cls, n, v)
