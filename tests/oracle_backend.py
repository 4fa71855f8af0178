"""TEST DOUBLE of diff_gauss's backend seam, backed by the CPU oracle (oracle/) -- test infrastructure only.

The product package has exactly one backend (diff_gauss._HipBackend: libsfgs.so on the GPU) and raises on CPU tensors.
To execute the reference's REAL glue -- gaussian_renderer.render() on the real GaussianModel
(/root/reference/gaussian_renderer/__init__.py:19-164) -- on a GPU-less host, a test swaps `diff_gauss._backend` for
this class: render() then runs through our package's own GaussianRasterizer.forward validation layer (argument checks,
dtype / shape / device rules, the 14-field settings tuple) and lands in the C oracle instead of the HIP library. Nothing
under skyfall-gs_amd/ imports this file.

`recording(trace)` additionally records, at the entry of GaussianRasterizer.forward, the exact argument set each
render() call hands the rasterizer (names, dtypes, shapes, strides, the settings tuple) and the oracle's outputs; the
GPU test tests/test_gpu_render_trace.py replays that trace into the HIP path.
"""
import contextlib

import numpy as np
import torch

from oracle import oracle as orc

TENSOR_ARGS = ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations")
SETTING_TENSORS = ("subpixel_offset", "bg", "viewmatrix", "projmatrix", "campos")
SETTING_SCALARS = ("image_height", "image_width", "tanfovx", "tanfovy", "kernel_size", "scale_modifier", "sh_degree",
                   "prefiltered", "debug")


def frame_from_settings(s):
    return dict(H=int(s.image_height), W=int(s.image_width), tanfovx=float(s.tanfovx), tanfovy=float(s.tanfovy),
                kernel_size=float(s.kernel_size), scale_modifier=float(s.scale_modifier), sh_degree=int(s.sh_degree),
                depth_mode=int(getattr(s, "depth_mode", 0)),
                subpix=None if s.subpixel_offset is None else s.subpixel_offset.detach().cpu().numpy(),
                bg=s.bg.detach().cpu().numpy(), view=s.viewmatrix.detach().cpu().numpy(),
                proj=s.projmatrix.detach().cpu().numpy(), campos=s.campos.detach().cpu().numpy())


class _OracleRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, settings):
        R = orc.OracleRender(frame_from_settings(settings), means3D, scales, rotations, opacities,
                             colors_precomp=colors_precomp, shs=shs)
        H, W = R.H, R.W
        color, depth, alpha = (torch.from_numpy(a.copy()) for a in (R.color, R.depth, R.alpha))
        radii = torch.from_numpy(R.radii.copy())
        norm = torch.zeros(1, 1, 1).expand(3, H, W)
        ctx.mark_non_differentiable(radii, norm)
        ctx.set_materialize_grads(False)
        ctx.R, ctx.has_colors, ctx.has_shs = R, colors_precomp is not None, shs is not None
        OracleBackend.last = R
        return color, depth, norm, alpha, radii

    @staticmethod
    def backward(ctx, g_color, g_depth, g_norm, g_alpha, g_radii):
        G = ctx.R.backward(g_color, g_depth, g_alpha)
        OracleBackend.last_grads = G
        t = lambda k: torch.from_numpy(G[k])
        return (t("means3D"), t("means2D"), t("shs") if ctx.has_shs else None,
                t("colors_precomp") if ctx.has_colors else None, t("opacities"), t("scales"), t("rotations"), None)


class OracleBackend:
    name = "oracle-cpu (test double)"
    last = None
    last_grads = None

    @staticmethod
    def check_device(t, name):
        if t.is_cuda:
            raise ValueError(f"{name}: the oracle double takes CPU tensors")

    @staticmethod
    def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, raster_settings):
        return _OracleRasterize.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                      raster_settings)


class _OracleRasterizeAnyDevice(torch.autograd.Function):
    """The same double for tensors on ANY device (the oracle copies its inputs to the host; outputs and gradients go back
    to the inputs' device): lets the reference's real code run on `cuda` around the C oracle, so that a GPU test can
    compare the HIP library with the oracle through the reference's own torch graph (tests/ref_real_driver.py)."""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, settings):
        dev = means3D.device
        R = orc.OracleRender(frame_from_settings(settings), means3D, scales, rotations, opacities,
                             colors_precomp=colors_precomp, shs=shs)
        color, depth, alpha = (torch.from_numpy(a.copy()).to(dev) for a in (R.color, R.depth, R.alpha))
        radii = torch.from_numpy(R.radii.copy()).to(dev)
        norm = torch.zeros(1, 1, 1, device=dev).expand(3, R.H, R.W)
        ctx.mark_non_differentiable(radii, norm)
        ctx.set_materialize_grads(False)
        ctx.R, ctx.has_colors, ctx.has_shs, ctx.dev = R, colors_precomp is not None, shs is not None, dev
        return color, depth, norm, alpha, radii

    @staticmethod
    def backward(ctx, g_color, g_depth, g_norm, g_alpha, g_radii):
        G = ctx.R.backward(g_color, g_depth, g_alpha)
        t = lambda k: torch.from_numpy(G[k]).to(ctx.dev)
        return (t("means3D"), t("means2D"), t("shs") if ctx.has_shs else None,
                t("colors_precomp") if ctx.has_colors else None, t("opacities"), t("scales"), t("rotations"), None)


class OracleBackendAnyDevice:
    name = "oracle-cpu behind device tensors (test double)"

    @staticmethod
    def check_device(t, name):
        pass

    @staticmethod
    def rasterize(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, raster_settings):
        return _OracleRasterizeAnyDevice.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                               raster_settings)


@contextlib.contextmanager
def installed():
    import diff_gauss
    saved = diff_gauss._backend
    diff_gauss._backend = OracleBackend
    try:
        yield OracleBackend
    finally:
        diff_gauss._backend = saved


def describe(t):
    return None if t is None else dict(dtype=str(t.dtype).replace("torch.", ""), shape=list(t.shape),
                                       stride=list(t.stride()), requires_grad=bool(t.requires_grad))


@contextlib.contextmanager
def recording(trace):
    """Record every call of diff_gauss.GaussianRasterizer.forward into `trace` (a list of dicts): the raw arguments as the
    caller passed them (before our validation layer touches them) and the five output tensors."""
    import diff_gauss
    orig = diff_gauss.GaussianRasterizer.forward

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3Ds_precomp=None):
        args = dict(means3D=means3D, means2D=means2D, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
                    scales=scales, rotations=rotations)
        s = self.raster_settings
        rec = dict(meta={k: describe(v) for k, v in args.items()}, cov3Ds_precomp_is_none=cov3Ds_precomp is None,
                   inputs={k: None if v is None else v.detach().cpu().numpy().copy() for k, v in args.items()},
                   settings_scalars={k: getattr(s, k) for k in SETTING_SCALARS},
                   settings_meta={k: describe(getattr(s, k)) for k in SETTING_TENSORS},
                   settings_tensors={k: None if getattr(s, k) is None else getattr(s, k).detach().cpu().numpy().copy()
                                     for k in SETTING_TENSORS},
                   settings_fields=list(type(s)._fields), arg_tensors=args)
        out = orig(self, means3D, means2D, opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
                   rotations=rotations, cov3Ds_precomp=cov3Ds_precomp)
        rec["outputs"] = dict(color=out[0].detach().cpu().numpy().copy(), depth=out[1].detach().cpu().numpy().copy(),
                              alpha=out[3].detach().cpu().numpy().copy(), radii=out[4].detach().cpu().numpy().copy())
        rec["out_tensors"] = out
        trace.append(rec)
        return out
    diff_gauss.GaussianRasterizer.forward = forward
    try:
        yield trace
    finally:
        diff_gauss.GaussianRasterizer.forward = orig


def rebuild(meta, array, device):
    """A tensor with the recorded dtype / shape / STRIDES holding `array` (the GPU replay hands the HIP path tensors laid
    out exactly as render() produced them)."""
    if meta is None:
        return None
    dt = getattr(torch, meta["dtype"])
    t = torch.empty_strided(meta["shape"], meta["stride"], dtype=dt, device=device)
    t.copy_(torch.from_numpy(np.asarray(array)).to(dt))
    return t
