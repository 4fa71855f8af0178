"""SURVEY 8f row 1 as the row words it: the activations + Mip-Splatting 3D filter FOLDED INTO preprocess fwd/bwd
(include/sfgs.h, SfgsGaussians raw-parameter mode).

The folded route must be indistinguishable from the two-step one it replaces -- sfgs.prepass.fused_activations (pinned to
goldens from the reference's real getters, tests/test_prepass.py), then the rasterizer on its outputs (pinned to the
oracle, tests/test_gpu_raster.py): both routes run the same device functions (csrc/act_math.h), so images, radii and
every gradient are compared for EQUALITY, over all four (filter dtype, raw opacity dtype) combinations the reference
produces (float64 filter during training, float64 _opacity after reset_opacity: scene/gaussian_model.py:258-308,483-501)
and both colour paths. The Deferred handles that make the reference's render() reach that mode unchanged are tested on
a class shaped like its GaussianModel."""
import numpy as np
import pytest
import torch

from sfgs.synth import scene, upstream_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _settings(frame, debug=False):
    from diff_gauss import GaussianRasterizationSettings
    sub = frame.get("subpix")
    return GaussianRasterizationSettings(
        image_height=frame["H"], image_width=frame["W"], tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"], subpixel_offset=None if sub is None else sub.to(DEV), bg=frame["bg"].to(DEV),
        scale_modifier=frame["scale_modifier"], viewmatrix=frame["view"].to(DEV), projmatrix=frame["proj"].to(DEV),
        sh_degree=frame["sh_degree"], campos=frame["campos"].to(DEV), prefiltered=False, debug=debug)


def _raw_scene(n, W, H, mode, filter_dtype, opacity_dtype, seed=3, **kw):
    """A synthetic scene expressed in the model's RAW parameters: _scaling = log s, _opacity = logit o (float64 after
    reset_opacity), _rotation = unnormalised quaternion, filter_3D comparable to the scales (so that it matters)."""
    frame, g = scene(n, W, H, seed=seed, mode=mode, sh_degree=1, **kw)
    gen = torch.Generator().manual_seed(100 + seed)
    raw = dict(scaling=torch.log(g["scales"]),
               opacity=torch.logit(g["opacities"].double()).to(opacity_dtype),
               rotation=g["rotations"] * torch.empty(n, 1).uniform_(0.3, 3.0, generator=gen),
               filter=(g["scales"].double().mean(1, keepdim=True) *
                       torch.empty(n, 1, dtype=torch.float64).uniform_(0.1, 1.5, generator=gen)).to(filter_dtype))
    return frame, g, raw


def _run(frame, g, raw, folded, gc, gd):
    from diff_gauss import GaussianRasterizer
    from sfgs import prepass
    leaves = {k: raw[k].to(DEV).requires_grad_(True) for k in ("scaling", "opacity", "rotation")}
    filt = raw["filter"].to(DEV)
    means3D = g["means3D"].to(DEV).requires_grad_(True)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    col = None if g["colors_precomp"] is None else g["colors_precomp"].to(DEV).requires_grad_(True)
    shs = None if g["shs"] is None else g["shs"].to(DEV).requires_grad_(True)
    if folded:
        shared = prepass._Shared(prepass._checked(leaves["scaling"], leaves["opacity"], leaves["rotation"], filt))
        n = means3D.shape[0]
        sc, op, ro = (prepass.Deferred(shared, i, s) for i, s in enumerate(((n, 3), (n, 1), (n, 4))))
    else:
        sc, op, ro = prepass.fused_activations(leaves["scaling"], leaves["opacity"], leaves["rotation"], filt)
        shared = None
    # exactly what render() does with the getters' results (gaussian_renderer/__init__.py:132-140)
    color, depth, norm, alpha, radii, _ = GaussianRasterizer(_settings(frame))(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=col, opacities=op.float(), scales=sc.float(),
        rotations=ro, cov3Ds_precomp=None)
    if folded:
        assert shared.real is None, "the rasterizer materialised the Deferred handles instead of using raw mode"
    ((color * gc.to(DEV)).sum() + (torch.nan_to_num(depth, nan=0.0) * gd.to(DEV)).sum() + 1e-3 * alpha.sum()).backward()
    out = dict(color=color, depth=depth, alpha=alpha, radii=radii, g_means3D=means3D.grad, g_means2D=means2D.grad,
               g_scaling=leaves["scaling"].grad, g_opacity=leaves["opacity"].grad, g_rotation=leaves["rotation"].grad)
    out["g_colour"] = (col if col is not None else shs).grad
    return {k: v.detach().cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("mode", ["precomp", "sh"])
@pytest.mark.parametrize("filter_dtype,opacity_dtype", [(torch.float64, torch.float32), (torch.float64, torch.float64),
                                                        (torch.float32, torch.float32), (torch.float32, torch.float64)])
def test_folded_route_equals_prepass_then_rasterizer(mode, filter_dtype, opacity_dtype):
    n, W, H = 60_000, 480, 272
    frame, g, raw = _raw_scene(n, W, H, mode, filter_dtype, opacity_dtype, zrange=(60., 90.), scale_range=(0.05, 0.5))
    gc, gd = upstream_grads(W, H, 1)
    a = _run(frame, g, raw, False, gc, gd)
    b = _run(frame, g, raw, True, gc, gd)
    assert (a["radii"] > 0).sum() > 0.9 * n
    assert b["g_opacity"].dtype == (np.float64 if opacity_dtype == torch.float64 else np.float32)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def test_folded_route_zero_gradients_for_culled_gaussians():
    """Gaussians behind the camera / off screen: radii 0 and all-zero raw gradients in the raw opacity's dtype."""
    n, W, H = 20_000, 256, 160
    frame, g, raw = _raw_scene(n, W, H, "precomp", torch.float64, torch.float64, zrange=(20., 40.), xy_fill=2.5)
    g["means3D"][::7, 2] = -5.0
    gc, gd = upstream_grads(W, H, 2)
    a = _run(frame, g, raw, False, gc, gd)
    b = _run(frame, g, raw, True, gc, gd)
    culled = b["radii"] == 0
    assert culled.sum() > n // 7
    for k in ("g_scaling", "g_opacity", "g_rotation"):
        assert not b[k][culled].any(), k
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


class _Model:
    """The attribute / property names of scene/gaussian_model.py that the hook and render() touch."""

    def __init__(self, raw):
        self._scaling = raw["scaling"].to(DEV).requires_grad_(True)
        self._opacity = raw["opacity"].to(DEV).requires_grad_(True)
        self._rotation = raw["rotation"].to(DEV).requires_grad_(True)
        self.filter_3D = raw["filter"].to(DEV)

    get_scaling_with_3D_filter = property(lambda self: None)
    get_opacity_with_3D_filter = property(lambda self: None)
    get_rotation = property(lambda self: None)


@pytest.mark.parametrize("fold", [True, False])
def test_install_fold_on_a_gaussian_model_shaped_class(fold, monkeypatch):
    """install(cls, fold=True): the getters hand out Deferred handles, render()'s `.float()` keeps them, the rasterizer
    takes the raw route (no fused_activations launch at all); every OTHER use of a getter materialises ordinary values.
    fold=False: one fused_activations launch per parameter version. Same numbers either way."""
    from diff_gauss import GaussianRasterizer
    from sfgs import prepass
    n, W, H = 30_000, 320, 200
    frame, g, raw = _raw_scene(n, W, H, "precomp", torch.float64, torch.float32, zrange=(40., 60.))
    gc, gd = upstream_grads(W, H, 3)
    want = _run(frame, g, raw, False, gc, gd)
    launches = []
    real_apply = prepass._FusedActivations.apply
    monkeypatch.setattr(prepass._FusedActivations, "apply", lambda *a: (launches.append(1), real_apply(*a))[1])
    prepass.install(_Model, fold=fold)
    try:
        m = _Model(raw)
        means3D = g["means3D"].to(DEV)
        col = g["colors_precomp"].to(DEV)
        sc, op, ro = m.get_scaling_with_3D_filter, m.get_opacity_with_3D_filter, m.get_rotation
        assert isinstance(sc, prepass.Deferred) == fold
        assert tuple(sc.shape) == (n, 3) and tuple(op.shape) == (n, 1) and ro.dtype == torch.float32 and sc.is_cuda
        assert sc.float() is sc and len(ro) == n and op.numel() == n and sc.dim() == 2
        assert len(launches) == (0 if fold else 1)           # metadata and .float() never materialise
        color, depth, _, alpha, radii, _ = GaussianRasterizer(_settings(frame))(
            means3D=means3D, means2D=None, shs=None, colors_precomp=col, opacities=op.float(), scales=sc.float(),
            rotations=ro, cov3Ds_precomp=None)
        ((color * gc.to(DEV)).sum() + (torch.nan_to_num(depth, nan=0.0) * gd.to(DEV)).sum() + 1e-3 * alpha.sum()).backward()
        assert len(launches) == (0 if fold else 1)
        np.testing.assert_array_equal(color.detach().cpu().numpy(), want["color"])
        np.testing.assert_array_equal(radii.cpu().numpy(), want["radii"])
        for k, p in (("g_scaling", m._scaling), ("g_opacity", m._opacity), ("g_rotation", m._rotation)):
            np.testing.assert_array_equal(p.grad.cpu().numpy(), want[k], err_msg=k)
        # any other consumer (save_fused_ply, get_covariance ...) sees ordinary tensors -- one launch for the three
        vals = [t * 1.0 for t in (m.get_scaling_with_3D_filter, m.get_opacity_with_3D_filter, m.get_rotation)]
        assert len(launches) == (1 if fold else 2)   # (fold=False: the first launch's graph was consumed by the backward)
        ref = prepass.fused_activations(m._scaling, m._opacity, m._rotation, m.filter_3D)
        for v, r in zip(vals, ref):
            assert type(v) is torch.Tensor and torch.equal(v, r)
        assert torch.equal(torch.cat([m.get_rotation, m.get_rotation])[:n], ref[2])
        # a mixed call (own scales, the model's rotation / opacity) falls back to materialised values
        color2, *_ = GaussianRasterizer(_settings(frame))(
            means3D=means3D, means2D=None, shs=None, colors_precomp=col, opacities=m.get_opacity_with_3D_filter,
            scales=ref[0].detach(), rotations=m.get_rotation, cov3Ds_precomp=None)
        np.testing.assert_array_equal(color2.detach().cpu().numpy(), want["color"])
        with torch.no_grad():   # an optimiser step: new parameter version, new handles
            m._scaling.add_(0.05)
        assert m.get_scaling_with_3D_filter is not sc
        assert not torch.equal(m.get_scaling_with_3D_filter + 0, ref[0])
    finally:
        prepass.uninstall(_Model)
