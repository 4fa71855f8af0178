"""Directions from centres (include/sfgs.h: SfgsGaussians.sh_centers, ABI 15): render()'s

    dir_pp = xyz - camera_center.repeat(N, 1);  dirs = dir_pp / dir_pp.norm(dim=1, keepdim=True)

(gaussian_renderer/__init__.py:114-115, :122-123) recorded on the sfgs.viewdirs handle `get_xyz` returns and evaluated
inside preprocess / preprocess_bwd, the direction's gradient added to means3D's there. Compared with the same statements
run by torch around the folding eval_sh (tests/test_gpu_sh_fold.py pins that route to the real eval_sh and the three-step
route): images, radii and every gradient -- xyz.grad carries BOTH roles of the positions -- within float rounding of the
two normalisations (torch's norm kernel vs the kernel's sqrt of the sum of squares); both coefficient layouts, the split
SH storage, raw-parameter mode, degrees 0-4."""
import pytest
import torch

from sfgs.synth import scene, upstream_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _settings(frame, deg):
    from diff_gauss import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=frame["H"], image_width=frame["W"], tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(DEV),
        scale_modifier=frame["scale_modifier"], viewmatrix=frame["view"].to(DEV), projmatrix=frame["proj"].to(DEV),
        sh_degree=deg, campos=frame["campos"].to(DEV), prefiltered=False, debug=False)


def _render(frame, g, sh_cm, deg, gc, gd, lazy, path, raw=None, center=None):
    """render()'s statements of one Python colour path around one rasterizer call. path "mlp": the coefficients are a
    contiguous [N,3,K] tensor (the appearance MLP's output, :111); "features": `pc.get_features.transpose(1, 2).view(...)`
    (:121); "features_split": that with sfgs.features' handle. lazy: `pc.get_xyz` is sfgs.viewdirs' handle."""
    from diff_gauss import GaussianRasterizer
    from sfgs import features, sh as sfsh, viewdirs
    xyz = torch.nn.Parameter(g["means3D"].to(DEV))
    get_xyz = (lambda: viewdirs.LazyDirs(viewdirs.XYZ, xyz, tuple(xyz.shape), xyz)) if lazy else (lambda: xyz)
    K = sh_cm.shape[2]
    if path == "mlp":
        leaves = [sh_cm.to(DEV).requires_grad_(True)]
        sh = leaves[0]
    else:
        feats = sh_cm.transpose(1, 2).contiguous().to(DEV)
        leaves = [feats[:, :1].contiguous().requires_grad_(True), feats[:, 1:].contiguous().requires_grad_(True)]
        get_features = ((lambda: features.DeferredFeatures(*leaves)) if path == "features_split"
                        else (lambda: torch.cat(leaves, dim=1)))
        sh = get_features().transpose(1, 2).view(-1, 3, K)
    camera_center = frame["campos"].to(DEV) if center is None else center.to(DEV)
    means3D = get_xyz()
    dir_pp = (get_xyz() - camera_center.repeat(xyz.shape[0], 1))
    dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    colors = sfsh.eval_sh_deferred(deg, sh, dir_pp_normalized)
    colors = torch.clamp_min(colors + 0.5, 0.0)
    fold = colors.folded_inputs()
    assert fold is not None and isinstance(fold[2], viewdirs.LazyDirs) == lazy
    means2D = torch.zeros_like(xyz, requires_grad=True)
    if raw is None:
        par = {k: g[k].to(DEV).requires_grad_(True) for k in ("scales", "opacities", "rotations")}
        sc, op, ro = par["scales"], par["opacities"], par["rotations"]
    else:
        from sfgs import prepass
        par = {k: raw[k].to(DEV).requires_grad_(True) for k in ("scaling", "opacity", "rotation")}
        shared = prepass._Shared(prepass._checked(par["scaling"], par["opacity"], par["rotation"], raw["filter"].to(DEV)))
        n = xyz.shape[0]
        sc, op, ro = (prepass.Deferred(shared, i, s) for i, s in enumerate(((n, 3), (n, 1), (n, 4))))
    image, depth, _, alpha, radii, _ = GaussianRasterizer(_settings(frame, deg))(
        means3D=means3D, means2D=means2D, shs=None, colors_precomp=colors, opacities=op.float(), scales=sc.float(),
        rotations=ro, cov3Ds_precomp=None)
    if lazy:
        assert dir_pp_normalized._sfgs_real is None and dir_pp._sfgs_real is None     # never evaluated by torch
    torch.autograd.backward([image, depth], [gc, gd])
    out = dict(image=image, depth=depth, alpha=alpha, radii=radii, g_xyz=xyz.grad, g_means2D=means2D.grad)
    out.update({f"g_sh{i}": t.grad for i, t in enumerate(leaves)})
    out.update({"g_" + k: v.grad for k, v in par.items()})
    return out


def _close(a, b, name, rtol=2e-5, atol_rel=2e-6):
    import numpy as np
    a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    scale = max(float(np.abs(b).max()), 1e-30)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * scale, err_msg=name, equal_nan=True)


def _compare(got, ref):
    torch.testing.assert_close(got["radii"], ref["radii"], rtol=0, atol=0)
    for k in ("image", "depth", "alpha"):
        _close(got[k], ref[k], k, rtol=1e-5, atol_rel=1e-6)
    for k in ref:
        if k.startswith("g_"):
            _close(got[k], ref[k], k)


@pytest.mark.parametrize("path", ["mlp", "features", "features_split"])
@pytest.mark.parametrize("deg,stored", [(0, 4), (1, 4), (2, 9), (3, 16), (1, 16), (4, 25)])
def test_directions_from_centres_equal_the_torch_statements(deg, stored, path):
    W, H, n = 320, 192, 30000
    frame, g = scene(n, W, H, seed=41 + deg, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="precomp")
    gen = torch.Generator().manual_seed(17 + stored)
    sh_cm = torch.randn(n, 3, stored, generator=gen)
    sh_cm[:, :, 1:] *= 0.4
    sh_cm[:, :, 0] -= 0.6
    gc, gd = (t.to(DEV) for t in upstream_grads(W, H, 3))
    ref = _render(frame, g, sh_cm, deg, gc, gd, lazy=False, path=path)
    got = _render(frame, g, sh_cm, deg, gc, gd, lazy=True, path=path)
    assert float(ref["g_xyz"].abs().max()) > 0
    _compare(got, ref)


def test_any_centres_tensor_and_raw_parameter_mode():
    """The subtrahend is taken as render() built it -- here NOT the frame's camera position, and different per Gaussian --
    and the mode composes with the raw-parameter mode (activations + 3D filter inside the same kernels)."""
    W, H, n = 256, 160, 20000
    frame, g = scene(n, W, H, seed=8, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="precomp")
    gen = torch.Generator().manual_seed(108)
    raw = dict(scaling=torch.log(g["scales"]), opacity=torch.logit(g["opacities"].double()),
               rotation=g["rotations"] * torch.empty(n, 1).uniform_(0.3, 3.0, generator=gen),
               filter=g["scales"].double().mean(1, keepdim=True) *
                      torch.empty(n, 1, dtype=torch.float64).uniform_(0.1, 1.5, generator=gen))
    sh_cm = torch.randn(n, 3, 9, generator=gen) * 0.5
    gc, gd = (t.to(DEV) for t in upstream_grads(W, H, 6))

    class PerGaussianCentre:           # `.repeat(N, 1)` of this "camera centre" returns a full [N,3] field
        def __init__(self, t): self.t = t
        def to(self, dev): return PerGaussianCentre(self.t.to(dev))
        def repeat(self, n_, one): return self.t

    centre = PerGaussianCentre(torch.randn(n, 3, generator=gen) * 30.0)
    ref = _render(frame, g, sh_cm, 2, gc, gd, lazy=False, path="features_split", raw=raw, center=centre)
    got = _render(frame, g, sh_cm, 2, gc, gd, lazy=True, path="features_split", raw=raw, center=centre)
    _compare(got, ref)
    assert got["g_opacity"].dtype == torch.float64


def test_positions_of_another_tensor_take_the_torch_route():
    """The library normalises ITS means3D: a direction handle recorded on other positions is evaluated by torch."""
    from diff_gauss import GaussianRasterizer
    from sfgs import sh as sfsh, viewdirs
    W, H, n = 128, 96, 4000
    frame, g = scene(n, W, H, seed=3, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="precomp")
    xyz = torch.nn.Parameter(g["means3D"].to(DEV))
    other = torch.nn.Parameter(g["means3D"].to(DEV) + 1.0)
    sh = (torch.randn(n, 3, 4, generator=torch.Generator().manual_seed(1)) * 0.4).to(DEV)
    c = frame["campos"].to(DEV).repeat(n, 1)
    args = dict(means2D=None, shs=None, opacities=g["opacities"].to(DEV), scales=g["scales"].to(DEV),
                rotations=g["rotations"].to(DEV), cov3Ds_precomp=None)
    r = GaussianRasterizer(_settings(frame, 1))

    def colours(positions, lazy):
        p = viewdirs.LazyDirs(viewdirs.XYZ, positions, tuple(positions.shape), positions) if lazy else positions
        d = p - c
        return torch.clamp_min(sfsh.eval_sh_deferred(1, sh, d / d.norm(dim=1, keepdim=True)) + 0.5, 0.0)

    ref = r(means3D=xyz, colors_precomp=colours(other, False), **args)[0]
    col = colours(other, True)
    got = r(means3D=xyz, colors_precomp=col, **args)[0]
    assert col.folded_inputs()[2]._sfgs_real is not None          # materialised by the validation layer
    assert torch.equal(got, ref)
