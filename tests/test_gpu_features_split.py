"""Split SH storage (include/sfgs.h: SfgsGaussians.shs_rest, ABI 13): the rasterizer reads the model's `_features_dc` and
`_features_rest` themselves instead of `get_features`' concatenation (scene/gaussian_model.py:227-231), through the
DeferredFeatures handle the patched getter returns (sfgs.features). Same float sequence as the concatenated route (which
is pinned to the oracle in tests/test_gpu_raster.py and to the real eval_sh in tests/test_gpu_sh_fold.py), so everything
is compared BIT for bit: images, radii, and the two parameters' gradients against the split of the concatenation's."""
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


def _render(frame, g, deg, gc, gd, split, python_sh=False, raw=None):
    """render()'s statements around one rasterizer call: `shs = pc.get_features` (:127) or, python_sh, the
    convert_SHs_python path (:121-125) with sfgs.sh's folding eval_sh. split: get_features is the handle."""
    from diff_gauss import GaussianRasterizer
    from sfgs import features, sh as sfsh
    dc = g["shs"][:, :1].contiguous().to(DEV).requires_grad_(True)
    rest = g["shs"][:, 1:].contiguous().to(DEV).requires_grad_(True)
    get_features = (lambda: features.DeferredFeatures(dc, rest)) if split else (lambda: torch.cat((dc, rest), dim=1))
    K = g["shs"].shape[1]
    xyz = g["means3D"].to(DEV).requires_grad_(True)
    means2D = torch.zeros_like(xyz, requires_grad=True)
    shs = colors = None
    if python_sh:
        shs_view = get_features().transpose(1, 2).view(-1, 3, K)
        dir_pp = xyz - frame["campos"].to(DEV).repeat(get_features().shape[0], 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        colors = torch.clamp_min(sfsh.eval_sh_deferred(deg, shs_view, dir_pp_normalized) + 0.5, 0.0)
        assert colors.folded_inputs() is not None and isinstance(colors.folded_inputs()[1], tuple) == split
    else:
        shs = get_features()
    if raw is None:
        leaves = {k: g[k].to(DEV).requires_grad_(True) for k in ("scales", "opacities", "rotations")}
        sc, op, ro = leaves["scales"], leaves["opacities"], leaves["rotations"]
    else:
        from sfgs import prepass
        leaves = {k: raw[k].to(DEV).requires_grad_(True) for k in ("scaling", "opacity", "rotation")}
        shared = prepass._Shared(prepass._checked(leaves["scaling"], leaves["opacity"], leaves["rotation"],
                                                  raw["filter"].to(DEV)))
        n = xyz.shape[0]
        sc, op, ro = (prepass.Deferred(shared, i, s) for i, s in enumerate(((n, 3), (n, 1), (n, 4))))
    image, depth, _, alpha, radii, _ = GaussianRasterizer(_settings(frame, deg))(
        means3D=xyz, means2D=means2D, shs=shs, colors_precomp=colors, opacities=op.float(), scales=sc.float(),
        rotations=ro, cov3Ds_precomp=None)
    if split and shs is not None:
        assert shs._sfgs_real is None          # the rasterizer did not concatenate
    torch.autograd.backward([image, depth], [gc, gd])
    out = dict(image=image, depth=depth, alpha=alpha, radii=radii, g_dc=dc.grad, g_rest=rest.grad, g_xyz=xyz.grad,
               g_means2D=means2D.grad)
    out.update({"g_" + k: v.grad for k, v in leaves.items()})
    return out


def _identical(a, b):
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        assert torch.equal(torch.nan_to_num(a[k], nan=-7.0), torch.nan_to_num(b[k], nan=-7.0)), k


@pytest.mark.parametrize("python_sh", [False, True])
@pytest.mark.parametrize("deg,max_deg,n", [(1, 1, 30001), (0, 1, 5000), (2, 2, 20000), (3, 3, 20003), (1, 3, 9999), (4, 4, 7001)])
def test_split_storage_equals_concatenated(deg, max_deg, n, python_sh):
    W, H = 320, 192
    frame, g = scene(n, W, H, seed=31 + deg, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="sh", sh_degree=max_deg)
    assert g["shs"].shape[1] == (max_deg + 1) ** 2
    gc, gd = (t.to(DEV) for t in upstream_grads(W, H, 2))
    ref = _render(frame, g, deg, gc, gd, split=False, python_sh=python_sh)
    got = _render(frame, g, deg, gc, gd, split=True, python_sh=python_sh)
    assert float(ref["g_rest"].abs().max()) > 0 or deg == 0
    _identical(got, ref)


@pytest.mark.parametrize("python_sh", [False, True])
def test_split_storage_composes_with_raw_parameter_mode(python_sh):
    W, H, n = 256, 160, 20001
    frame, g = scene(n, W, H, seed=6, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="sh", sh_degree=2)
    gen = torch.Generator().manual_seed(106)
    raw = dict(scaling=torch.log(g["scales"]), opacity=torch.logit(g["opacities"].double()),
               rotation=g["rotations"] * torch.empty(n, 1).uniform_(0.3, 3.0, generator=gen),
               filter=g["scales"].double().mean(1, keepdim=True) *
                      torch.empty(n, 1, dtype=torch.float64).uniform_(0.1, 1.5, generator=gen))
    gc, gd = (t.to(DEV) for t in upstream_grads(W, H, 5))
    ref = _render(frame, g, 2, gc, gd, split=False, python_sh=python_sh, raw=raw)
    got = _render(frame, g, 2, gc, gd, split=True, python_sh=python_sh, raw=raw)
    _identical(got, ref)
    assert got["g_opacity"].dtype == torch.float64


def test_a_handle_something_looked_into_takes_the_ordinary_route():
    from diff_gauss import GaussianRasterizer
    from sfgs import features
    W, H, n = 128, 96, 3000
    frame, g = scene(n, W, H, seed=2, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="sh", sh_degree=1)
    dc, rest = g["shs"][:, :1].contiguous().to(DEV), g["shs"][:, 1:].contiguous().to(DEV)
    args = dict(means3D=g["means3D"].to(DEV), means2D=None, colors_precomp=None, opacities=g["opacities"].to(DEV),
                scales=g["scales"].to(DEV), rotations=g["rotations"].to(DEV), cov3Ds_precomp=None)
    r = GaussianRasterizer(_settings(frame, 1))
    ref = r(shs=torch.cat((dc, rest), 1), **args)[0]
    h = features.DeferredFeatures(dc, rest)
    _ = float(h.sum())                                     # materialises
    assert torch.equal(r(shs=h, **args)[0], ref)
    assert torch.equal(r(shs=features.DeferredFeatures(dc, rest), **args)[0], ref)
    with pytest.raises(ValueError):                        # the handle's coefficient count is validated like a tensor's
        r(shs=features.DeferredFeatures(dc, rest[:, :1]), **args)
    t = features.DeferredFeatures(dc, rest).transpose(1, 2)   # [N,3,K] view handed to `shs`: not [N,K,3]
    with pytest.raises(ValueError):
        r(shs=t, **args)
