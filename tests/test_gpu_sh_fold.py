"""SURVEY 8f row 1, last part: render()'s `clamp_min(eval_sh(deg, sh, dirs) + 0.5, 0.0)` (gaussian_renderer/__init__.py
:112-118 appearance path, :121-125 convert_SHs_python) FOLDED INTO preprocess fwd/bwd (include/sfgs.h:
SfgsGaussians.sh_dirs).

The folded route must be indistinguishable from the three-step one it replaces -- sfgs.sh.eval_sh (pinned to goldens from
the reference's REAL eval_sh with autograd, tests/test_sh_eval.py), `+ 0.5`, `clamp_min`, then the rasterizer on
colors_precomp (pinned to the oracle, tests/test_gpu_raster.py). Compared: images, radii and every gradient (the
coefficients channel-major as eval_sh takes them, the directions' gradient as it arrives at dir_pp_normalized, means3D
through BOTH its roles, scales ...), degrees 0-3, with and without raw-parameter mode; the golden vectors of the real
eval_sh pushed through the folded route; and that any OTHER use of the handle yields ordinary eval_sh values."""
import os

import numpy as np
import pytest
import torch

from sfgs.synth import scene, upstream_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_sh_grad.npz"))


def _settings(frame, deg):
    from diff_gauss import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=frame["H"], image_width=frame["W"], tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(DEV),
        scale_modifier=frame["scale_modifier"], viewmatrix=frame["view"].to(DEV), projmatrix=frame["proj"].to(DEV),
        sh_degree=deg, campos=frame["campos"].to(DEV), prefiltered=False, debug=False)


def _render_like_the_reference(frame, g, sh_cm, deg, eval_fn, gc, gd, raw=None, features_view=False):
    """The statements of render()'s Python colour path around one rasterizer call; returns outputs and gradients.
    features_view: the coefficients reach eval_sh as convert_SHs_python passes them (:121-122), a transposed view of
    the model's [N,K,3] features; otherwise as the appearance path does (:111), a contiguous [N,3,K] tensor."""
    from diff_gauss import GaussianRasterizer
    xyz = g["means3D"].to(DEV).requires_grad_(True)
    if features_view:
        feats = sh_cm.transpose(1, 2).contiguous().to(DEV).requires_grad_(True)       # pc.get_features: [N,K,3]
        sh = feats.transpose(1, 2).view(-1, 3, sh_cm.shape[2])
        sh_leaf = feats
    else:
        sh = sh_leaf = sh_cm.to(DEV).requires_grad_(True)
    campos = frame["campos"].to(DEV)
    dir_pp = xyz - campos.repeat(xyz.shape[0], 1)
    dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    dir_pp_normalized.retain_grad()
    colors = eval_fn(deg, sh, dir_pp_normalized)
    colors = torch.clamp_min(colors + 0.5, 0.0)
    means2D = torch.zeros_like(xyz, requires_grad=True)
    if raw is None:
        scales = g["scales"].to(DEV).requires_grad_(True)
        rots = g["rotations"].to(DEV).requires_grad_(True)
        opac = g["opacities"].to(DEV).requires_grad_(True)
        leaves = dict(scales=scales, rotations=rots, opacities=opac)
        sc, op, ro = scales, opac, rots
    else:
        from sfgs import prepass
        leaves = {k: raw[k].to(DEV).requires_grad_(True) for k in ("scaling", "opacity", "rotation")}
        shared = prepass._Shared(prepass._checked(leaves["scaling"], leaves["opacity"], leaves["rotation"],
                                                  raw["filter"].to(DEV)))
        n = xyz.shape[0]
        sc, op, ro = (prepass.Deferred(shared, i, s) for i, s in enumerate(((n, 3), (n, 1), (n, 4))))
    image, depth, _, alpha, radii, _ = GaussianRasterizer(_settings(frame, deg))(
        means3D=xyz, means2D=means2D, shs=None, colors_precomp=colors, opacities=op.float(), scales=sc.float(),
        rotations=ro, cov3Ds_precomp=None)
    torch.autograd.backward([image, depth], [gc, gd])
    g_sh = sh_leaf.grad.transpose(1, 2) if features_view else sh_leaf.grad
    out = dict(image=image, depth=depth, alpha=alpha, radii=radii, g_sh=g_sh, g_xyz=xyz.grad, g_means2D=means2D.grad,
               g_dirs=dir_pp_normalized.grad)
    out.update({"g_" + k: v.grad for k, v in leaves.items()})
    return out, colors


def _close(a, b, name, rtol=2e-5, atol_rel=2e-6):
    a, b = a.detach().double().cpu().numpy(), b.detach().double().cpu().numpy()
    scale = max(float(np.abs(b).max()), 1e-30)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol_rel * scale, err_msg=name, equal_nan=True)


@pytest.mark.parametrize("features_view", [False, True])
@pytest.mark.parametrize("deg,stored", [(0, 1), (0, 4), (1, 4), (2, 9), (1, 16), (3, 16), (4, 25), (2, 25)])
def test_folded_route_equals_eval_sh_then_colors_precomp(deg, stored, features_view):
    from sfgs import sh as sfsh
    W, H, n = 320, 192, 30000
    frame, g = scene(n, W, H, seed=21 + deg, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="precomp")
    gen = torch.Generator().manual_seed(7 + stored)
    sh_cm = torch.randn(n, 3, stored, generator=gen)
    sh_cm[:, :, 1:] *= 0.4
    sh_cm[:, :, 0] -= 0.6          # a good share of the channels ends below -0.5: the clamp and its mask matter
    gc, gd = (t.to(DEV) for t in upstream_grads(W, H, 3))
    ref, ref_col = _render_like_the_reference(frame, g, sh_cm, deg, sfsh.eval_sh, gc, gd, features_view=features_view)
    got, got_col = _render_like_the_reference(frame, g, sh_cm, deg, sfsh.eval_sh_deferred, gc, gd,
                                              features_view=features_view)
    assert not isinstance(ref_col, sfsh.DeferredColor)
    assert isinstance(got_col, sfsh.DeferredColor) and got_col.folded_inputs() is not None   # never materialised
    assert got_col.folded_inputs()[3] == (not features_view or stored == 1)   # [N,K,3] handed over as it is (K = 1: both)
    assert float((ref_col == 0).float().mean()) > 0.05
    torch.testing.assert_close(got["radii"], ref["radii"], rtol=0, atol=0)
    for k in ("image", "depth", "alpha"):
        _close(got[k], ref[k], k, rtol=1e-5, atol_rel=1e-6)
    for k in ("g_sh", "g_dirs", "g_xyz", "g_means2D", "g_scales", "g_rotations", "g_opacities"):
        _close(got[k], ref[k], k)
    # coefficients beyond the active degree receive exactly zero, like eval_sh's
    M = (deg + 1) ** 2
    assert float(got["g_sh"][:, :, M:].abs().max() if stored > M else 0.0) == 0.0


def test_folded_colour_path_composes_with_raw_parameter_mode():
    from sfgs import sh as sfsh
    W, H, n = 256, 160, 20000
    frame, g = scene(n, W, H, seed=5, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="precomp")
    gen = torch.Generator().manual_seed(105)
    raw = dict(scaling=torch.log(g["scales"]), opacity=torch.logit(g["opacities"].double()),
               rotation=g["rotations"] * torch.empty(n, 1).uniform_(0.3, 3.0, generator=gen),
               filter=g["scales"].double().mean(1, keepdim=True) *
                      torch.empty(n, 1, dtype=torch.float64).uniform_(0.1, 1.5, generator=gen))
    sh_cm = torch.randn(n, 3, 4, generator=gen) * 0.5
    gc, gd = (t.to(DEV) for t in upstream_grads(W, H, 4))
    ref, _ = _render_like_the_reference(frame, g, sh_cm, 1, sfsh.eval_sh, gc, gd, raw=raw)
    got, col = _render_like_the_reference(frame, g, sh_cm, 1, sfsh.eval_sh_deferred, gc, gd, raw=raw)
    assert col.folded_inputs() is not None
    torch.testing.assert_close(got["radii"], ref["radii"], rtol=0, atol=0)
    for k in ("image", "depth", "alpha"):
        _close(got[k], ref[k], k, rtol=1e-5, atol_rel=1e-6)
    for k in ("g_sh", "g_dirs", "g_xyz", "g_means2D", "g_scaling", "g_rotation", "g_opacity"):
        _close(got[k], ref[k], k)
    assert got["g_opacity"].dtype == torch.float64


G4 = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_sh4.npz"))   # the real eval_sh at degree 4


def _golden(deg, key):
    return G4[key] if deg == 4 else G[f"shg{deg}_{key}"]


@pytest.mark.parametrize("shift", [0.0, 12.0])
@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_golden_vectors_of_the_real_eval_sh_through_the_folded_route(deg, shift):
    """The REAL utils/sh_utils.py eval_sh's inputs / outputs / autograd gradients (tests/golden/make_golden.py) pushed
    through the rasterizer's own evaluation: every Gaussian is parked alone on its own pixel with opacity 0.99 and a
    tiny footprint, so that the rendered pixel IS alpha * colour and dL/dcolour reaches the coefficients undiluted.
    shift = 0: the golden values as they are (a third of the channels is clamped at 0); shift = 12: the DC coefficients
    raised by 12 (the value moves by C0 * 12, exactly known; the gradients do not move: eval_sh is linear in sh) so that
    no channel is clamped and the golden gradients apply to every element."""
    from diff_gauss import GaussianRasterizer
    from sfgs import sh as sfsh
    sh_np, dirs_np, want = _golden(deg, "sh").copy(), _golden(deg, "dirs"), _golden(deg, "out").astype(np.float64)
    sh_np[:, :, 0] += np.float32(shift)
    want = want + 0.28209479177387814 * shift
    n = sh_np.shape[0]
    side = int(np.ceil(np.sqrt(n)))
    W = H = side * 8
    frame, g = scene(n, W, H, seed=1, mode="precomp")
    # one Gaussian per 8x8 tile centre, 100 units in front of the camera
    ix, iy = np.arange(n) % side, np.arange(n) // side
    z = 100.0
    px, py = ix * 8 + 3.5, iy * 8 + 3.5
    fx, fy = W / (2 * frame["tanfovx"]), H / (2 * frame["tanfovy"])
    xyz = torch.tensor(np.stack([(px - (W - 1) / 2) / fx * z, (py - (H - 1) / 2) / fy * z, np.full(n, z)], 1),
                       dtype=torch.float32, device=DEV)
    sh = torch.tensor(sh_np, device=DEV, requires_grad=True)
    dirs = torch.tensor(dirs_np, device=DEV, requires_grad=True)
    colors = torch.clamp_min(sfsh.eval_sh_deferred(deg, sh, dirs) + 0.5, 0.0)
    assert colors.folded_inputs() is not None
    s = z / fx * 0.6     # sigma = 0.6 px
    image, _, _, alpha, radii, _ = GaussianRasterizer(_settings(frame, deg))(
        means3D=xyz, means2D=torch.zeros_like(xyz), shs=None, colors_precomp=colors,
        opacities=torch.full((n, 1), 0.99, device=DEV), scales=torch.full((n, 3), s, device=DEV),
        rotations=torch.tensor([[1., 0, 0, 0]], device=DEV).repeat(n, 1), cov3Ds_precomp=None)
    assert int((radii > 0).sum()) == n
    cy, cx = torch.tensor(iy * 8 + 4, device=DEV), torch.tensor(ix * 8 + 4, device=DEV)   # a pixel next to the centre
    a = alpha[0, cy, cx]
    got = (image[:, cy, cx] / a).T
    ref = np.maximum(want + 0.5, 0.0)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref, rtol=2e-5, atol=3e-6)
    assert bool((ref == 0).any()) == (shift == 0.0)
    # gradients: d(sum w * image_pixel)/d(sh, dirs) = alpha * [colour > 0] * eval_sh's golden gradients
    wgt = torch.tensor(_golden(deg, "w"), device=DEV)
    loss = (image[:, cy, cx].T * wgt / a.detach()[:, None]).sum()
    loss.backward()
    live = torch.tensor((want + 0.5 > 0).astype(np.float32), device=DEV)            # the clamp's mask
    if float(live.min()) == 1.0:
        np.testing.assert_allclose(sh.grad.cpu().numpy(), _golden(deg, "g_sh"), rtol=3e-4, atol=3e-5)
        np.testing.assert_allclose(dirs.grad.cpu().numpy(), _golden(deg, "g_dirs"), rtol=3e-4, atol=1e-4)
    else:   # clamped channels contribute nothing: compare channel-wise on the live ones
        M = sh.grad * live[:, :, None]
        np.testing.assert_allclose(M.cpu().numpy(), _golden(deg, "g_sh") * live.cpu().numpy()[:, :, None], rtol=3e-4,
                                   atol=3e-5)


def test_any_other_use_of_the_handle_is_an_ordinary_tensor():
    from sfgs import sh as sfsh
    gen = torch.Generator().manual_seed(3)
    sh = torch.randn(500, 3, 16, generator=gen).to(DEV).requires_grad_(True)
    dirs = torch.nn.functional.normalize(torch.randn(500, 3, generator=gen), dim=1).to(DEV)
    plain = sfsh.eval_sh(2, sh, dirs)
    h = sfsh.eval_sh_deferred(2, sh, dirs)
    assert isinstance(h, sfsh.DeferredColor) and h.shape == (500, 3) and h.dtype == torch.float32 and h.requires_grad
    assert h.device == sh.device and h.dim() == 2 and len(h) == 500
    torch.testing.assert_close(h * 2.0, plain * 2.0, rtol=0, atol=0)                       # other arithmetic
    torch.testing.assert_close(sfsh.eval_sh_deferred(2, sh, dirs)[10:20], plain[10:20], rtol=0, atol=0)   # indexing
    h2 = torch.clamp_min(sfsh.eval_sh_deferred(2, sh, dirs) + 0.25, 0.0)                    # another constant: recorded, not foldable
    assert isinstance(h2, sfsh.DeferredColor) and h2.folded_inputs() is None
    torch.testing.assert_close(h2.sum(), torch.clamp_min(plain + 0.25, 0.0).sum(), rtol=0, atol=0)
    h3 = torch.clamp_min(sfsh.eval_sh_deferred(2, sh, dirs) + 0.5, 0.0)
    assert h3.folded_inputs() is not None
    g1, = torch.autograd.grad(h3.sum(), sh)                                                   # materialises, with its graph
    g2, = torch.autograd.grad(torch.clamp_min(plain + 0.5, 0.0).sum(), sh)
    torch.testing.assert_close(g1, g2, rtol=0, atol=0)
    assert h3.folded_inputs() is None                                                         # ... and stays a tensor
    h4 = sfsh.eval_sh_deferred(1, torch.randn(8, 3, 6, device=DEV), dirs[:8])                 # 6 stored coefficients: eval_sh handles it
    assert torch.clamp_min(h4 + 0.5, 0.0).folded_inputs() is None
    with pytest.raises(AssertionError):
        sfsh.eval_sh_deferred(2, sh[:, :, :4], dirs)


@pytest.mark.parametrize("deg,stored", [(4, 25), (3, 25), (3, 16)])
def test_in_kernel_sh_path_equals_the_python_colour_path(deg, stored):
    """SURVEY 8c item 2 at the degrees the upstream rasterizer does not have: `shs = pc.get_features` (the rasterizer
    evaluates the SH itself, direction = normalize(xyz - campos)) against `convert_SHs_python` (eval_sh in Python, then
    colors_precomp) -- images and every gradient, xyz through both of its roles. Degree 4 / 25 coefficients: round 4."""
    from diff_gauss import GaussianRasterizer
    from sfgs import sh as sfsh
    W, H, n = 256, 160, 20000
    frame, g = scene(n, W, H, seed=31, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="precomp")
    gen = torch.Generator().manual_seed(77)
    feats0 = torch.randn(n, stored, 3, generator=gen) * 0.3
    feats0[:, 0] += 0.2
    gc, gd = (t.to(DEV) for t in upstream_grads(W, H, 5))
    res = []
    for in_kernel in (True, False):
        xyz = g["means3D"].to(DEV).requires_grad_(True)
        feats = feats0.to(DEV).requires_grad_(True)
        leaves = {k: g[k].to(DEV).requires_grad_(True) for k in ("scales", "rotations", "opacities")}
        means2D = torch.zeros_like(xyz, requires_grad=True)
        if in_kernel:
            kw = dict(shs=feats, colors_precomp=None)
        else:
            shs_view = feats.transpose(1, 2).view(-1, 3, stored)
            dir_pp = xyz - frame["campos"].to(DEV).repeat(n, 1)
            kw = dict(shs=None, colors_precomp=torch.clamp_min(
                sfsh.eval_sh(deg, shs_view, dir_pp / dir_pp.norm(dim=1, keepdim=True)) + 0.5, 0.0))
        image, depth, _, alpha, radii, _ = GaussianRasterizer(_settings(frame, deg))(
            means3D=xyz, means2D=means2D, opacities=leaves["opacities"], scales=leaves["scales"],
            rotations=leaves["rotations"], cov3Ds_precomp=None, **kw)
        torch.autograd.backward([image, depth], [gc, gd])
        res.append(dict(image=image, depth=depth, alpha=alpha, radii=radii, g_feats=feats.grad, g_xyz=xyz.grad,
                        g_means2D=means2D.grad, **{"g_" + k: v.grad for k, v in leaves.items()}))
    a, b = res
    torch.testing.assert_close(a["radii"], b["radii"], rtol=0, atol=0)
    for k in ("image", "depth", "alpha"):
        _close(a[k], b[k], k, rtol=1e-5, atol_rel=1e-6)
    for k in ("g_feats", "g_xyz", "g_means2D", "g_scales", "g_rotations", "g_opacities"):
        _close(a[k], b[k], k, rtol=5e-5, atol_rel=5e-6)
    M = (deg + 1) ** 2
    assert float(a["g_feats"][:, M:].abs().max() if stored > M else 0.0) == 0.0
