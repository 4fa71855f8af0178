"""Known-answer and self-consistency tests of the CPU oracle (SURVEY 8c substitutes for the golden vectors
the reference does not have: analytic KATs, dense float64 autograd restatement, finite differences)."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle.dense_torch import render_dense
from sfgs.camera import fovy_from_fovx, make_frame
from sfgs.synth import scene, upstream_grads


def _frame(W=65, H=49, ks=0.1):
    fovx = math.radians(60)
    return make_frame(np.eye(3), np.zeros(3), fovx, fovy_from_fovx(fovx, W, H), W, H, kernel_size=ks)


def _centered_gaussian(frame, z, s, opacity, rgb, px, py):
    """isotropic Gaussian whose centre projects exactly onto pixel (px, py)"""
    W, H = frame["W"], frame["H"]
    fx = W / (2 * frame["tanfovx"])
    fy = H / (2 * frame["tanfovy"])
    x = (px + 0.5 - W / 2) / fx * z
    y = (py + 0.5 - H / 2) / fy * z
    return dict(means3D=torch.tensor([[x, y, z]], dtype=torch.float32), scales=torch.full((1, 3), s),
                rotations=torch.tensor([[1., 0, 0, 0]]), opacities=torch.tensor([[opacity]]),
                colors_precomp=torch.tensor([rgb], dtype=torch.float32), shs=None)


def test_single_isotropic_gaussian_closed_form():
    fr = _frame()
    z, s, op = 5.0, 0.2, 0.8
    g = _centered_gaussian(fr, z, s, op, [0.9, 0.5, 0.1], 32, 24)
    R = orc.OracleRender(fr, **g)
    fx = fr["W"] / (2 * fr["tanfovx"])
    var0 = (s * fx / z) ** 2
    var = var0 + fr["kernel_size"]
    coef = math.sqrt(var0 * var0 / (var * var))  # sqrt(det0/det1), isotropic at the image centre
    ys, xs = np.mgrid[0:fr["H"], 0:fr["W"]]
    d2 = (xs - 32) ** 2 + (ys - 24) ** 2
    alpha = np.minimum(0.99, op * coef * np.exp(-0.5 * d2 / var))
    alpha[alpha < 1 / 255] = 0
    # radius rule [UPSTREAM]: lambda_max = mid + sqrt(max(0.1, mid^2 - det)); isotropic -> var + sqrt(0.1)
    rad = math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    assert int(R.radii[0]) == rad
    # only compare near the centre: the EWA Jacobian makes off-centre splats slightly anisotropic
    sl = (slice(20, 29), slice(28, 37))
    np.testing.assert_allclose(R.alpha[0][sl], alpha[sl], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(R.color[0][sl], 0.9 * alpha[sl], rtol=2e-3, atol=2e-4)
    assert abs(R.depth[0, 24, 32] - z) < 1e-5
    assert np.isnan(R.depth[0, 0, 0])  # nothing hit -> NaN (normalised depth)


def test_two_stacked_gaussians_composite_front_to_back():
    fr = _frame()
    a = _centered_gaussian(fr, 4.0, 0.3, 0.6, [1.0, 0.0, 0.0], 32, 24)
    b = _centered_gaussian(fr, 6.0, 0.5, 0.7, [0.0, 1.0, 0.0], 32, 24)
    both = {k: (torch.cat([b[k], a[k]]) if a[k] is not None else None) for k in a}  # far one first: order must not matter
    Ra, Rb, R = orc.OracleRender(fr, **a), orc.OracleRender(fr, **b), orc.OracleRender(fr, **both)
    a1, a2 = Ra.alpha[0, 24, 32], Rb.alpha[0, 24, 32]
    assert abs(R.color[0, 24, 32] - a1) < 1e-6                     # red = alpha1
    assert abs(R.color[1, 24, 32] - a2 * (1 - a1)) < 1e-6          # green = alpha2 (1 - alpha1)
    assert abs(R.alpha[0, 24, 32] - (1 - (1 - a1) * (1 - a2))) < 1e-6
    w1, w2 = a1, a2 * (1 - a1)
    assert abs(R.depth[0, 24, 32] - (4 * w1 + 6 * w2) / (w1 + w2)) < 1e-5


@pytest.mark.parametrize("mode", ["precomp", "sh"])
def test_c_oracle_matches_dense_float64_autograd(mode):
    W, H, n = 48, 40, 60
    frame, g = scene(n, W, H, seed=3, zrange=(4., 8.), scale_range=(0.05, 0.4), mode=mode,
                     sh_degree=3 if mode == "sh" else 0, jitter=True)
    R = orc.OracleRender(frame, **g)
    gi = {k: (v.double().requires_grad_(True) if v is not None else None) for k, v in g.items()}
    m2 = torch.zeros(n, 3, dtype=torch.float64, requires_grad=True)
    c, d, a, rad = render_dense(frame, gi["means3D"], gi["scales"], gi["rotations"], gi["opacities"],
                                gi["colors_precomp"], gi["shs"], means2D=m2)
    assert (R.radii == rad.numpy()).all()
    assert np.abs(R.color - c.detach().numpy()).max() < 1e-5
    assert np.abs(R.alpha - a.detach().numpy()).max() < 1e-5
    dn = d.detach().numpy()
    assert (np.isnan(R.depth) == np.isnan(dn)).all()
    fin = np.isfinite(dn)
    assert np.abs(R.depth - dn)[fin].max() < 2e-4
    gc, gd = upstream_grads(W, H, 0)
    gc, gd = gc * W * H, gd * W * H
    loss = (c * gc.double()).sum() + (torch.where(torch.isnan(d), torch.zeros_like(d), d) * gd.double()).sum()
    loss.backward()
    gdm = gd.clone()
    gdm[torch.isnan(d.detach().float())] = 0
    Gr = R.backward(gc, gdm)
    for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp", "shs"):
        if k in Gr:
            ref = gi[k].grad.numpy().reshape(Gr[k].shape)
            assert np.abs(Gr[k] - ref).max() <= 2e-4 * np.abs(ref).max(), k
    ref = m2.grad.numpy()
    assert np.abs(Gr["means2D"][:, :2] - ref[:, :2]).max() <= 2e-4 * np.abs(ref[:, :2]).max()
    assert (Gr["means2D"][:, 2] >= 0).all()  # abs-grad column (scene/gaussian_model.py:744-749)
    # the abs column bounds the signed one
    assert (Gr["means2D"][:, 2] + 1e-12 >= np.linalg.norm(Gr["means2D"][:, :2], axis=1) * (1 - 1e-5)).all()


def test_oracle_backward_against_finite_differences():
    W, H, n = 40, 32, 25
    frame, g = scene(n, W, H, seed=5, zrange=(4., 8.), scale_range=(0.1, 0.5))
    gc, _ = upstream_grads(W, H, 0)
    gc = gc * W * H

    def loss(gg):
        R = orc.OracleRender(frame, **gg)
        return float((R.color.astype(np.float64) * gc.numpy()).sum())

    R = orc.OracleRender(frame, **g)
    Gr = R.backward(gc, None)
    rng = np.random.default_rng(0)
    checked = 0
    # Only the colours are FD-checked: changing geometry or opacity moves pixels across the alpha >= 1/255
    # cut-off (image jumps of ~1/255 whose number grows with the step, i.e. a bias that does not vanish),
    # which a finite difference sees and the analytic gradient -- like the reference's -- does not. Those
    # gradients are checked against float64 autograd (same cut-offs treated as constants) above.
    for key in ("colors_precomp",):
        for _ in range(10):
            i = int(rng.integers(n))
            j = int(rng.integers(g[key].shape[1]))
            h = 2e-3 * max(1.0, abs(float(g[key][i, j])))
            gp = {k: (v.clone() if v is not None else None) for k, v in g.items()}
            gm = {k: (v.clone() if v is not None else None) for k, v in g.items()}
            gp[key][i, j] += h
            gm[key][i, j] -= h
            fd = (loss(gp) - loss(gm)) / (2 * h)
            an = float(Gr[key][i, j])
            if abs(fd) > 1e-3:  # float32 forward: skip gradients below the difference noise
                assert abs(fd - an) <= 0.05 * abs(fd) + 2e-3, (key, i, j, fd, an)
                checked += 1
    assert checked >= 6


def test_knn_oracle_matches_kdtree():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(3000, 3)).astype(np.float32)
    pts[10] = pts[11]  # a duplicate point: distance 0 counts
    got = orc.knn_dist2(pts)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    ref = (d[:, 1:] ** 2).mean(axis=1)
    np.testing.assert_allclose(got, ref, rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(orc.knn_dist2(pts[:2]), [((pts[0] - pts[1]) ** 2).sum()] * 2, rtol=1e-5)


def test_tiled_torch_restatement_matches_the_c_oracle():
    """oracle/tiled_torch.py (bench.py's "pytorch-restatement" CPU baseline) against the C oracle: forward to
    float32 round-off, autograd gradients against the hand-derived backward."""
    import torch
    from oracle import tiled_torch
    from sfgs.synth import scene, upstream_grads
    frame, g = scene(1500, 100, 70, seed=4, zrange=(4., 8.), scale_range=(0.01, 0.25))
    frame = dict(frame, bg=torch.tensor([0.2, 0.5, 0.1]))
    R = orc.OracleRender(frame, **g)
    gc, gd = upstream_grads(100, 70, 3)
    gd = gd.clone()
    gd[torch.from_numpy(np.isnan(R.depth))] = 0
    G = R.backward(gc, gd)
    t = {k: v.clone().requires_grad_(True) for k, v in g.items() if v is not None}
    means2D = torch.zeros(1500, 3, requires_grad=True)
    color, depth, alpha, radii, stats = tiled_torch.render_tiled(frame, t["means3D"], t["scales"], t["rotations"],
                                                                 t["opacities"], colors_precomp=t["colors_precomp"],
                                                                 means2D=means2D, dtype=torch.float64)
    np.testing.assert_array_equal(radii.numpy(), R.radii)
    assert stats["num_duplicates"] == R.num_duplicates and stats["num_visible"] == R.num_visible
    for name, got in (("color", color), ("alpha", alpha), ("depth", depth)):
        ref = getattr(R, name)
        got = got.detach().numpy()
        assert (np.isnan(got) == np.isnan(ref)).all(), name
        m = ~np.isnan(ref)
        assert np.abs(got[m] - ref[m]).max() <= 2e-5 * max(np.abs(ref[m]).max(), 1e-30), name
    ((color * gc.double()).sum() + (torch.nan_to_num(depth) * gd.double()).sum()).backward()
    for k in ("means3D", "scales", "rotations", "opacities", "colors_precomp"):
        got, ref = t[k].grad.numpy(), G[k]
        assert np.linalg.norm(got - ref) <= 5e-4 * np.linalg.norm(ref), k
    got, ref = means2D.grad.numpy()[:, :2], G["means2D"][:, :2]
    assert np.linalg.norm(got - ref) <= 5e-4 * np.linalg.norm(ref)


def test_decision_margins_report_a_pair_on_the_alpha_threshold():
    """orc_decision_margins (test diagnostics): a splat centred on a pixel whose alpha there is 1/255 (1 + 3e-7) is
    reported as a decision within ~3e-7 of its threshold (and power = 0 at the centre); far from every threshold the
    margins are large. tests/test_gpu_raster.py uses them to tell a legitimate float32 flip from a defect."""
    W, H = 32, 24
    frame, g = scene(1, W, H, seed=1, zrange=(5., 5.), scale_range=(0.2, 0.2))
    # pixel centres sit at integer coordinates: mean2D = ((ndc + 1) W - 1) / 2, so ndc = (1 / W, 1 / H) lands on (W / 2, H / 2)
    g["means3D"] = torch.tensor([[5.0 * frame["tanfovx"] / W, 5.0 * frame["tanfovy"] / H, 5.0]])
    R = orc.OracleRender(frame, **g)
    mx, my, op_eff = R.geom()[0][0], R.geom()[0][1], R.geom()[0][5]
    assert abs(mx - round(float(mx))) < 1e-3 and abs(my - round(float(my))) < 1e-3
    far = R.decision_margins()
    assert far["alpha"] > 1e-3
    g2 = dict(g, opacities=g["opacities"] * float((1.0 / 255.0) / op_eff * (1.0 + 3e-7)))
    near = orc.OracleRender(frame, **g2).decision_margins()
    assert near["alpha"] < 2e-6, near
    assert near["power"] < 1e-6
