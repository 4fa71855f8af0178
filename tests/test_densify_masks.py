"""Densification decisions (clone / split / prune masks) bit-exactly (BASELINE.json north_star).

Golden = masks captured from inside the REAL GaussianModel.densify_and_prune / densify_and_clone / densify_and_split
(scene/gaussian_model.py:653-742) fed with the oracle's gradients through the real add_densification_stats
(tests/golden/make_golden_r2.py). CPU: the restated decision rule (tests/densify_rule.py) reproduces the golden masks
from the golden statistics. GPU: the HIP path's means2D.grad / radii, accumulated by the product's
add_densification_stats kernel, lead to the SAME masks bit for bit."""
import json
import os
import types

import numpy as np
import pytest
import torch

import densify_rule

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_densify.npz")


def _golden():
    z = np.load(GOLD)
    return z, json.loads(str(z["config"]))


def _decide(z, c, accum, accum_abs, denom):
    return densify_rule.decisions(accum.clone(), accum_abs.clone(), denom.clone(), torch.from_numpy(z["scales_in"]),
                                  torch.from_numpy(z["opacities_in"]), float(z["max_grad"]), c["min_opacity"], c["extent"],
                                  c["max_screen_size"], c["percent_dense"])


def test_restated_rule_reproduces_the_real_methods_masks():
    z, c = _golden()
    d = _decide(z, c, torch.from_numpy(z["stats_xyz_gradient_accum"]), torch.from_numpy(z["stats_xyz_gradient_accum_abs"]),
                torch.from_numpy(z["stats_denom"]))
    for k in ("clone", "split", "prune"):
        np.testing.assert_array_equal(d[k].numpy(), z[k])
    assert abs(float(d["Q"]) - float(z["Q"])) == 0.0
    n_clone, n_split, n_pruned = (int(v) for v in z["counts"])
    assert n_clone == int(z["clone"].sum()) > 100 and int(z["split"].sum()) > 1000 and n_pruned == int(z["prune"].sum()) > 1000
    assert n_split == int(z["split"].sum())       # split - clone in densify_and_prune's return value: +2N children, -N parents


@pytest.mark.gpu
def test_hip_gradients_give_the_same_masks_bit_for_bit():
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    from sfgs import densify_stats
    from sfgs.synth import scene, upstream_grads
    z, c = _golden()
    dev = torch.device("cuda:0")
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c["scene_kw"].items()}
    frame, g = scene(c["n"], c["W"], c["H"], seed=int(z["seed"]), **kw)
    scales, opac = torch.from_numpy(z["scales_in"]).to(dev), torch.from_numpy(z["opacities_in"]).to(dev)
    settings = GaussianRasterizationSettings(
        image_height=c["H"], image_width=c["W"], tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0,
        viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev),
        prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    n = c["n"]
    import types
    model = types.SimpleNamespace(**{k: torch.zeros(n, 1, device=dev) for k in
                                     ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom")})
    max_radii = torch.zeros(n, device=dev)
    for k in range(c["frames"]):
        means2D = torch.zeros(n, 3, device=dev, requires_grad=True)
        color, depth, _, _, radii, _ = rast(means3D=g["means3D"].to(dev), means2D=means2D, opacities=opac, scales=scales,
                                            rotations=g["rotations"].to(dev), colors_precomp=g["colors_precomp"].to(dev))
        gc, gd = upstream_grads(c["W"], c["H"], 100 + k)
        gd = gd.to(dev).clone()
        gd[torch.isnan(depth)] = 0
        torch.autograd.backward([color, torch.nan_to_num(depth)], [gc.to(dev), gd])
        vis = radii > 0
        max_radii[vis] = torch.max(max_radii[vis], radii[vis].float())                  # train.py:314
        densify_stats.add_densification_stats(model, means2D, vis)                       # train.py:315
    np.testing.assert_array_equal(radii.cpu().numpy(), z["radii"])
    np.testing.assert_array_equal(max_radii.cpu().numpy(), z["stats_max_radii2D"])
    accum, accum_abs, denom = model.xyz_gradient_accum, model.xyz_gradient_accum_abs, model.denom
    np.testing.assert_array_equal(denom.cpu().numpy(), z["stats_denom"])
    d = _decide(z, c, accum.cpu(), accum_abs.cpu(), denom.cpu())
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    e1 = rel(accum.cpu().numpy(), z["stats_xyz_gradient_accum"])
    e2 = rel(accum_abs.cpu().numpy(), z["stats_xyz_gradient_accum_abs"])
    m1, m2 = (float(v) for v in z["margins"])
    print(f"statistics vs oracle-driven golden: signed {e1:.2e} abs {e2:.2e}; decision margins {m1:.2e} {m2:.2e}; "
          f"Q {float(d['Q']):.6e} vs {float(z['Q']):.6e}")
    for k in ("clone", "split", "prune"):
        np.testing.assert_array_equal(d[k].numpy(), z[k], err_msg=k)
    # the PRODUCT's decision kernel (sfgs.densify / csrc/densify.hip) on the same HIP statistics: same masks, same Q
    from sfgs import densify

    class M(types.SimpleNamespace):
        get_scaling = property(lambda self: self.scaling_act)
        get_opacity = property(lambda self: self.opacity_act)
    pm = M(_xyz=g["means3D"].to(dev), xyz_gradient_accum=accum, xyz_gradient_accum_abs=accum_abs, denom=denom,
           scaling_act=scales, opacity_act=opac, percent_dense=c["percent_dense"])
    clone, split, keep, Q = densify.decide_masks(pm, float(z["max_grad"]), c["min_opacity"], c["extent"], c["max_screen_size"])
    np.testing.assert_array_equal(clone.cpu().numpy(), z["clone"][: c["n"]])
    np.testing.assert_array_equal(split.cpu().numpy(), z["split"][: c["n"]])
    # radix-select quantile == torch.quantile evaluated ON THE DEVICE on the same statistics (the CPU build of torch
    # forms `ratio = mask.float().mean()` as sum / n, the device build as sum * (1 / n): one ulp apart, visible in Q)
    ga = (accum / denom).nan_to_num(0.0)
    gb = (accum_abs / denom).nan_to_num(0.0)
    assert float(Q) == float(densify_rule.quantile_threshold(ga, gb, float(z["max_grad"])))
    # survivors in the reference's order [originals not split | clones | children x 2] = complement of its prune mask
    kept_ref = ~z["prune"]
    k = keep.cpu().numpy()
    n_o, n_c = int((~z["split"][: c["n"]]).sum()), int(z["clone"].sum())
    np.testing.assert_array_equal(k[:, 0][~z["split"][: c["n"]]], kept_ref[:n_o])
    np.testing.assert_array_equal(k[:, 1][z["clone"]], kept_ref[n_o:n_o + n_c])
    ch = k[:, 2][z["split"][: c["n"]]]
    np.testing.assert_array_equal(np.concatenate([ch, ch]), kept_ref[n_o + n_c:])
