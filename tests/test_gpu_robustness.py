"""Non-finite and degenerate inputs must neither hang nor crash the rasterizer, nor disturb the Gaussians around them
more than the reference's semantics imply (a poisoned splat may poison the pixels it touches, nothing else): NaN / Inf
positions, zero / negative / infinite scales, NaN opacities, zero quaternions, all-invisible frames."""
import numpy as np
import pytest
import torch

from sfgs.synth import scene, upstream_grads
from test_gpu_raster import run_hip

pytestmark = pytest.mark.gpu


def _poison(g, idx):
    g = {k: (v.clone() if v is not None else None) for k, v in g.items()}
    m, s, r, o = g["means3D"], g["scales"], g["rotations"], g["opacities"]
    m[idx[0]] = float("nan"); m[idx[1], 2] = float("inf"); m[idx[2]] = -float("inf")
    s[idx[3]] = 0.0; s[idx[4]] = -1.0; s[idx[5]] = float("inf"); s[idx[6], 1] = float("nan")
    r[idx[7]] = 0.0; r[idx[8]] = float("nan")
    o[idx[9]] = float("nan"); o[idx[10]] = -3.0; o[idx[11]] = float("inf")
    return g


def test_poisoned_gaussians_do_not_hang_or_disturb_the_others():
    frame, g = scene(6000, 256, 160, seed=4, zrange=(4., 9.), scale_range=(0.02, 0.3))
    gc, gd = upstream_grads(256, 160, 1)
    clean = run_hip(frame, g, gc, gd)
    idx = list(range(100, 1300, 100))
    bad = run_hip(frame, _poison(g, idx), gc, torch.zeros_like(gd))     # depth may be NaN where poisoned splats land
    keep = np.ones(6000, bool)
    keep[idx] = False
    np.testing.assert_array_equal(bad["radii"][keep], clean["radii"][keep])
    assert bad["color"].shape == clean["color"].shape
    # splats with NaN / Inf geometry are culled like the reference culls them (no finite radius): they touch no pixel
    for i in idx[:3]:
        assert bad["radii"][i] == 0
    # gradients of the untouched Gaussians stay finite wherever the image they contributed to is finite
    if np.isfinite(bad["color"]).all():
        for k, v in bad["grads"].items():
            assert np.isfinite(v[keep]).all(), k


def test_all_invisible_and_empty_frames():
    frame, g = scene(500, 128, 96, seed=2)
    g["means3D"][:, 2] = -5.0          # everything behind the camera
    gc, gd = upstream_grads(128, 96, 0)
    out = run_hip(frame, g, gc, torch.zeros_like(gd))
    assert (out["radii"] == 0).all() and out["counters"]["num_duplicates"] == 0
    np.testing.assert_array_equal(out["color"], np.zeros_like(out["color"]))   # black background
    for k, v in out["grads"].items():
        assert not np.any(v), k
