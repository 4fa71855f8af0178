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


def test_a_reused_settings_tuple_sees_in_place_edits_of_its_camera_tensors():
    """the wrapper caches the SfgsFrame of a settings tuple that comes back (video / benchmark loops). The
    reference's world_view_transform is a TRANSPOSED view (scene/cameras.py:62), which the wrapper has to copy to make it
    contiguous -- a cached copy would freeze the view matrix while campos / bg alias live memory. Moving the camera by
    editing the tuple's tensors in place must move the picture: same frame as a freshly built tuple."""
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    W, H = 192, 128
    frame, g = scene(3000, W, H, seed=6, zrange=(4., 9.), scale_range=(0.02, 0.3))
    dev = "cuda"
    t = {k: v.to(dev) for k, v in g.items() if v is not None}

    def settings(view_t, proj, campos):
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"], kernel_size=0.1,
            subpixel_offset=None, bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=view_t, projmatrix=proj,
            sh_degree=0, campos=campos, prefiltered=False, debug=False)

    def render(s):
        with torch.no_grad():
            out = GaussianRasterizer(s)(means3D=t["means3D"], means2D=None, opacities=t["opacities"],
                                        colors_precomp=t["colors_precomp"], scales=t["scales"], rotations=t["rotations"])
        return out[0].cpu().numpy()
    base_view = frame["view"].to(dev)                       # stored transposed, as the reference stores it
    store = base_view.t().contiguous()                      # ... so that `.t()` of the storage is NON-contiguous
    view_nc = store.t()
    assert not view_nc.is_contiguous() and torch.equal(view_nc, base_view)
    proj, campos = frame["proj"].to(dev).clone(), frame["campos"].to(dev).clone()
    s = settings(view_nc, proj, campos)
    a0 = render(s)
    a1 = render(s)                                          # the same tuple object again
    np.testing.assert_array_equal(a0, a1)
    # move the camera 0.3 to the right IN PLACE (world -> view translation row of the transposed matrix, the centre, the
    # full projection), keeping the tuple
    shift = torch.tensor([0.3, 0.0, 0.0], device=dev)
    new_view = base_view.clone()
    new_view[3, :3] -= shift
    store.copy_(new_view.t())
    campos += shift
    P = torch.linalg.solve(base_view, frame["proj"].to(dev))     # proj = view @ P  (full_proj_transform, cameras.py:73)
    proj.copy_(new_view @ P)
    moved_same_tuple = render(s)
    moved_fresh = render(settings(new_view.contiguous(), proj.clone(), campos.clone()))
    assert np.abs(moved_fresh - a0).max() > 1e-3            # the picture really changed
    np.testing.assert_array_equal(moved_same_tuple, moved_fresh)
