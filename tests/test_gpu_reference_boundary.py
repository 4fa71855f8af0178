"""The drop-in claim, end to end: a render()-shaped call sequence -- the exact statements of the reference's
gaussian_renderer/__init__.py:19-164 (settings tuple by keyword, rasterizer call by keyword, result dict,
retain_grad on the dummy means2D) -- runs against OUR diff_gauss, followed by the consumers of its integer
outputs (scene/gaussian_model.py:744-749 add_densification_stats, :731-735 max_radii2D update).
/root/reference does not exist on the GPU box, so the call sequence is restated here rather than imported."""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from sfgs.synth import scene

pytestmark = pytest.mark.gpu


def render_like_reference(frame, pc, bg_color, kernel_size, scaling_modifier=1.0, subpixel_offset=None):
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    screenspace_points = torch.zeros_like(pc["xyz"], dtype=pc["xyz"].dtype, requires_grad=True, device="cuda") + 0
    screenspace_points.retain_grad()
    H, W = frame["H"], frame["W"]
    if subpixel_offset is None:
        subpixel_offset = torch.zeros((H, W, 2), dtype=torch.float32, device="cuda")
    raster_settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"], kernel_size=kernel_size,
        subpixel_offset=subpixel_offset, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=frame["view"].cuda(), projmatrix=frame["proj"].cuda(), sh_degree=pc["active_sh_degree"],
        campos=frame["campos"].cuda(), prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    rendered_image, rendered_depth, rendered_norm, rendered_alpha, radii, extra = rasterizer(
        means3D=pc["xyz"], means2D=screenspace_points, shs=pc["shs"], colors_precomp=None,
        opacities=pc["opacity"].float(), scales=pc["scaling"].float(), rotations=pc["rotation"], cov3Ds_precomp=None)
    return {"render": rendered_image, "render_depth": rendered_depth, "render_norm": rendered_norm,
            "render_alpha": rendered_alpha, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "extra": extra}


def test_render_dict_and_densification_consumers():
    W, H, n = 320, 200, 20000
    frame, g = scene(n, W, H, seed=3, zrange=(250., 350.), scale_range=(0.2, 3.0), mode="sh", sh_degree=1)
    pc = dict(xyz=g["means3D"].cuda().requires_grad_(True), shs=g["shs"].cuda().requires_grad_(True),
              opacity=g["opacities"].cuda().requires_grad_(True), scaling=g["scales"].cuda().requires_grad_(True),
              rotation=g["rotations"].cuda().requires_grad_(True), active_sh_degree=1)
    pkg = render_like_reference(frame, pc, torch.zeros(3, device="cuda"), 0.1)
    image, depth = pkg["render"], pkg["render_depth"]
    assert image.shape == (3, H, W) and depth.shape == (1, H, W) and pkg["render_alpha"].shape == (1, H, W)
    assert pkg["radii"].dtype == torch.int32 and pkg["visibility_filter"].dtype == torch.bool and pkg["extra"] is None
    gt = torch.rand(3, H, W, device="cuda")
    from fused_ssim import fused_ssim
    Ll1 = (image - gt).abs().mean()
    loss = 0.8 * Ll1 + 0.2 * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))  # train.py:221-224
    depth_clean = torch.nan_to_num(depth, nan=0.0, posinf=0.0, neginf=0.0)          # train.py:229-231
    loss = loss + 1e-3 * depth_clean.mean()
    loss.backward()
    vs = pkg["viewspace_points"]
    assert vs.grad is not None and vs.grad.shape == (n, 3)
    # scene/gaussian_model.py:744-749
    update_filter = pkg["visibility_filter"]
    xyz_gradient_accum = torch.zeros(n, 1, device="cuda")
    xyz_gradient_accum_abs = torch.zeros(n, 1, device="cuda")
    xyz_gradient_accum[update_filter] += torch.norm(vs.grad[update_filter, :2], dim=-1, keepdim=True)
    xyz_gradient_accum_abs[update_filter] += torch.norm(vs.grad[update_filter, 2:], dim=-1, keepdim=True)
    assert (xyz_gradient_accum_abs + 1e-12 >= xyz_gradient_accum * (1 - 1e-4)).all()
    assert float(xyz_gradient_accum_abs.max()) > 0  # an unmodified 3DGS backward would leave this column at 0
    # scene/gaussian_model.py:731-735 (max_radii2D is float32, radii int32)
    max_radii2D = torch.zeros(n, device="cuda")
    max_radii2D[update_filter] = torch.max(max_radii2D[update_filter], pkg["radii"][update_filter])
    # integers against the oracle
    R = orc.OracleRender(frame, **g)
    np.testing.assert_array_equal(pkg["radii"].cpu().numpy(), R.radii)
    for p in (pc["xyz"], pc["shs"], pc["opacity"], pc["scaling"], pc["rotation"]):
        assert p.grad is not None and torch.isfinite(p.grad).all()
