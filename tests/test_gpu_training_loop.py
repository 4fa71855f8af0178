"""Integration: a train.py-shaped loop (render -> L1 + SSIM + depth loss -> backward -> densification statistics ->
Adam step -> periodic prune) run twice on the same GPU -- once with the reference's own torch code for everything
AROUND the rasterizer (getters, eval_sh, add_densification_stats, torch.optim.Adam, boolean-index pruning; restated
below from scene/gaussian_model.py and utils/sh_utils.py because /root/reference does not exist on the GPU box), once
with every sfgs `install()` hook active (fused pre-pass, eval_sh, densification statistics, Adam, prune compaction).
Both runs use our rasterizer and fused_ssim. The two parameter trajectories must agree: the fused operators are
drop-ins for the torch code they replace, not approximations of it."""
import types

import numpy as np
import pytest
import torch
from torch import nn

from sfgs.synth import scene

pytestmark = pytest.mark.gpu
C0, C1 = 0.28209479177387814, 0.4886025119029199


def eval_sh(deg, sh, dirs):  # utils/sh_utils.py:57-76, degrees 0 and 1
    result = C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = result - C1 * y * sh[..., 1] + C1 * z * sh[..., 2] - C1 * x * sh[..., 3]
    return result


class GaussianModel:
    """The parts of scene/gaussian_model.py the hooks replace, in the reference's own torch formulation."""
    appearance_enabled = False
    max_sh_degree = 1
    active_sh_degree = 1

    def __init__(self, g, filter_3D):
        dev = "cuda"
        self._xyz = nn.Parameter(g["means3D"].to(dev))
        shs = g["shs"].to(dev)
        self._features_dc = nn.Parameter(shs[:, :1].contiguous())
        self._features_rest = nn.Parameter(shs[:, 1:].contiguous())
        self._opacity = nn.Parameter(torch.logit(g["opacities"].to(dev).clamp(1e-4, 1 - 1e-4)))
        self._scaling = nn.Parameter(torch.log(g["scales"].to(dev)))
        self._rotation = nn.Parameter(g["rotations"].to(dev) * 1.7)  # not unit length: get_rotation normalises
        self.filter_3D = filter_3D.to(dev)
        self.max_radii2D = torch.zeros(self._xyz.shape[0], device=dev)

    get_xyz = property(lambda self: self._xyz)
    get_features = property(lambda self: torch.cat((self._features_dc, self._features_rest), dim=1))
    get_rotation = property(lambda self: torch.nn.functional.normalize(self._rotation))

    @property
    def get_scaling_with_3D_filter(self):     # :207-213
        scales = torch.exp(self._scaling)
        return torch.sqrt(torch.square(scales) + torch.square(self.filter_3D))

    @property
    def get_opacity_with_3D_filter(self):     # :237-249
        opacity = torch.sigmoid(self._opacity)
        scales_square = torch.square(torch.exp(self._scaling))
        det1 = scales_square.prod(dim=1)
        det2 = (scales_square + torch.square(self.filter_3D)).prod(dim=1)
        return opacity * torch.sqrt(det1 / det2)[..., None]

    def training_setup(self, training_args=None):   # :350-382
        n = self._xyz.shape[0]
        for k in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom"):
            setattr(self, k, torch.zeros((n, 1), device="cuda"))
        groups = [{"params": [self._xyz], "lr": 0.00016 * 5.0, "name": "xyz"},
                  {"params": [self._features_dc], "lr": 0.0025, "name": "f_dc"},
                  {"params": [self._features_rest], "lr": 0.0025 / 20.0, "name": "f_rest"},
                  {"params": [self._opacity], "lr": 0.05, "name": "opacity"},
                  {"params": [self._scaling], "lr": 0.005, "name": "scaling"},
                  {"params": [self._rotation], "lr": 0.001, "name": "rotation"}]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):   # :744-749
        g = viewspace_point_tensor.grad
        self.xyz_gradient_accum[update_filter] += torch.norm(g[update_filter, :2], dim=-1, keepdim=True)
        self.xyz_gradient_accum_abs[update_filter] += torch.norm(g[update_filter, 2:], dim=-1, keepdim=True)
        self.xyz_gradient_accum_abs_max[update_filter] = torch.max(self.xyz_gradient_accum_abs_max[update_filter],
                                                                   torch.norm(g[update_filter, 2:], dim=-1, keepdim=True))
        self.denom[update_filter] += 1

    def prune_points(self, mask):   # :563-603
        valid = ~mask
        tensors = {}
        for group in self.optimizer.param_groups:
            stored = self.optimizer.state.get(group["params"][0], None)
            if stored is not None:
                stored["exp_avg"] = stored["exp_avg"][valid]
                stored["exp_avg_sq"] = stored["exp_avg_sq"][valid]
                del self.optimizer.state[group["params"][0]]
                group["params"][0] = nn.Parameter(group["params"][0][valid].requires_grad_(True))
                self.optimizer.state[group["params"][0]] = stored
            else:
                group["params"][0] = nn.Parameter(group["params"][0][valid].requires_grad_(True))
            tensors[group["name"]] = group["params"][0]
        self._xyz, self._features_dc, self._features_rest = tensors["xyz"], tensors["f_dc"], tensors["f_rest"]
        self._opacity, self._scaling, self._rotation = tensors["opacity"], tensors["scaling"], tensors["rotation"]
        for k in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom", "max_radii2D"):
            setattr(self, k, getattr(self, k)[valid])


renderer = types.ModuleType("gaussian_renderer_standin")
renderer.eval_sh = eval_sh


def render(frame, pc, bg):   # gaussian_renderer/__init__.py:19-164 with pipe.convert_SHs_python (:120-125)
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    screenspace_points = torch.zeros_like(pc.get_xyz, requires_grad=True, device="cuda") + 0
    screenspace_points.retain_grad()
    H, W = frame["H"], frame["W"]
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"], kernel_size=0.1,
        subpixel_offset=torch.zeros((H, W, 2), dtype=torch.float32, device="cuda"), bg=bg, scale_modifier=1.0,
        viewmatrix=frame["view"].cuda(), projmatrix=frame["proj"].cuda(), sh_degree=pc.active_sh_degree,
        campos=frame["campos"].cuda(), prefiltered=False, debug=False)
    shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
    dir_pp = pc.get_xyz - frame["campos"].cuda().repeat(pc.get_features.shape[0], 1)
    dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    sh2rgb = renderer.eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)   # the name render() looks up
    colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
    image, depth, norm, alpha, radii, extra = GaussianRasterizer(raster_settings=settings)(
        means3D=pc.get_xyz, means2D=screenspace_points, shs=None, colors_precomp=colors_precomp,
        opacities=pc.get_opacity_with_3D_filter.float(), scales=pc.get_scaling_with_3D_filter.float(),
        rotations=pc.get_rotation, cov3Ds_precomp=None)
    return {"render": image, "render_depth": depth, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii}


def train(fused, iters=14, prune_at=8):
    from fused_ssim import fused_ssim
    from sfgs import adam, compact, densify_stats, prepass, sh
    hooks = [(prepass, GaussianModel), (densify_stats, GaussianModel), (adam, GaussianModel), (compact, GaussianModel),
             (sh, renderer)]
    if fused:
        for mod, target in hooks:
            mod.install(target)
    try:
        W, H, n = 256, 160, 12000
        frame, g = scene(n, W, H, seed=11, zrange=(250., 350.), scale_range=(0.3, 3.0), mode="sh", sh_degree=1)
        gen = torch.Generator().manual_seed(5)
        filter_3D = torch.exp(torch.randn(n, 1, generator=gen, dtype=torch.float64) * 0.5 - 1.0)  # float64 as in training
        model = GaussianModel(g, filter_3D)
        model.training_setup()
        assert isinstance(model.optimizer, adam.FusedAdam) == fused
        gts = [torch.rand(3, H, W, generator=gen).cuda() for _ in range(3)]
        bg = torch.zeros(3, device="cuda")
        losses = []
        for it in range(iters):
            pkg = render(frame, model, bg)
            image, depth, gt = pkg["render"], pkg["render_depth"], gts[it % 3]
            loss = 0.8 * (image - gt).abs().mean() + 0.2 * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))
            loss = loss + 1e-3 * torch.nan_to_num(depth, nan=0.0, posinf=0.0, neginf=0.0).mean()
            loss.backward()
            with torch.no_grad():
                vis = pkg["visibility_filter"]
                model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], pkg["radii"][vis])   # train.py:314
                model.add_densification_stats(pkg["viewspace_points"], vis)
                if it == prune_at:
                    mask = (torch.sigmoid(model._opacity) < 0.25).squeeze()
                    model.prune_points(mask)
                    model.filter_3D = model.filter_3D[~mask]   # train.py recomputes the 3D filter after densification
                model.optimizer.step()
                model.optimizer.zero_grad(set_to_none=True)
            losses.append(float(loss.detach()))
        out = {k: getattr(model, k).detach().double().cpu().numpy() for k in
               ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "xyz_gradient_accum",
                "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom", "max_radii2D")}
        out["exp_avg_xyz"] = model.optimizer.state[model._xyz]["exp_avg"].double().cpu().numpy()
        out["exp_avg_sq_opacity"] = model.optimizer.state[model._opacity]["exp_avg_sq"].double().cpu().numpy()
        return out, losses
    finally:
        if fused:
            for mod, target in hooks:
                mod.uninstall(target)


def test_fused_hooks_reproduce_the_torch_training_trajectory():
    ref, ref_losses = train(fused=False)
    got, got_losses = train(fused=True)
    np.testing.assert_allclose(got_losses, ref_losses, rtol=2e-5)
    assert ref["_xyz"].shape[0] < 12000 and got["_xyz"].shape == ref["_xyz"].shape      # the prune happened, identically
    np.testing.assert_array_equal(got["denom"], ref["denom"])
    np.testing.assert_array_equal(got["max_radii2D"], ref["max_radii2D"])
    for k in ref:
        scale = max(float(np.abs(ref[k]).max()), 1e-30)
        err = float(np.abs(got[k] - ref[k]).max()) / scale
        # parameters have moved by ~ iters * lr; Adam's m / sqrt(v) turns 1e-6-relative gradient differences into
        # at most a few 1e-5 of that movement
        assert err < 2e-4, (k, err)
