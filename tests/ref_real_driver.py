#!/usr/bin/env python
"""ref_real_driver.py -- the reference's REAL code on the HIP path, on a GPU (VERDICT r4 "Next round" item 2).

What executes is the reference's own Python, imported from /root/reference when it is there (authoring container) or
from the staged archive tests/_refstage/skyfall_ref.zip (tools/stage_reference.py; a git-ignored build artefact that
travels to the GPU box with the snapshot, like the built libraries):

  gaussian_renderer.render()             gaussian_renderer/__init__.py:19-164
  scene.gaussian_model.GaussianModel     getters :203-249, EmbeddingModel :44-69, compute_3D_filter :255-308,
                                         training_setup :350-392, add_densification_stats :744-749,
                                         densify_and_prune :707-742, reset_opacity, optimizer surgery :564-651
  scene.cameras.Camera, utils.loss_utils.l1_loss, utils.sh_utils.eval_sh

on real `cuda` tensors -- nothing is redirected -- with `diff_gauss` / `fused_ssim` / `simple_knn` resolving to this
repo's packages (libsfgs.so). Two modes, each a process of its own because the hooks patch classes process-wide:

  --mode render   every colour path of render() (appearance MLP -> eval_sh, in-kernel SH, convert_SHs_python,
                  override_color) x subpixel_offset on / off, each rendered + differentiated THREE ways on the same
                  parameters: (1) HIP library, no hook installed; (2) HIP library, EVERY sfgs hook installed on the real
                  classes (the storage-less handles travel through the reference's own statements into the raw-parameter /
                  split-SH / eval_sh-folded / directions-from-centres kernels); (3) the C oracle behind the same
                  validation layer (tests/oracle_backend.py, hooks off: the reference's plain torch code around it).
                  (1) and (2) are compared with (3): radii bit for bit, images and EVERY parameter gradient the loss
                  reaches (Gaussian parameters, per-Gaussian embeddings, appearance embedding, MLP weights,
                  viewspace_points.grad) within the parity bars of tests/parity.py.
  --mode train    a training()-shaped loop (train.py:176-340: random camera, render, L1 + fused_ssim + depth loss,
                  backward, max_radii2D / add_densification_stats, densify_and_prune + compute_3D_filter on schedule,
                  reset_opacity, optimizer.step) built from the real methods, run with every hook on and with none:
                  same decisions (model sizes after every densification, visible sets, loss curve to 5e-5), parameters
                  equal to 2e-5 over the first iterations and boundedly apart after 50 (see mode_train for why).

Prints one JSON line per case and `REF-REAL OK <n>`; exits non-zero on the first failure.
Test infrastructure: only tests/ runs it (tests/test_gpu_reference_real.py).
"""
import argparse
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "skyfall-gs_amd")
STAGE = os.path.join(HERE, "_refstage", "skyfall_ref.zip")
for p in (HERE, ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

W, H, N = 256, 160, 20000
KERNEL_SIZE = 0.1
DEV = "cuda"


def locate_reference():
    """/root/reference (authoring container), else the staged archive, else None."""
    env = os.environ.get("SFGS_REFERENCE")
    for p in ([env] if env else []) + ["/root/reference"]:
        if p and os.path.isdir(os.path.join(p, "gaussian_renderer")):
            return p
    return STAGE if os.path.isfile(STAGE) else None


def import_reference(path):
    from sfgs import ply as sply
    for name in ("OpenEXR", "Imath", "mediapy"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if "plyfile" not in sys.modules:     # the reference imports plyfile (absent here): sfgs.ply provides the two classes
        m = types.ModuleType("plyfile")
        m.PlyData, m.PlyElement = sply.PlyData, sply.PlyElement
        sys.modules["plyfile"] = m
    sys.path.insert(0, path)
    import gaussian_renderer                       # executes `from diff_gauss import ...` -> this repo's package
    from scene.cameras import Camera
    from scene.gaussian_model import GaussianModel
    from utils import loss_utils
    import diff_gauss
    assert gaussian_renderer.GaussianRasterizer is diff_gauss.GaussianRasterizer
    assert gaussian_renderer.__file__.startswith(path) and GaussianModel.__module__ == "scene.gaussian_model"
    assert os.path.abspath(diff_gauss.__file__).startswith(PKG)
    return gaussian_renderer, Camera, GaussianModel, loss_utils


def training_args():
    return types.SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                                 position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                                 opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001, embedding_lr=0.005,
                                 appearance_embedding_lr=0.001, appearance_embedding_regularization=0.01,
                                 appearance_mlp_lr=0.0005, idu_position_lr_max_steps=10000)


def make_cameras(Camera, n=4):
    from sfgs.camera import fovy_from_fovx
    fovx = math.radians(60.0)
    fovy = fovy_from_fovx(fovx, W, H)
    cams = []
    g = torch.Generator().manual_seed(77)
    for uid in range(n):
        a = 0.03 * uid
        R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        T = np.array([0.15 * uid, -0.1 * uid, 0.05 * uid])
        img = torch.rand(3, H, W, generator=g)
        depth = 4.0 + 4.0 * torch.rand(1, H, W, generator=g)
        cx, cy = (0.0, 0.0) if uid != 1 else (0.02, -0.015)   # a principal-point offset (scene/cameras.py:65-72)
        cams.append(Camera(colmap_id=uid, R=R, T=T, FoVx=fovx, FoVy=fovy, cx=cx, cy=cy, image=img, gt_alpha_mask=None,
                           image_name=f"synthetic_{uid}", uid=uid, depth=depth, data_device=DEV))
    return cams


def make_model(GaussianModel, appearance, cams, seed, n=None, scale_range=None):
    n = N if n is None else n
    scale_range = scale_range or ((0.01, 0.15) if W < 1000 else (0.004, 0.05))
    from torch import nn
    from sfgs.synth import scene
    _, g = scene(n, W, H, seed=seed, zrange=(4.0, 8.0), scale_range=scale_range, xy_fill=1.05)
    torch.manual_seed(1234 + seed)                         # EmbeddingModel init / appearance_embeddings.normal_
    m = GaussianModel(1, appearance_enabled=appearance, appearance_n_fourier_freqs=4, appearance_embedding_dim=32)
    gen = torch.Generator().manual_seed(5 + seed)
    c = lambda t: nn.Parameter(t.to(DEV).contiguous())
    m._xyz = c(g["means3D"].clone())
    m._features_dc = c(torch.randn(n, 1, 3, generator=gen) * 0.5)
    m._features_rest = c(torch.randn(n, 3, 3, generator=gen) * 0.1)
    m._opacity = c(torch.log(g["opacities"] / (1 - g["opacities"])))     # inverse_sigmoid
    m._scaling = c(torch.log(g["scales"]))
    m._rotation = c(g["rotations"].clone() * 1.7)                        # NOT unit: get_rotation normalises
    if appearance:
        m._embeddings = c(torch.randn(n, 24, generator=gen))
    m.max_radii2D = torch.zeros(n, device=DEV)
    m.spatial_lr_scale = 1.0
    m.oneupSHdegree()                                      # active degree 1 = --sh_degree 1 of every script
    m.training_setup(training_args(), num_train_cameras=len(cams), from_scratch=True)
    m.compute_3D_filter(cameras=cams)                      # float64 filter_3D, as in training
    return m


def loss_fn(loss_utils, image, depth, cam, lambda_dssim=0.2, lambda_depth=0.5):
    """train.py:205-232 (mask, L1 + fused_ssim, NaN scrubbing, depth term); the Pearson loss of train.py:970-973 spelled
    with torch (torchmetrics is absent from the image)."""
    from fused_ssim import fused_ssim
    mask = cam.original_mask.cuda()
    gt_image = mask * cam.original_image.cuda()
    gt_depth = mask * cam.original_depth.cuda()
    image = mask * image
    depth = mask * depth
    Ll1 = loss_utils.l1_loss(image, gt_image)
    ssim_value = fused_ssim(image.unsqueeze(0), gt_image.unsqueeze(0))
    loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim_value)
    gt_depth = gt_depth.reshape(-1, 1)
    depth = depth.reshape(-1, 1)
    nan_inf_mask = torch.isnan(depth) | torch.isinf(depth) | torch.isnan(gt_depth) | torch.isinf(gt_depth)
    depth[nan_inf_mask] = 0.0
    gt_depth[nan_inf_mask] = 0.0
    a, b = gt_depth - gt_depth.mean(), depth - depth.mean()
    pearson = (a * b).sum() / (a.norm() * b.norm())
    return loss + lambda_depth * (1 - pearson)


# ---- hooks -------------------------------------------------------------------------------------------------------------
def hook_modules():
    from sfgs import adam, appearance, compact, densify, densify_stats, filter3d, max_radii, prepass, sh
    return prepass, filter3d, densify_stats, adam, compact, densify, sh, max_radii, appearance


def install_all(GaussianModel, renderer):
    prepass, filter3d, densify_stats, adam, compact, densify, sh, max_radii, appearance = hook_modules()
    prepass.install(GaussianModel, fold=True)              # + sfgs.features (get_features) and sfgs.viewdirs (get_xyz)
    for mod in (filter3d, densify_stats, adam, compact, max_radii, appearance):
        mod.install(GaussianModel)
    densify.install(GaussianModel)
    sh.install(renderer, fold=True)


def uninstall_all(GaussianModel, renderer):
    prepass, filter3d, densify_stats, adam, compact, densify, sh, max_radii, appearance = hook_modules()
    sh.uninstall(renderer)
    densify.uninstall(GaussianModel)
    for mod in (appearance, max_radii, compact, adam, densify_stats, filter3d):
        mod.uninstall(GaussianModel)
    prepass.uninstall(GaussianModel)


class Spy:
    """Records, WITHOUT touching them (a handle that something looked into takes the ordinary route), the arguments
    render() hands GaussianRasterizer.forward and the route the call then takes beneath the validation layer."""

    def __enter__(self):
        import diff_gauss
        self.dg, self.calls, self.routes = diff_gauss, [], []
        self.orig_fwd = diff_gauss.GaussianRasterizer.forward
        spy = self

        def forward(mod, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                    cov3Ds_precomp=None):
            spy.calls.append(dict(means3D=means3D, means2D=means2D, opacities=opacities, shs=shs,
                                  colors_precomp=colors_precomp, scales=scales, rotations=rotations))
            return spy.orig_fwd(mod, means3D, means2D, opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
                                rotations=rotations, cov3Ds_precomp=cov3Ds_precomp)
        diff_gauss.GaussianRasterizer.forward = forward
        self.orig_apply = diff_gauss._Rasterize.apply

        def apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, settings, filter_3D=None,
                  sh_dirs=None, sh_degree=None, sh_channel_major=False, shs_rest=None, sh_centers=None):
            spy.routes.append(dict(raw=filter_3D is not None, sh_fold=sh_degree is not None, split=shs_rest is not None,
                                   centers=sh_centers is not None, channel_major=bool(sh_channel_major),
                                   subpixel=settings.subpixel_offset is not None))
            return spy.orig_apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, settings, filter_3D,
                                  sh_dirs, sh_degree, sh_channel_major, shs_rest, sh_centers)
        diff_gauss._Rasterize.apply = apply
        return self

    def __exit__(self, *exc):
        self.dg.GaussianRasterizer.forward = self.orig_fwd
        self.dg._Rasterize.apply = self.orig_apply


# ---- mode render -------------------------------------------------------------------------------------------------------
CASES = [dict(name="A_mlp", colour="A", jitter=False, cam=0, bg=0.0),
         dict(name="A_mlp_jitter_cxcy", colour="A", jitter=True, cam=1, bg=0.0),
         dict(name="B_sh_kernel", colour="B", jitter=False, cam=0, bg=1.0),
         dict(name="B_sh_kernel_jitter", colour="B", jitter=True, cam=2, bg=0.0),
         dict(name="B_sh_python", colour="Bpy", jitter=False, cam=0, bg=1.0),
         dict(name="B_sh_python_jitter", colour="Bpy", jitter=True, cam=3, bg=0.0),
         dict(name="C_override", colour="C", jitter=False, cam=2, bg=0.0),
         dict(name="C_override_jitter_scaled", colour="C", jitter=True, cam=1, bg=0.0, scaling_modifier=0.8)]


def one_pass(ref, case, route):
    """route: 'hip' | 'hip+hooks' | 'oracle'. Returns dict of numpy arrays (outputs, every gradient) + bookkeeping."""
    renderer, Camera, GaussianModel, loss_utils = ref
    import diff_gauss
    import oracle_backend as ob
    cams = make_cameras(Camera)
    hooks = route == "hip+hooks"
    if hooks:
        install_all(GaussianModel, renderer)
    saved = diff_gauss._backend
    if route == "oracle":
        diff_gauss._backend = ob.OracleBackendAnyDevice
    seen = []
    try:
        model = make_model(GaussianModel, case["colour"] == "A", cams, seed=3 if case["colour"] == "A" else 4)
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=case["colour"] == "Bpy")
        gj = torch.Generator().manual_seed(99)
        kw = {}
        if case["jitter"]:
            kw["subpixel_offset"] = (torch.rand((H, W, 2), generator=gj) - 0.5).to(DEV)            # train.py:190
        if case["colour"] == "C":
            oc = torch.rand(3, N, generator=gj).to(DEV).t()                                            # non-contiguous
            assert not oc.is_contiguous()
            kw["override_color"] = oc
        if "scaling_modifier" in case:
            kw["scaling_modifier"] = case["scaling_modifier"]
        bg = torch.full((3,), case["bg"], device=DEV)
        cam = cams[case["cam"]]
        with Spy() as spy:
            pkg = renderer.render(cam, model, pipe, bg, kernel_size=KERNEL_SIZE, **kw)
        assert len(spy.calls) == 1
        a = spy.calls[0]
        route = spy.routes[0] if spy.routes else None
        if hooks:   # the handles really travelled through the reference's statements into the rasterizer ...
            from sfgs import features, prepass, sh, viewdirs
            assert all(isinstance(a[k], prepass.Deferred) for k in ("scales", "opacities", "rotations")), case["name"]
            assert isinstance(a["means3D"], viewdirs.LazyDirs), case["name"]
            if case["colour"] == "B":
                assert isinstance(a["shs"], features.DeferredFeatures), case["name"]
            if case["colour"] in ("A", "Bpy"):
                assert isinstance(a["colors_precomp"], sh.DeferredColor), case["name"]
            seen = sorted(k for k, v in a.items() if v is not None and type(v) is not torch.Tensor and
                          type(v) is not torch.nn.Parameter)
            # ... and the call took the folded kernels, not a materialising fall-back
            want = dict(raw=True, sh_fold=case["colour"] in ("A", "Bpy"), split=case["colour"] in ("B", "Bpy"),
                        centers=case["colour"] in ("A", "Bpy"), subpixel=True)
            assert route is not None and all(route[k] == v for k, v in want.items()), (case["name"], route, want)
        elif route is not None:
            assert all(v is None or type(v) in (torch.Tensor, torch.nn.Parameter) for v in a.values())
            assert not (route["raw"] or route["sh_fold"] or route["split"] or route["centers"]), (case["name"], route)
        loss = loss_fn(loss_utils, pkg["render"], pkg["render_depth"], cam)
        loss.backward()
        out = dict(render=pkg["render"], depth=pkg["render_depth"], alpha=pkg["render_alpha"], radii=pkg["radii"])
        assert torch.equal(pkg["visibility_filter"], pkg["radii"] > 0) and pkg["extra"] is None
        assert pkg["render"].shape == (3, H, W) and pkg["render_depth"].shape == (1, H, W)
        assert pkg["radii"].dtype == torch.int32 and pkg["radii"].is_cuda
        grads = {"viewspace_points": pkg["viewspace_points"].grad}
        for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_embeddings",
                     "appearance_embeddings"):
            p = getattr(model, name, None)
            if isinstance(p, torch.nn.Parameter) and p.numel():
                grads[name] = p.grad
        mlp = getattr(model, "appearance_mlp", None)
        if case["colour"] == "A" and mlp is not None:
            for k, p in mlp.named_parameters():
                grads["mlp." + k] = p.grad
        res = {"out_" + k: v.detach().cpu().numpy() for k, v in out.items()}
        res.update({"grad_" + k: (None if v is None else v.detach().double().cpu().numpy()) for k, v in grads.items()})
        res["loss"] = float(loss.detach())
        res["handles"] = seen
        res["route"] = route
        return res
    finally:
        diff_gauss._backend = saved
        if hooks:
            uninstall_all(GaussianModel, renderer)


def compare(name, got, ref, hooks=False):
    import parity
    rep = {"case": name}
    if not hooks:
        np.testing.assert_array_equal(got["out_radii"], ref["out_radii"], err_msg=name + ": radii")
    else:
        # With the hooks the rasterizer evaluates exp / sigmoid / normalize of the RAW parameters itself (act_math.h, pinned bit
        # for bit to the getters' CPU values: tests/test_prepass.py); the oracle pass was fed what torch's GPU kernels made of
        # the same parameters. Two correct float32 evaluations of one formula may differ in the last ulp (reduction order of
        # F.normalize's norm, the exponential's last bit), and radius = ceil(3 sigma) then flips by one for a Gaussian that sits
        # on an integer boundary: observed 2 of 500 000 at 1080p, 0 of 20 000 in the small cases. Allowed: a handful, by exactly 1.
        d = got["out_radii"].astype(np.int64) - ref["out_radii"].astype(np.int64)
        nflip = int((d != 0).sum())
        rep["radii_flips"] = nflip
        assert nflip <= max(2, int(2e-5 * d.size)) and (np.abs(d).max() if nflip else 0) <= 1, (name, nflip, np.abs(d).max())
        assert np.array_equal(got["out_radii"] > 0, ref["out_radii"] > 0), name + ": visibility differs"
    for k in ("render", "depth", "alpha"):
        r = parity.assert_image_close(f"{name}:{k}", got["out_" + k], ref["out_" + k], borderline_min=4)
        rep[k] = [r["max_rel"], r["bad"]]
    worst = 0.0
    for k in sorted(ref):
        if not k.startswith("grad_"):
            continue
        assert (got[k] is None) == (ref[k] is None), (name, k, "gradient presence differs")
        if ref[k] is None:
            continue
        if not np.abs(ref[k]).max() > 0:
            assert not np.abs(got[k]).max() > 0, (name, k)
            continue
        r = parity.assert_grad_close(f"{name}:{k}", got[k], ref[k])
        worst = max(worst, r["rel_l2"])
    rep["worst_grad_rel_l2"] = worst
    rep["loss_rel"] = abs(got["loss"] - ref["loss"]) / max(abs(ref["loss"]), 1e-30)
    assert rep["loss_rel"] < 1e-4, rep
    return rep


LARGE_CASES = [dict(name="A_mlp_1080p_500k", colour="A", jitter=False, cam=0, bg=0.0),
               dict(name="B_sh_kernel_1080p_500k_jitter", colour="B", jitter=True, cam=1, bg=0.0)]


def mode_render(ref, only=None, large=False):
    global W, H, N
    if large:       # the training viewport and configs[1]'s starting size: the same comparisons at 1920x1080 / 500 000 Gaussians
        W, H, N = 1920, 1080, 500_000
        os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count() or 1))
    n = 0
    for case in (LARGE_CASES if large else CASES):
        if only and case["name"] not in only:
            continue
        orc = one_pass(ref, case, "oracle")
        for route in ("hip", "hip+hooks"):
            got = one_pass(ref, case, route)
            rep = compare(f"{case['name']}[{route}]", got, orc, hooks=route == "hip+hooks")
            rep["handles"], rep["route"] = got["handles"], got["route"]
            print(json.dumps(rep), flush=True)
            n += 1
    return n


# ---- mode train --------------------------------------------------------------------------------------------------------
CHECKPOINTS = (1, 5, 10, 19, 29)     # before the first densification (20) and before reset_opacity (30)


def train(ref, hooks, iters, seed=11):
    """The statements of train.py:176-340 on the real classes. Returns (state dict of numpy arrays, losses, log)."""
    import random
    renderer, Camera, GaussianModel, loss_utils = ref
    if hooks:
        install_all(GaussianModel, renderer)
    try:
        cams = make_cameras(Camera)
        model = make_model(GaussianModel, True, cams, seed=seed, n=12000, scale_range=(0.01, 0.2))
        if hooks:
            from sfgs import adam
            assert isinstance(model.optimizer, adam.FusedAdam)
        pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, convert_SHs_python=False)
        background = torch.zeros(3, device=DEV)
        random.seed(7)
        torch.manual_seed(4242)
        densify_from, interval, reset_at, extent = 10, 10, 30, 6.0
        losses, log = [], []
        stack = None
        for iteration in range(1, iters + 1):
            model.update_learning_rate(iteration)                                            # train.py:168
            if not stack:
                stack = cams.copy()
            cam = stack.pop(random.randint(0, len(stack) - 1))                               # :176-178
            sub = None
            if iteration % 3 == 0:                                                           # ray_jitter on some iterations
                g = torch.Generator().manual_seed(1000 + iteration)
                sub = (torch.rand((H, W, 2), generator=g) - 0.5).to(DEV)                     # :189-193
            pkg = renderer.render(cam, model, pipe, background, kernel_size=KERNEL_SIZE, subpixel_offset=sub)
            image, depth = pkg["render"], pkg["render_depth"]
            vsp, vis, radii = pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
            loss = loss_fn(loss_utils, image, depth, cam)
            loss.backward()                                                                  # :279
            with torch.no_grad():
                losses.append(loss.item())                                                   # :285
                model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis])       # :314
                model.add_densification_stats(vsp, vis)                                      # :315
                if iteration > densify_from and iteration % interval == 0 and iteration < iters:   # :317-322 (not on the last one)
                    n0 = model.get_xyz.shape[0]
                    torch.manual_seed(9000 + iteration)     # densify_and_split draws its samples from the global generator
                    model.densify_and_prune(0.0002, 0.005, extent, 20)
                    model.compute_3D_filter(cameras=cams)
                    log.append((iteration, n0, int(model.get_xyz.shape[0])))
                if iteration == reset_at:                                                    # :324-328
                    model.reset_opacity()
                model.optimizer.step()                                                       # :339-340
                model.optimizer.zero_grad(set_to_none=True)
                if iteration in CHECKPOINTS:
                    ck = {k: getattr(model, k).detach().double().cpu().numpy() for k in
                          ("_xyz", "_features_dc", "_opacity", "_embeddings")}
                    ck["mlp.2.weight"] = dict(model.appearance_mlp.named_parameters())["mlp.2.weight"].detach().double().cpu().numpy()
                    log.append(("ckpt", iteration, ck))
        st = {k: getattr(model, k).detach().double().cpu().numpy() for k in
              ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "_embeddings",
               "appearance_embeddings", "filter_3D", "xyz_gradient_accum", "xyz_gradient_accum_abs",
               "xyz_gradient_accum_abs_max", "denom", "max_radii2D")}
        for k, p in model.appearance_mlp.named_parameters():
            st["mlp." + k] = p.detach().double().cpu().numpy()
        st["exp_avg_xyz"] = model.optimizer.state[model._xyz]["exp_avg"].double().cpu().numpy()
        st["exp_avg_sq_opacity"] = model.optimizer.state[model._opacity]["exp_avg_sq"].double().cpu().numpy()
        return st, losses, log
    finally:
        if hooks:
            uninstall_all(GaussianModel, renderer)


def mode_train(ref, iters):
    ref_st, ref_losses, ref_log = train(ref, False, iters)
    got_st, got_losses, got_log = train(ref, True, iters)
    # how the two trajectories separate over the iterations: a defect shows at iteration 1, the amplification of float32
    # rounding by Adam's sign-like steps grows with the iteration count
    growth = {}
    for a, b in zip([e for e in ref_log if e[0] == "ckpt"], [e for e in got_log if e[0] == "ckpt"]):
        growth[a[1]] = {k: float(np.abs(a[2][k] - b[2][k]).max() / max(np.abs(a[2][k]).max(), 1e-30)) for k in a[2]}
    ref_log = [e for e in ref_log if e[0] != "ckpt"]
    got_log = [e for e in got_log if e[0] != "ckpt"]
    rep = {"case": "train", "iters": iters, "densify_log": got_log, "final_n": int(got_st["_xyz"].shape[0]),
           "max_rel_diff_by_iteration": growth}
    worst, table = ("", 0.0), {}
    for k in ref_st:
        assert got_st[k].shape == ref_st[k].shape, (k, got_st[k].shape, ref_st[k].shape)
        scale = max(float(np.abs(ref_st[k]).max()), 1e-30)
        d = np.abs(got_st[k] - ref_st[k]).reshape(-1) / scale
        q999 = float(np.quantile(d, 0.999)) if d.size else 0.0
        table[k] = dict(max=float(d.max()) if d.size else 0.0, q999=q999, over=int((d > 3e-4).sum()), n=int(d.size))
        if q999 > worst[1]:
            worst = (k, q999)
    rep["per_tensor"] = table
    print(json.dumps(rep), flush=True)
    # 1. the DECISIONS are the same: every densify_and_prune produced the same model size, every iteration saw the same
    #    set of visible Gaussians (denom), and the loss curve agrees to 5e-5 over the whole run
    assert got_log == ref_log, ("densify_and_prune produced different row counts", got_log, ref_log)
    assert len(ref_log) >= 2 and any(a != b for _, a, b in ref_log), ref_log      # densification really changed the model
    np.testing.assert_allclose(got_losses, ref_losses, rtol=5e-5)
    np.testing.assert_array_equal(got_st["denom"], ref_st["denom"])
    # 2. SHORT horizon (iterations 1 and 5): the fused operators are drop-ins, not approximations -- every checkpointed
    #    tensor agrees to 2e-5 of its range. The per-Gaussian parameters whose gradient can be ARBITRARILY small (`_opacity`,
    #    `_embeddings` of nearly invisible Gaussians) get half of one Adam step instead: Adam with eps = 1e-15
    #    (scene/gaussian_model.py:382) steps lr g / (|g| + eps'), and at |g| ~ 1e-15 a 1e-7-relative difference between two
    #    correct float32 gradients moves the step by several per cent of lr (measured at iteration 1: 7 % / 5 % of a step)
    eps_level = {"_opacity": 0.5 * 0.05, "_embeddings": 0.5 * 0.005}            # half a step of the group's learning rate
    ranges = {k: max(float(np.abs(ref_st[k]).max()), 1e-30) for k in ("_opacity", "_embeddings")}
    for it in (1, 5):
        for k, v in growth[it].items():
            bar = max(2e-5, eps_level[k] / ranges[k]) if k in eps_level else 2e-5
            assert v <= bar, (it, k, v, bar, growth)
    # 3. LONG horizon: the two runs are two trajectories of a sensitive dynamical system (Adam's sign-like steps on the
    #    appearance MLP, whose weights feed every Gaussian's colour): float32 rounding differences grow ~10x per 5 iterations
    #    (measured, `max_rel_diff_by_iteration`: mlp.2.weight 1e-8, 1e-6, 3e-4, 1.4e-3, 5e-3 at iterations 1, 5, 10, 19, 29;
    #    the positions, which the MLP does not touch, stay at 2e-6) -- what two runs of the reference itself with different
    #    reduction orders do. Bounded, not identical: 99.9 % of every tensor within 2 % of its range, positions and 3D filter
    #    within 1e-4, no element further than the steps it had (iters x its group's learning rate), radii statistics equal
    #    but for a handful of Gaussians.
    lr_of = {"_xyz": 0.00016, "_features_dc": 0.0025, "_features_rest": 0.0025 / 20, "_opacity": 0.05, "_scaling": 0.005,
             "_rotation": 0.001, "_embeddings": 0.005, "appearance_embeddings": 0.001}
    for k, t in table.items():
        assert t["q999"] < (1e-4 if k in ("_xyz", "filter_3D") else 2e-2), (k, t)
        lr = lr_of.get(k, 0.0005 if k.startswith("mlp.") else None)
        if lr is not None:
            scale = max(float(np.abs(ref_st[k]).max()), 1e-30)
            assert t["max"] * scale <= 2.0 * iters * lr, (k, t)
    mism = int((got_st["max_radii2D"] != ref_st["max_radii2D"]).sum())
    assert mism <= max(3, int(2e-3 * got_st["max_radii2D"].size)), mism
    rep = {"case": "train-ok", "worst_q999": list(worst), "max_radii2D_mismatches": mism,
           "loss_first_last": [ref_losses[0], ref_losses[-1]]}
    print(json.dumps(rep), flush=True)
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["render", "train"], required=True)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--large", action="store_true", help="mode render: two cases at 1920x1080 with 500 000 Gaussians")
    a = ap.parse_args()
    path = locate_reference()
    if path is None:
        print("REF-REAL SKIP: neither /root/reference nor tests/_refstage/skyfall_ref.zip (tools/stage_reference.py)")
        return 3
    assert torch.cuda.is_available(), "needs a GPU"
    ref = import_reference(path)
    from sfgs import _lib as L
    L.load()
    print(json.dumps({"reference": path, "libsfgs": L.LIB_PATH}), flush=True)
    n = mode_render(ref, a.only, a.large) if a.mode == "render" else mode_train(ref, a.iters)
    print(f"REF-REAL OK {n}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
