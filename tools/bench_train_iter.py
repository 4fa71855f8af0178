#!/usr/bin/env python
"""Whole training iteration at the headline size (2 M Gaussians, 1920x1080, SH degree 1 through eval_sh): render ->
L1 + SSIM + depth loss -> backward -> densification statistics -> Adam step, with the reference's torch code around
our rasterizer vs with every sfgs hook installed (fused pre-pass, eval_sh, statistics, Adam). The model / render code
is the restatement used by tests/test_gpu_training_loop.py.

REAL=1 (round 6): the same measurement on the reference's OWN code -- scene.gaussian_model.GaussianModel,
gaussian_renderer.render(), scene.cameras.Camera, utils.loss_utils.l1_loss, imported from /root/reference or the staged archive
(tools/stage_reference.py) through tests/ref_real_driver.py -- with the statements of train.py:176-340 (render, masked L1 +
fused_ssim + Pearson depth loss, backward, max_radii2D / add_densification_stats, optimizer.step): no hook vs every hook."""
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))


def headline_model(rr, GaussianModel, appearance, cams, n):
    """The reference's GaussianModel holding the HEADLINE scene (sfgs.synth.scene defaults: z ~ U(250, 350), scales
    exp(U(ln 0.05, ln 0.6)) -- SURVEY 8d cfg 2), parameters set the way tests/ref_real_driver.py::make_model sets them."""
    from torch import nn
    from sfgs.synth import scene
    _, g = scene(n, rr.W, rr.H, seed=0)
    torch.manual_seed(1234)
    m = GaussianModel(1, appearance_enabled=appearance, appearance_n_fourier_freqs=4, appearance_embedding_dim=32)
    gen = torch.Generator().manual_seed(5)
    c = lambda t: nn.Parameter(t.to("cuda").contiguous())
    m._xyz = c(g["means3D"].clone())
    m._features_dc = c(torch.randn(n, 1, 3, generator=gen) * 0.5)
    m._features_rest = c(torch.randn(n, 3, 3, generator=gen) * 0.1)
    m._opacity = c(torch.log(g["opacities"] / (1 - g["opacities"])))
    m._scaling = c(torch.log(g["scales"]))
    m._rotation = c(g["rotations"].clone() * 1.7)
    if appearance:
        m._embeddings = c(torch.randn(n, 24, generator=gen))
    m.max_radii2D = torch.zeros(n, device="cuda")
    m.spatial_lr_scale = 1.0
    m.oneupSHdegree()
    m.training_setup(rr.training_args(), num_train_cameras=len(cams), from_scratch=True)
    m.compute_3D_filter(cameras=cams)
    return m


def real_reference():
    """The reference's real classes at the headline size; prints one JSON line like the restated variant."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ref_real_driver as rr
    rr.W, rr.H = 1920, 1080
    n = int(os.environ.get("N", 2_000_000))
    ref = rr.locate_reference()
    if ref is None:
        print(json.dumps({"error": "no reference tree and no staged archive (tools/stage_reference.py)"}))
        return
    renderer, Camera, GaussianModel, loss_utils = rr.import_reference(ref)
    import types
    pipe = types.SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)
    bg = torch.zeros(3, device="cuda")
    out = {"N": n, "W": rr.W, "H": rr.H, "code": "the reference's real GaussianModel / render() / Camera", "reference": ref,
           "appearance_enabled": os.environ.get("APPEARANCE", "1") != "0"}
    only = os.environ.get("ONLY", "")
    for fused in (False, True):
        if only and only != ("fused" if fused else "torch"):
            continue
        if fused:
            rr.install_all(GaussianModel, renderer)
        cams = rr.make_cameras(Camera, n=4)
        appearance = os.environ.get("APPEARANCE", "1") != "0"     # --appearance_enabled, as every shipped script trains (run_jax.py:22)
        model = headline_model(rr, GaussianModel, appearance, cams, n)
        k = [0]

        def iteration():
            cam = cams[k[0] % len(cams)]            # a different camera (a fresh settings tuple inside render()) every iteration
            k[0] += 1
            pkg = renderer.render(cam, model, pipe, bg, kernel_size=0.1, subpixel_offset=None)
            loss = rr.loss_fn(loss_utils, pkg["render"], pkg["render_depth"], cam)
            loss.backward()
            with torch.no_grad():
                vis, radii = pkg["visibility_filter"], pkg["radii"]
                model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis])
                model.add_densification_stats(pkg["viewspace_points"], vis)
                model.optimizer.step()
                model.optimizer.zero_grad(set_to_none=True)
        for _ in range(30):
            iteration()
        torch.cuda.synchronize()
        import diff_gauss
        cnt = diff_gauss.last_counters()
        out["raster_counters"] = {k: cnt[k] for k in ("num_visible", "num_duplicates", "max_tile_list", "plan_attempts", "fwd_hints")}
        t0 = time.perf_counter()
        for _ in range(100):
            iteration()
        torch.cuda.synchronize()
        out["fused_hooks_ms" if fused else "torch_around_ms"] = round((time.perf_counter() - t0) / 100 * 1e3, 3)
        if os.environ.get("PROFILE"):     # where the iteration's GPU time goes: top kernels by device time over 5 iterations
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                for _ in range(5):
                    iteration()
                torch.cuda.synchronize()
            rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:int(os.environ.get('PROFILE_ROWS', '18'))]
            out["top_kernels_fused" if fused else "top_kernels_torch"] = [
                [e.key[:70], round(e.device_time_total / 5 / 1e3, 3), e.count // 5] for e in rows if e.device_time_total > 0]
        if fused:
            rr.uninstall_all(GaussianModel, renderer)
        del model
        torch.cuda.empty_cache()
    if not only:
        out["speedup"] = round(out["torch_around_ms"] / out["fused_hooks_ms"], 2)
    print(json.dumps(out))


if os.environ.get("REAL"):
    real_reference()
    sys.exit(0)
spec = importlib.util.spec_from_file_location("loop", os.path.join(ROOT, "tests", "test_gpu_training_loop.py"))
loop = importlib.util.module_from_spec(spec)
spec.loader.exec_module(loop)
from fused_ssim import fused_ssim  # noqa: E402
from sfgs import adam, compact, densify_stats, prepass, sh  # noqa: E402
from sfgs.synth import scene  # noqa: E402

N, W, H = int(os.environ.get("N", 2_000_000)), 1920, 1080
frame, g = scene(N, W, H, seed=0, mode="sh", sh_degree=1)
# the reference's Camera keeps its matrices on the GPU (scene/cameras.py:62-72): render()'s `.cuda()` calls are no-ops
frame = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in frame.items()}
gen = torch.Generator().manual_seed(5)
filter_3D = torch.exp(torch.randn(N, 1, generator=gen, dtype=torch.float64) * 0.5 - 1.0)
gt = torch.rand(3, H, W, generator=gen).cuda()
bg = torch.zeros(3, device="cuda")
hooks = [(prepass, loop.GaussianModel), (densify_stats, loop.GaussianModel), (adam, loop.GaussianModel),
         (compact, loop.GaussianModel), (sh, loop.renderer)]
out = {"N": N, "W": W, "H": H}
only = os.environ.get("ONLY", "")          # "fused" / "torch": run one variant (for rocprofv3 kernel statistics)
for fused in (False, True):
    if only and only != ("fused" if fused else "torch"):
        continue
    if fused:
        for mod, target in hooks:
            mod.install(target)
    model = loop.GaussianModel(g, filter_3D)
    model.training_setup()

    def iteration():
        pkg = loop.render(frame, model, bg)
        image, depth = pkg["render"], pkg["render_depth"]
        loss = 0.8 * (image - gt).abs().mean() + 0.2 * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))
        loss = loss + 1e-3 * torch.nan_to_num(depth, nan=0.0, posinf=0.0, neginf=0.0).mean()
        loss.backward()
        with torch.no_grad():
            model.add_densification_stats(pkg["viewspace_points"], pkg["visibility_filter"])
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
    for _ in range(30):     # a process's first ~30 iterations run a few per cent below steady state
        iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        iteration()
    torch.cuda.synchronize()
    out["fused_hooks_ms" if fused else "torch_around_ms"] = round((time.perf_counter() - t0) / 100 * 1e3, 3)
    if fused:
        for mod, target in hooks:
            mod.uninstall(target)
    del model
    torch.cuda.empty_cache()
if not only:
    out["speedup"] = round(out["torch_around_ms"] / out["fused_hooks_ms"], 2)
print(json.dumps(out))
