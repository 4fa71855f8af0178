#!/usr/bin/env python
"""Whole training iteration at the headline size (2 M Gaussians, 1920x1080, SH degree 1 through eval_sh): render ->
L1 + SSIM + depth loss -> backward -> densification statistics -> Adam step, with the reference's torch code around
our rasterizer vs with every sfgs hook installed (fused pre-pass, eval_sh, statistics, Adam). The model / render code
is the restatement used by tests/test_gpu_training_loop.py."""
import importlib.util
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
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
