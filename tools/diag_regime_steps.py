#!/usr/bin/env python
"""Diagnostic: per-step wall time, plan attempts and planned capacities of the first steps of a regime of
tools/bench_regimes.py, run after another regime as in the measurement set (where a 10-step timed window once read 6.9 ms
for a 1.6 ms step and once 7.7 for 5.8). usage: diag_regime_steps.py <regime before> <regime> [steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
sys.argv, args = sys.argv[:1] + ["__none__"], sys.argv[1:]
import importlib.util  # noqa: E402
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer, last_counters  # noqa: E402
from sfgs.synth import city_scene, orbit_scene, scene, upstream_grads  # noqa: E402

spec = importlib.util.spec_from_file_location("regimes", os.path.join(ROOT, "tools", "bench_regimes.py"))
src = open(os.path.join(ROOT, "tools", "bench_regimes.py")).read()
REGIMES = {}
exec(src[src.index("REGIMES = {"):src.index("dev = torch.device")], {}, REGIMES)
REGIMES = REGIMES["REGIMES"]
dev = torch.device("cuda:0")
steps = int(args[2]) if len(args) > 2 else 40


def make(name):
    c = REGIMES[name]
    if "city" in c:
        frame, g = city_scene(c["n"], c["W"], c["H"], c["city"], seed=0)
    elif "orbit" in c:
        frame, g = orbit_scene(c["n"], c["W"], c["H"], c["orbit"], seed=0)
    else:
        frame, g = scene(c["n"], c["W"], c["H"], seed=0, **c["kw"])
    gc, gd = (t.to(dev) for t in upstream_grads(c["W"], c["H"], 0))
    sub = torch.zeros(c["H"], c["W"], 2) if c.get("zero_subpix") else frame.get("subpix")
    settings = GaussianRasterizationSettings(
        image_height=frame["H"], image_width=frame["W"], tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"], subpixel_offset=None if sub is None else sub.to(dev), bg=frame["bg"].to(dev),
        scale_modifier=1.0, viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0,
        campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
    means2D = torch.zeros(c["n"], 3, device=dev, requires_grad=True)

    def step():
        for v in list(t.values()) + [means2D]:
            v.grad = None
        color, depth, *_ = rast(means3D=t["means3D"], means2D=means2D, colors_precomp=t["colors_precomp"],
                                opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([color, torch.nan_to_num(depth)], [gc, gd])
    return step


before = make(args[0])
for _ in range(24):
    before()
torch.cuda.synchronize()
del before
torch.cuda.empty_cache()
step = make(args[1])
rows = []
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    c = last_counters()
    rows.append(dict(step=i, ms=round((time.perf_counter() - t0) * 1e3, 3), attempts=c["plan_attempts"], cap=c["dup_capacity"],
                     ccap=c["coarse_capacity"], D=c["num_duplicates"], reserved_MB=torch.cuda.memory_reserved() >> 20))
for r in rows:
    print(json.dumps(r))
