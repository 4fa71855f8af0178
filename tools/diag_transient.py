#!/usr/bin/env python
"""Diagnostic: per-step GPU time (one event after every step) of the first steps after a full synchronisation, with the
device already warm -- what a short timed region bracketed by synchronisations sees. usage: diag_transient.py [idle_ms]"""
import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
from sfgs.synth import scene, upstream_grads
dev = torch.device("cuda:0")
N, W, H = 2_000_000, 1920, 1080
frame, g = scene(N, W, H, seed=0)
gc, gd = (t.to(dev) for t in upstream_grads(W, H, 0))
settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
    kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0,
    viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev), prefiltered=False, debug=False)
rast = GaussianRasterizer(settings)
t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
m2 = torch.zeros(N, 3, device=dev, requires_grad=True)
def step():
    for v in list(t.values()) + [m2]: v.grad = None
    c, d, *_ = rast(means3D=t["means3D"], means2D=m2, colors_precomp=t["colors_precomp"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    torch.autograd.backward([c, torch.nan_to_num(d)], [gc, gd])
for _ in range(300): step()
torch.cuda.synchronize()
idle = float(sys.argv[1]) * 1e-3 if len(sys.argv) > 1 else 0.0
for trial in range(3):
    time.sleep(idle)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(121)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(120):
        step(); ev[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 120 * 1e3
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(120)]
    grp = [round(sum(d[i:i + 10]) / 10, 4) for i in range(0, 120, 10)]
    print(f"idle {idle*1e3:.0f} ms, trial {trial}: wall {wall:.4f} ms/step; first 10 steps {[round(x, 3) for x in d[:10]]}; means of 10: {grp}")
