#!/usr/bin/env python
"""Diagnostic: wall time of consecutive groups of 5 rasterizer steps from a cold process (is there a ramp?)."""
import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
from sfgs import _lib as L
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
torch.cuda.synchronize()
out = []
for grp in range(24):
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 5 * 1e3, 3))
print("ms/step per group of 5:", out)
if len(sys.argv) > 1:   # idle gap, then again
    time.sleep(float(sys.argv[1]))
    out = []
    for grp in range(8):
        t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / 5 * 1e3, 3))
    print("after idle:", out)
