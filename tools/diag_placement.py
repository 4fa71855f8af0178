#!/usr/bin/env python
"""Diagnostic: is preprocess's per-process bimodality (0.18 vs 0.23 ms) a property of WHERE a buffer lives? One process,
forward-only frames at the headline size; between measurements either the INPUT tensors or the SCRATCH blob are moved to
fresh device memory (the old copies are kept alive so that new physical pages are used)."""
import json, os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
from sfgs import _lib as L
from sfgs.synth import scene
dev = torch.device("cuda:0")
N, W, H = 2_000_000, 1920, 1080
frame, g = scene(N, W, H, seed=0)
settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
    kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0,
    viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev), prefiltered=False, debug=False)
rast = GaussianRasterizer(settings)
t = {k: v.to(dev) for k, v in g.items() if v is not None}
L.load()

def measure(n=30):
    with torch.no_grad():
        for _ in range(5):
            rast(means3D=t["means3D"], means2D=None, colors_precomp=t["colors_precomp"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.cuda.synchronize()
        L.profile_enable(True)
        for _ in range(n):
            rast(means3D=t["means3D"], means2D=None, colors_precomp=t["colors_precomp"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.cuda.synchronize()
        p = L.profile_collect()
        L.profile_enable(False)
    return {k: round(ms / n, 4) for k, (ms, c) in p.items() if k in ("preprocess", "fine_bin", "composite_fwd")}

out = [("baseline", measure())]
keep = []
for i in range(3):     # inputs to fresh memory
    keep.append(t)
    t = {k: v.clone() for k, v in t.items()}
    out.append((f"inputs moved {i}", measure()))
import diff_gauss
for i in range(4):     # scratch to fresh memory: grab the cached block(s) the frame would reuse
    sizes = [s["total_size"] for s in torch.cuda.memory_snapshot() if s["total_size"] > (64 << 20)]
    free_blocks = [b["size"] for s in torch.cuda.memory_snapshot() for b in s["blocks"] if b["state"] == "inactive" and b["size"] > (64 << 20)]
    for sz in free_blocks:
        keep.append(torch.empty(sz - 1024, dtype=torch.uint8, device=dev))
    out.append((f"scratch moved {i} (held {len(free_blocks)} blocks, {sum(free_blocks) >> 20} MiB)", measure()))
for name, m in out:
    print(name, json.dumps(m))
