#!/usr/bin/env python
"""The launch hints are chosen from the PREVIOUS frame's statistics, and training draws a random camera per iteration
(train.py:176-178): does a mixed schedule cost anything against every camera's own steady state? One opaque city (2 M
Gaussians, 1080p), its orbit cameras at 25 / 45 / 60 / 75 / 89 degrees of elevation: each camera alone (its hints settle on
itself), then the five in a seeded random order. Prints ms per fwd+bwd step and the hint words seen. Design tool; GPU box."""
import collections
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer, last_counters  # noqa: E402
from sfgs.synth import city_scene, upstream_grads  # noqa: E402

N, W, H = 2_000_000, 1920, 1080
ELEV = [25.0, 45.0, 60.0, 75.0, 89.0]
dev = torch.device("cuda:0")
frames = {}
g = None
for e in ELEV:
    f, g = city_scene(N, W, H, e, seed=0)     # the same Gaussians, another camera
    frames[e] = GaussianRasterizer(GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=f["tanfovx"], tanfovy=f["tanfovy"], kernel_size=f["kernel_size"],
        subpixel_offset=None, bg=f["bg"].to(dev), scale_modifier=1.0, viewmatrix=f["view"].to(dev), projmatrix=f["proj"].to(dev),
        sh_degree=0, campos=f["campos"].to(dev), prefiltered=False, debug=False))
t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
means2D = torch.zeros(N, 3, device=dev, requires_grad=True)
gc, gd = (x.to(dev) for x in upstream_grads(W, H, 0))


def step(e):
    for v in list(t.values()) + [means2D]:
        v.grad = None
    color, depth, *_ = frames[e](means3D=t["means3D"], means2D=means2D, colors_precomp=t["colors_precomp"],
                                 opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    torch.autograd.backward([color, torch.nan_to_num(depth)], [gc, gd])
    return last_counters()["fwd_hints"]


def timed(seq):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hints = collections.Counter(step(e) for e in seq)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(seq) * 1e3, dict(hints)


alone = {}
for e in ELEV:
    for _ in range(12):
        step(e)
    alone[e] = timed([e] * 40)
rng = random.Random(7)
seq = [rng.choice(ELEV) for _ in range(240)]
for e in seq[:40]:
    step(e)
mixed = timed(seq[40:])
ideal = sum(alone[e][0] for e in seq[40:]) / len(seq[40:])
print(json.dumps({"alone_ms": {str(e): round(alone[e][0], 4) for e in ELEV}, "alone_hints": {str(e): alone[e][1] for e in ELEV},
                  "mixed_ms": round(mixed[0], 4), "mixed_hints": mixed[1], "ideal_mixed_ms": round(ideal, 4),
                  "mixed_over_ideal": round(mixed[0] / ideal, 4)}))
