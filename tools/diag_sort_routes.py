#!/usr/bin/env python
"""Which sort route wins on the list-heavy regimes (design tool; GPU box): per regime the hint state the wrapper learnt (long tiles,
longest list, tiles over 512) and the step time / sort-kernel time under every forced route (sfgs_set_option "sort")."""
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
import diff_gauss  # noqa: E402
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from sfgs import _lib as L  # noqa: E402
from sfgs.synth import city_scene, orbit_scene, scene, upstream_grads  # noqa: E402

dev = torch.device("cuda:0")
W, H, N = 1920, 1080, 2_000_000
REG = {"city_e25": lambda: city_scene(N, W, H, 25.0, seed=0), "city_e45": lambda: city_scene(N, W, H, 45.0, seed=0),
       "orbit_e25": lambda: orbit_scene(N, W, H, 25.0, seed=0),
       "low_elevation": lambda: scene(N, W, H, seed=0, pitch_deg=45.0, zrange=(40.0, 400.0))}
gc, gd = (t.to(dev) for t in upstream_grads(W, H, 0))
for name in sys.argv[1:] or list(REG):
    frame, g = REG[name]()
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"], kernel_size=frame["kernel_size"],
        subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0, viewmatrix=frame["view"].to(dev),
        projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros(N, 3, device=dev, requires_grad=True)

    def step():
        for v in list(t.values()) + [m2]:
            v.grad = None
        c, d, *_ = rast(means3D=t["means3D"], means2D=m2, colors_precomp=t["colors_precomp"], opacities=t["opacities"],
                        scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([c, torch.nan_to_num(d)], [gc, gd])
    row = {"regime": name}
    for route in ("auto", "fused", "fused1024", "split"):
        L.set_option("sort", route)
        for _ in range(12):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        L.profile_enable(True)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        prof = L.profile_collect(); L.profile_enable(False)
        sort_ms = sum(v[0] for k, v in prof.items() if k.startswith(("sort_", "fine_bin"))) / 3
        row[route] = {"ms": round(ms, 3), "sort_ms": round(sort_ms, 3), "fwd_hints": diff_gauss.last_counters()["fwd_hints"]}
        if route == "auto":
            hs = list(diff_gauss._hint_state.values())[0]
            row["hints"] = {k: int(hs[k]) for k in ("long", "maxlist", "over512", "cmax")}
    L.set_option("sort", "auto")
    print(json.dumps(row), flush=True)
    del t, m2, rast
    torch.cuda.empty_cache()
