#!/usr/bin/env python
"""Looks for performance cliffs away from the headline scene: the same rasterizer step on other regimes (pitched
low-elevation camera with long lists, a few screen-filling splats, tiny scenes, 4K, dense 1080p). Prints one JSON line
per regime with ms/step and the per-kernel split (HIP events of the library, all kernels bracketed)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer, collect_full_counters, last_counters  # noqa: E402
from sfgs import _lib as L  # noqa: E402
from sfgs.synth import city_scene, orbit_scene, scene, upstream_grads  # noqa: E402

REGIMES = {
    "headline_2M_1080p": dict(n=2_000_000, W=1920, H=1080, kw={}),
    "low_elevation_2M_1080p": dict(n=2_000_000, W=1920, H=1080, kw=dict(pitch_deg=45.0, zrange=(40.0, 400.0))),
    # the IDU stage's orbit cameras over a city-like slab (train.py:364-420): elevation 45 and 25 degrees
    "orbit_e45_2M_1080p": dict(n=2_000_000, W=1920, H=1080, orbit=45.0),
    "orbit_e25_2M_1080p": dict(n=2_000_000, W=1920, H=1080, orbit=25.0),
    # opaque surfaces (ground + boxes covered with flat, mostly opaque disks): pixels saturate early, a large part of
    # every tile list lies behind the last contributor
    "city_e45_2M_1080p": dict(n=2_000_000, W=1920, H=1080, city=45.0),
    "city_e25_2M_1080p": dict(n=2_000_000, W=1920, H=1080, city=25.0),
    # near-nadir views of the same city (the satellite cameras the first training stage uses): 18 - 33 % of the list entries
    # lie behind their tile's last contributor, the zone where the backward's live-flag choice is decided
    "city_e60_2M_1080p": dict(n=2_000_000, W=1920, H=1080, city=60.0),
    "city_e75_2M_1080p": dict(n=2_000_000, W=1920, H=1080, city=75.0),
    "city_e89_2M_1080p": dict(n=2_000_000, W=1920, H=1080, city=89.0),
    "near_big_splats_200k": dict(n=200_000, W=1920, H=1080, kw=dict(zrange=(3.0, 6.0), scale_range=(0.01, 0.3))),
    "screen_filling_2k": dict(n=2_000, W=1920, H=1080, kw=dict(zrange=(3.0, 6.0), scale_range=(0.5, 3.0), opacity_range=(0.01, 0.05))),
    "tiny_scene_1k": dict(n=1_000, W=1920, H=1080, kw={}),
    "uhd_2M_2160p": dict(n=2_000_000, W=3840, H=2160, kw={}),
    "dense_8M_1080p": dict(n=8_000_000, W=1920, H=1080, kw={}),
    "jittered_2M_1080p": dict(n=2_000_000, W=1920, H=1080, kw=dict(jitter=True)),
    # exactly what the reference's render() passes when ray jitter is off: an all-zero [H,W,2] tensor
    "zero_subpixel_tensor_2M_1080p": dict(n=2_000_000, W=1920, H=1080, kw={}, zero_subpix=True),
}
REGIMES.update(json.loads(os.environ.get("EXTRA_REGIMES", "{}")))   # e.g. {"city_e35_4M": {"n": 4000000, "W": 1920, "H": 1080, "city": 35.0}}
dev = torch.device("cuda:0")
only = sys.argv[1:]
if int(os.environ.get("PIN_CORES", "4")) > 0:   # as bench.py / tools/launch_scenes.py: the host-bound regimes (tiny scene) depend on it
    from sfgs import affinity
    affinity.auto(cores=int(os.environ.get("PIN_CORES", "4")))   # (pinned only while a regime's scene has < 500 k Gaussians)
for kv in filter(None, os.environ.get("SFGS_OPTIONS", "").split(",")):   # route options for A/B runs, e.g. SFGS_OPTIONS=prefill=always
    L.set_option(*kv.split("="))
for name, c in REGIMES.items():
    if only and name not in only:
        continue
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
    collect_full_counters(True); step(); cnt = last_counters(); collect_full_counters(False)
    for _ in range(10):    # the caching allocator settles (the capacity hint of the previous regime shrinks over a few frames)
        step()
    # three timed windows of 10 steps, the median reported: one host hiccup inside a 10-step window (seen twice in round 5:
    # a 1.6 ms step read 6.9, a 5.8 ms one 7.7, same kernel times, not reproducible -- tools/diag_regime_steps.py) would
    # otherwise pass for a cliff; the windows are kept in the record
    windows = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        windows.append((time.perf_counter() - t0) / 10 * 1e3)
    ms = sorted(windows)[1]
    L.profile_enable(True)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof = L.profile_collect(); L.profile_enable(False)
    import diff_gauss as _dg
    hs = next(iter(_dg._hint_state.values()), {})
    print(json.dumps({"regime": name, "ms_per_step": round(ms, 3), "fwd_hints": last_counters().get("fwd_hints"),
                      "tiles_over_512": hs.get("over512"), "long_tiles": hs.get("long"), "max_bin_items": hs.get("cmax"), "windows_ms": [round(w, 3) for w in windows], "N_vis": cnt["num_visible"],
                      "D_binned": cnt["num_duplicates"], "max_tile_list": cnt["max_tile_list"],
                      "kernel_ms": {k: round(v[0] / 3, 3) for k, v in prof.items() if v[1]}}), flush=True)
    del t, means2D, rast
    torch.cuda.empty_cache()
