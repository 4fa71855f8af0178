#!/usr/bin/env python
"""Wave timeline of a compositing kernel on the headline scene (design tool; runs on the GPU box).
Needs a library built with tools/variants/timeline.patch, which makes every wave store
(entry time, first-batch time, exit time, [nbatch | XCC | HW_ID]) per tile, 100 MHz wall clock:
    tools/build_variant.sh timeline "" tools/variants/timeline.patch
    SFGS_LIB=$PWD/skyfall-gs_amd/sfgs/_exp/lib_timeline.so python tools/timeline.py [--n N] [--steps K]
Prints: kernel span, wave-time integral -> mean resident waves per SIMD, share of wave time spent in the per-tile
prologue, how the kernel's last stretch drains (resident waves over time), per-XCD finish times."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyfall-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--out", default=None)
    ap.add_argument("--kernel", default="bwd", choices=["bwd", "fwd"], help="composite_bwd or composite_fwd<TRAIN> (batches of 64 entries)")
    ap.add_argument("--regime", default="headline", choices=["headline", "city_e25", "city_e45", "city_e60", "city_e75", "city_e89", "orbit_e25", "orbit_e45", "low_elevation"],
                    help="scene / camera (tools/bench_regimes.py's geometry)")
    a = ap.parse_args()
    import numpy as np
    import torch
    from sfgs import _lib as L
    from sfgs.synth import city_scene, orbit_scene, scene, upstream_grads
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
    L.load()
    dev = torch.device("cuda:0")
    W, H, N = a.width, a.height, a.n
    if a.regime == "headline":
        frame, g = scene(N, W, H, seed=0)
    elif a.regime.startswith("city"):
        frame, g = city_scene(N, W, H, float(a.regime[-2:]), seed=0)
    elif a.regime.startswith("orbit"):
        frame, g = orbit_scene(N, W, H, float(a.regime[-2:]), seed=0)
    else:
        frame, g = scene(N, W, H, seed=0, pitch_deg=45.0, zrange=(40.0, 400.0))
    gc, gd = upstream_grads(W, H, 0)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"], kernel_size=frame["kernel_size"],
        subpixel_offset=torch.zeros(H, W, 2, dtype=torch.float32, device=dev), bg=frame["bg"].to(dev), scale_modifier=1.0,
        viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev),
        prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    t = {k: (v.to(dev).requires_grad_(True) if v is not None else None) for k, v in g.items()}
    means2D = torch.zeros(N, 3, device=dev, requires_grad=True)
    gc, gd = gc.to(dev), gd.to(dev)
    for _ in range(a.steps):
        for v in list(t.values()) + [means2D]:
            if v is not None:
                v.grad = None
        color, depth, *_ = rast(means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=t["colors_precomp"],
                                opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([color, torch.nan_to_num(depth)], [gc, gd])
    torch.cuda.synchronize()
    lib = C.CDLL(L.LIB_PATH)
    T8 = ((W + 7) // 8) * ((H + 7) // 8)
    n = min(T8, 65536)
    buf = (C.c_uint64 * (4 * n))()
    rc = (lib.sfgs_dbg_timeline if a.kernel == "bwd" else lib.sfgs_dbg_timeline_fwd)(buf, 4 * n)
    assert rc == 0, rc
    tl = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    ok = tl[:, 2] > 0
    tl = tl[ok]
    t0, t1, t2, meta = tl[:, 0], tl[:, 1], tl[:, 2], tl[:, 3]
    base = t0.min()
    t0, t1, t2 = (t0 - base) / 100.0, (t1 - base) / 100.0, (t2 - base) / 100.0   # microseconds
    nbatch = (meta >> 48) & 0xffff
    xcc = (meta >> 32) & 0xf
    span = t2.max()
    life = t2 - t0
    pro = t1 - t0
    res = {"kernel": a.kernel, "regime": a.regime, "nbatch_mean": round(float(nbatch.mean()), 2), "nbatch_p99": int(np.percentile(nbatch, 99)),
           "nbatch_max": int(nbatch.max()), "tiles_with_work": int(len(tl)), "kernel_span_us": round(float(span), 1),
           "wave_time_integral_us": round(float(life.sum()), 0),
           "mean_resident_waves_per_simd": round(float(life.sum() / span / 1024.0), 3),
           "prologue_share_of_wave_time": round(float(pro.sum() / life.sum()), 4),
           "prologue_us_mean": round(float(pro.mean()), 2), "prologue_us_p90": round(float(np.percentile(pro, 90)), 2),
           "tile_us_mean": round(float(life.mean()), 2), "tile_us_p99": round(float(np.percentile(life, 99)), 2),
           "tile_us_max": round(float(life.max()), 2),
           "us_per_batch_mean": round(float(((t2 - t1) / np.maximum(nbatch, 1)).mean()), 3)}
    # resident waves over time, 5-us buckets
    edges = np.arange(0, span + 5, 5.0)
    occ = np.zeros(len(edges))
    for a_, b_ in zip(t0, t2):
        i0, i1 = int(a_ // 5), int(b_ // 5)
        if i0 == i1:
            occ[i0] += (b_ - a_) / 5
        else:
            occ[i0] += ((i0 + 1) * 5 - a_) / 5
            occ[i0 + 1:i1] += 1
            occ[i1] += (b_ - i1 * 5) / 5
    res["resident_waves_per_simd_over_time_5us"] = [round(float(x / 1024), 2) for x in occ[:-1]]
    full = occ[:-1] / 1024 >= 3.5
    res["time_at_ge_3p5_waves_us"] = float(full.sum() * 5)
    res["xcd_finish_us"] = [round(float(t2[xcc == x].max()), 1) if (xcc == x).any() else None for x in range(8)]
    res["xcd_tiles"] = [int((xcc == x).sum()) for x in range(8)]
    print(json.dumps(res))
    if a.out:
        np.save(a.out, tl)


if __name__ == "__main__":
    main()
