#!/usr/bin/env python
"""Hashes every output of a forward + backward of the library SFGS_LIB points at (default: the shipped one) on a few
seeded scenes, so that an experiment build can be checked for BIT-identical results against the shipped build on the
same box:   python tools/bitcompare.py > a.json;  SFGS_LIB=$PWD/.../lib_x.so python tools/bitcompare.py > b.json;
python tools/bitcompare.py --diff a.json b.json      (design tool; runs on the GPU box)"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "skyfall-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

SCENES = {
    "small_precomp": dict(n=3000, W=200, H=120, kw=dict(zrange=(4., 8.), scale_range=(0.01, 0.2))),
    "sh3_jitter": dict(n=20000, W=320, H=200, kw=dict(zrange=(250., 350.), scale_range=(0.2, 3.0), mode="sh", sh_degree=3,
                                                       jitter=True)),
    "ragged": dict(n=5000, W=130, H=77, kw=dict(zrange=(2., 50.), scale_range=(0.01, 2.0), mode="sh", sh_degree=1)),
    "lists_3k": dict(n=3000, W=24, H=24, kw=dict(zrange=(3., 6.), scale_range=(0.8, 1.5), opacity_range=(0.006, 0.012))),
    "big_splats": dict(n=400, W=256, H=192, kw=dict(zrange=(3., 6.), scale_range=(0.3, 2.0))),
    "headline_2M_1080p": dict(n=2000000, W=1920, H=1080, kw=dict()),
    # (round 6) a pitched camera: splats of 33..96 tiles (the wave-wide walk of preprocess), lists of 513..1 024 entries
    "low_elevation_300k_720p": dict(n=300000, W=1280, H=720, kw=dict(pitch_deg=45.0, zrange=(40.0, 400.0))),
}


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--diff":
        a, b = (json.load(open(f)) for f in sys.argv[2:4])
        bad = [(s, k) for s in a for k in a[s] if a[s][k] != b.get(s, {}).get(k)]
        print("IDENTICAL" if not bad else f"DIFFERENT: {bad}")
        sys.exit(1 if bad else 0)
    import numpy as np
    import torch
    from sfgs.synth import scene, upstream_grads
    from test_gpu_raster import run_hip
    res = {}
    for name, c in SCENES.items():
        frame, g = scene(c["n"], c["W"], c["H"], seed=3, **c["kw"])
        gc, gd = upstream_grads(c["W"], c["H"], 0)
        out = run_hip(frame, g, gc, gd, debug=False)
        h = {k: hashlib.sha256(np.ascontiguousarray(out[k]).tobytes()).hexdigest()[:16] for k in ("color", "depth", "alpha", "radii")}
        for k, v in out["grads"].items():
            h["grad_" + k] = hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()[:16]
        res[name] = h
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
