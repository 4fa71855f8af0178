#!/usr/bin/env python
"""Extended random parity soak: tests/test_gpu_raster.py::test_random_configurations beyond its 40 committed seeds.
usage (GPU box): python tools/soak.py FIRST LAST   -> one line per failure, summary at the end"""
import os
import sys
import traceback

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "skyfall-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_raster as T  # noqa: E402



def classify(i):
    """Why did configuration i fail? Small scenes are put to an independent arbiter, the dense float64 autograd renderer
    (oracle/dense_torch.py): if the C oracle's OWN float32 gradients are as far from the float64 ones as the HIP library's
    are, the configuration is ill-conditioned in float32 (a faint, sub-pixel-sharp splat behind a nearly opaque pixel), not
    a defect of either."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    from oracle.dense_torch import render_dense
    from sfgs.synth import scene, upstream_grads
    c = T._random_config(i)
    if c["n"] > 400 or c["W"] * c["H"] > 20000:
        return "too large for the dense float64 arbiter"
    frame, g = scene(c["n"], c["W"], c["H"], seed=200 + i, **c["kw"])
    frame["bg"] = torch.tensor(c["bg"]); frame["depth_mode"] = c["depth_mode"]
    R = orc.OracleRender(frame, **g)
    gc, gd = upstream_grads(c["W"], c["H"], i)
    gd = gd.clone() * (0.0 if c["zero_depth_grad"] else 1.0)
    gd[torch.from_numpy(np.isnan(R.depth))] = 0
    G = R.backward(gc, gd)
    out = T.run_hip(frame, g, gc, gd, depth_mode=c["depth_mode"], debug=False)
    gi = {k: (v.double().requires_grad_(True) if v is not None else None) for k, v in g.items()}
    col, d, a, rad = render_dense(frame, gi["means3D"], gi["scales"], gi["rotations"], gi["opacities"],
                                  gi["colors_precomp"], gi["shs"])
    loss = (col * gc.double()).sum() + (torch.where(torch.isnan(d), torch.zeros_like(d), d) * gd.double()).sum()
    loss.backward()
    rep = {}
    for k in ("means3D", "scales", "rotations", "opacities"):
        ref = gi[k].grad.numpy().reshape(c["n"], -1)
        e_hip = np.linalg.norm(out["grads"][k].reshape(c["n"], -1) - ref) / np.linalg.norm(ref)
        e_orc = np.linalg.norm(G[k].reshape(c["n"], -1) - ref) / np.linalg.norm(ref)
        rep[k] = dict(hip_vs_f64=float(e_hip), c_oracle_vs_f64=float(e_orc))
    verdict = "float32 conditioning (the C oracle is as far from float64 as the library)" if all(
        v["hip_vs_f64"] <= 3.0 * v["c_oracle_vs_f64"] + 1e-4 for v in rep.values()) else "UNEXPLAINED"
    return dict(verdict=verdict, margins=R.decision_margins(), errors=rep)


first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for i in range(first, last):
    try:
        T.test_random_configurations(i)
    except Exception as e:  # noqa: BLE001
        bad.append(i)
        print(f"FAIL {i}: {T._random_config(i)}\n{traceback.format_exc(limit=2)}", flush=True)
        try:
            print("  classification:", classify(i), flush=True)
        except Exception:  # noqa: BLE001
            print("  classification failed:", traceback.format_exc(limit=1), flush=True)
print(f"soak {first}..{last}: {last - first - len(bad)} passed, {len(bad)} failed {bad}")
