#!/usr/bin/env python
"""Diagnostic: one configuration of tests/test_gpu_raster.py::test_random_configurations (seed i), HIP vs oracle, the
Gaussians whose gradients differ most and the oracle's decision margins.   usage (GPU box): tools/diag_seed.py i [tensor]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in (ROOT, os.path.join(ROOT, "skyfall-gs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import test_gpu_raster as T
from oracle import oracle as orc
from sfgs.synth import scene, upstream_grads

i = int(sys.argv[1]); which = sys.argv[2] if len(sys.argv) > 2 else "means2D"
c = T._random_config(i)
frame, g = scene(c["n"], c["W"], c["H"], seed=200 + i, **c["kw"])
frame["bg"] = torch.tensor(c["bg"]); frame["depth_mode"] = c["depth_mode"]
R = orc.OracleRender(frame, **g)
gc, gd = upstream_grads(c["W"], c["H"], i)
gd = gd.clone() * (0.0 if c["zero_depth_grad"] else 1.0)
gd[torch.from_numpy(np.isnan(R.depth))] = 0
G = R.backward(gc, gd)
out = T.run_hip(frame, g, gc, gd, depth_mode=c["depth_mode"], debug=False)
print("config", c); print("margins", R.decision_margins())
for k in G:
    a, b = out["grads"][k].reshape(c["n"], -1).astype(np.float64), G[k].reshape(c["n"], -1).astype(np.float64)
    print(k, "rel_l2", np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30), "max|b|", np.abs(b).max())
a, b = out["grads"][which].reshape(c["n"], -1).astype(np.float64), G[which].reshape(c["n"], -1).astype(np.float64)
err = np.linalg.norm(a - b, axis=1)
geom = R.geom()
for j in np.argsort(-err)[:5]:
    print("gaussian", j, "err", err[j], "hip", a[j], "oracle", b[j], "radius", R.radii[j], "tiles", R.tiles_touched()[j],
          "geom", np.round(geom[j], 4), "opacity", float(g["opacities"][j]), "scale", g["scales"][j].numpy())
