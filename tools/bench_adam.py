#!/usr/bin/env python
"""Optimizer step at the headline scene size: torch.optim.Adam as the reference constructs it
(scene/gaussian_model.py:382 -> foreach path), torch's own fused=True variant, and sfgs.adam.FusedAdam.
Algorithmic traffic: 28 B per float32 element (read p,g,m,v; write p,m,v)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "skyfall-gs_amd"))
from sfgs.adam import FusedAdam  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2_000_000)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
# per-Gaussian tensors of the reference at sh_degree 1 with appearance embeddings (scene/gaussian_model.py:357-376)
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (3, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,),
          "embeddings": (24,)}
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3,
       "embeddings": 5e-3}


def build(cls, **kw):
    g = torch.Generator().manual_seed(0)
    groups = [{"params": [torch.nn.Parameter(torch.randn(a.n, *s, generator=g).to(dev))], "lr": LRS[k], "name": k}
              for k, s in SHAPES.items()]
    for grp in groups:
        grp["params"][0].grad = torch.randn_like(grp["params"][0]) * 1e-3
    return cls(groups, lr=0.0, eps=1e-15, **kw)


def time_it(opt):
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


elems = a.n * sum(int(torch.tensor(s).prod()) for s in SHAPES.values())
out = {"n": a.n, "elements": elems, "algorithmic_bytes": 28 * elems}
for name, mk in (("torch_foreach", lambda: build(torch.optim.Adam)), ("torch_fused", lambda: build(torch.optim.Adam, fused=True)),
                 ("sfgs_fused", lambda: build(FusedAdam))):
    ms = time_it(mk())
    out[name + "_ms"] = round(ms, 4)
    out[name + "_GBps"] = round(28 * elems / ms / 1e6, 1)
    torch.cuda.empty_cache()
out["speedup_vs_reference_path"] = round(out["torch_foreach_ms"] / out["sfgs_fused_ms"], 2)
print(json.dumps(out))
