#!/usr/bin/env python
"""Which of the reference getters' torch GPU kernels differs in the last ulp from the fused activations (act_math.h)?
(round 5: 2 of 500 000 radii differ by one between the hook-less route and the raw-parameter route.)  GPU box only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
import torch  # noqa: E402
from sfgs import prepass  # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
n = 2_000_000
raw_s = (torch.randn(n, 3, generator=g) * 1.5 - 3.0).to(dev)
raw_o = (torch.randn(n, 1, generator=g) * 2.0).to(dev)
raw_q = torch.randn(n, 4, generator=g).to(dev)
out = {}
for fdt in (torch.float32, torch.float64):
    filt = torch.exp(torch.randn(n, 1, generator=g, dtype=torch.float64) * 0.5 - 2.0).to(fdt).to(dev)
    s, o, q = prepass.fused_activations(raw_s, raw_o, raw_q, filt)
    # the reference's statements (scene/gaussian_model.py:207-249) as torch runs them on this device
    scales = torch.exp(raw_s)
    ts = torch.sqrt(torch.square(scales) + torch.square(filt)).float()
    opacity = torch.sigmoid(raw_o)
    sq = torch.square(scales)
    det1 = sq.prod(dim=1)
    det2 = (sq + torch.square(filt)).prod(dim=1)
    to = (opacity * torch.sqrt(det1 / det2)[..., None]).float()
    tq = torch.nn.functional.normalize(raw_q)
    cnt = lambda a, b: int((a != b).sum())
    ulp = lambda a, b: float(((a - b).abs() / b.abs().clamp_min(1e-30)).max())
    # and the pieces
    e_cpu = torch.exp(raw_s.cpu()).to(dev)
    out[str(fdt)] = dict(scales_differ=cnt(s, ts), opacity_differ=cnt(o, to), rotation_differ=cnt(q, tq),
                         scales_max_rel=ulp(s, ts), opacity_max_rel=ulp(o, to), rotation_max_rel=ulp(q, tq),
                         exp_gpu_vs_cpu_differ=cnt(scales, e_cpu), sigmoid_gpu_vs_cpu_differ=cnt(opacity, torch.sigmoid(raw_o.cpu()).to(dev)),
                         normalize_gpu_vs_cpu_differ=cnt(tq, torch.nn.functional.normalize(raw_q.cpu()).to(dev)),
                         prod_gpu_vs_cpu_differ=cnt(det1, sq.cpu().prod(dim=1).to(dev)), elements=n)
print(json.dumps(out))
