#!/usr/bin/env python
"""Timings of the 8f "next row" ops against the torch code the reference runs, same GPU, headline scene size:
eval_sh fwd+bwd (utils/sh_utils.py:57-112 restated inline below for timing only) and prune_points
(scene/gaussian_model.py:563-603: 26 boolean-index operations)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "skyfall-gs_amd"))
from sfgs import compact, densify, sh  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2_000_000)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
N = a.n


def timed(fn, iters=a.iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


C0, C1 = 0.28209479177387814, 0.4886025119029199


def torch_eval_sh_deg1(shc, dirs):  # the reference's degree-1 expression, op for op
    result = C0 * shc[..., 0]
    x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
    return result - C1 * y * shc[..., 1] + C1 * z * shc[..., 2] - C1 * x * shc[..., 3]


out = {"n": N}
g = torch.Generator().manual_seed(0)
shc = torch.randn(N, 3, 4, generator=g).to(dev).requires_grad_(True)
d = torch.randn(N, 3, generator=g)
dirs = (d / d.norm(dim=1, keepdim=True)).to(dev).requires_grad_(True)
w = torch.randn(N, 3, generator=g).to(dev)


def run(fn):
    shc.grad = dirs.grad = None
    (fn(shc, dirs) * w).sum().backward()


out["eval_sh_deg1_torch_ms"] = round(timed(lambda: run(torch_eval_sh_deg1)), 4)
out["eval_sh_deg1_fused_ms"] = round(timed(lambda: run(lambda s, dd: sh.eval_sh(1, s, dd))), 4)
out["eval_sh_speedup"] = round(out["eval_sh_deg1_torch_ms"] / out["eval_sh_deg1_fused_ms"], 1)

# prune: the 26 tensors of the reference (7 params + 14 moments + 5 statistics), 10 % removed
shapes = [(3,), (1, 3), (3, 3), (1,), (3,), (4,), (24,)]
tensors = [torch.randn(N, *s, device=dev) for s in shapes for _ in range(3)] + [torch.randn(N, 1, device=dev) for _ in range(4)] \
          + [torch.randn(N, device=dev)]
mask = torch.rand(N, device=dev) < 0.1
valid = ~mask
out["prune_torch_ms"] = round(timed(lambda: [t[valid] for t in tensors], 5), 3)
out["prune_fused_ms"] = round(timed(lambda: compact.compact_rows(valid, tensors), 5), 3)
out["prune_speedup"] = round(out["prune_torch_ms"] / out["prune_fused_ms"], 1)
row_bytes = sum(t.numel() // N * 4 for t in tensors)
out["prune_fused_GBps"] = round((row_bytes * N * 1.9 + 5 * N) / out["prune_fused_ms"] / 1e6, 1)  # read all, write 90 %

# densify_and_prune (scene/gaussian_model.py:603-742): the reference's op sequence restated for timing only (quantile by
# sort, boolean-index / repeat / cat over 7 parameters + 14 Adam moments, two prunes over 26 tensors) vs sfgs.densify
import types


class _M(types.SimpleNamespace):
    get_scaling = property(lambda self: torch.exp(self._scaling))
    get_opacity = property(lambda self: torch.sigmoid(self._opacity))


def make_model():
    g = torch.Generator().manual_seed(1)
    r = lambda *sh_, **k: torch.randn(*sh_, generator=g, **k).to(dev)
    m = _M(appearance_enabled=True, percent_dense=0.01)
    m._xyz, m._features_dc, m._features_rest = r(N, 3), r(N, 1, 3), r(N, 3, 3)
    m._opacity, m._scaling, m._rotation, m._embeddings = r(N, 1) - 1.0, r(N, 3) * 0.7 - 2.0, r(N, 4), r(N, 24)
    names = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
                 rotation="_rotation", embeddings="_embeddings")
    groups = []
    for k, attr in names.items():
        p = torch.nn.Parameter(getattr(m, attr))
        setattr(m, attr, p)
        groups.append(dict(params=[p], lr=1e-3, name=k))
    m.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for grp in groups:
        p = grp["params"][0]
        m.optimizer.state[p] = dict(step=torch.tensor(1.0), exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
    m.denom = torch.randint(1, 4, (N, 1), generator=g).float().to(dev)
    m.xyz_gradient_accum = (torch.rand(N, 1, generator=g) * 3e-4).to(dev) * m.denom
    m.xyz_gradient_accum_abs = (torch.rand(N, 1, generator=g) * 9e-4).to(dev) * m.denom
    m.xyz_gradient_accum_abs_max = torch.zeros(N, 1, device=dev)
    m.max_radii2D = torch.zeros(N, device=dev)
    return m


def torch_densify(m, max_grad=2e-4, min_opacity=0.005, extent=30.0):
    """timing-only restatement of the reference's sequence on the same data"""
    grads = (m.xyz_gradient_accum / m.denom).nan_to_num(0.0)
    grads_abs = (m.xyz_gradient_accum_abs / m.denom).nan_to_num(0.0)
    ratio = (grads.norm(dim=-1) >= max_grad).float().mean()
    Q = torch.quantile(grads_abs.reshape(-1), 1 - ratio)
    sel = (grads.norm(dim=-1) >= max_grad) | (grads_abs.norm(dim=-1) >= Q)
    smax = m.get_scaling.max(dim=1).values
    params = [m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation, m._embeddings]
    moments = [m.optimizer.state[p][k] for p in params for k in ("exp_avg", "exp_avg_sq")]
    clone = sel & (smax <= m.percent_dense * extent)
    new = [p[clone] for p in params]
    params = [torch.cat((p, n_), 0) for p, n_ in zip(params, new)]
    moments = [torch.cat((mm, torch.zeros_like(new[i // 2])), 0) for i, mm in enumerate(moments)]
    n1 = params[0].shape[0]
    pg = torch.zeros(n1, device=dev); pg[:N] = grads.squeeze()
    pa = torch.zeros(n1, device=dev); pa[:N] = grads_abs.squeeze()
    split = ((pg >= max_grad) | (pa >= Q)) & (torch.exp(params[4]).max(dim=1).values > m.percent_dense * extent)
    new = [p[split].repeat(2, *([1] * (p.dim() - 1))) for p in params]
    params = [torch.cat((p, n_), 0) for p, n_ in zip(params, new)]
    moments = [torch.cat((mm, torch.zeros_like(new[i // 2])), 0) for i, mm in enumerate(moments)]
    valid = ~torch.cat((split, torch.zeros(2 * int(split.sum()), device=dev, dtype=torch.bool)))
    stats = [torch.zeros(params[0].shape[0], 1, device=dev) for _ in range(4)] + [torch.zeros(params[0].shape[0], device=dev)]
    params, moments, stats = [t[valid] for t in params], [t[valid] for t in moments], [t[valid] for t in stats]
    prune = (torch.sigmoid(params[3]) < min_opacity).squeeze() | (torch.exp(params[4]).max(dim=1).values > 0.1 * extent)
    valid = ~prune
    params, moments, stats = [t[valid] for t in params], [t[valid] for t in moments], [t[valid] for t in stats]
    return params[0].shape[0]


import contextlib, io
m0 = make_model()
torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); n_ref = torch_densify(m0); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
out["densify_torch_sequence_ms"] = round(min(ts), 3)
ts = []
for _ in range(3):
    m1 = make_model()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with contextlib.redirect_stdout(io.StringIO()):
        densify.densify_and_prune(m1, 2e-4, 0.005, 30.0, 20)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
out["densify_fused_ms"] = round(min(ts), 3)
out["densify_rows"] = [N, int(m1._xyz.shape[0]), int(n_ref)]
out["densify_speedup"] = round(out["densify_torch_sequence_ms"] / out["densify_fused_ms"], 1)
print(json.dumps(out))
