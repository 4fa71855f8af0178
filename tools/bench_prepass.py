#!/usr/bin/env python
"""Fused pre-pass vs the reference's torch-eager getters on the same GPU (N = 2 M, float64 filter as in training)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
import torch
from sfgs.prepass import fused_activations


def prepass_reference(scaling_raw, opacity_raw, rotation_raw, filter_3D):
    """The reference's three getters as torch eager ops (scene/gaussian_model.py:207-213,216-217,237-249) + render()'s
    .float() casts: the baseline this tool times on the GPU. (Restated here: tools do not import oracle/.)"""
    scales = torch.exp(scaling_raw)
    s_filt = torch.sqrt(torch.square(scales) + torch.square(filter_3D))
    scales_square = torch.square(scales)
    coef = torch.sqrt(scales_square.prod(dim=1) / (scales_square + torch.square(filter_3D)).prod(dim=1))
    o_filt = torch.sigmoid(opacity_raw) * coef[..., None]
    return s_filt.float(), o_filt.float(), torch.nn.functional.normalize(rotation_raw)


dev = torch.device("cuda:0")
N = 2_000_000
g = torch.Generator().manual_seed(0)
a = (torch.randn(N, 3, generator=g) - 2).to(dev).requires_grad_(True)
b = torch.randn(N, 1, generator=g).to(dev).requires_grad_(True)
c = torch.randn(N, 4, generator=g).to(dev).requires_grad_(True)
f = torch.exp(torch.randn(N, 1, generator=g, dtype=torch.float64) - 3).to(dev)
w = [torch.randn(N, k, device=dev) for k in (3, 1, 4)]


def run(fn):
    def step():
        for p in (a, b, c):
            p.grad = None
        o = fn(a, b, c, f)
        torch.autograd.backward(list(o), w)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e3


t_ref, t_fused = run(prepass_reference), run(fused_activations)
# algorithmic bytes: fwd 36+8 read, 32 written; bwd 44 + 32 read, 32 written  (per Gaussian)
print(json.dumps({"N": N, "torch_eager_ms": round(t_ref, 4), "fused_ms": round(t_fused, 4), "speedup": round(t_ref / t_fused, 2),
                  "fused_GBps": round(184 * N / (t_fused * 1e-3) / 1e9, 1)}))

# ---- compute_3D_filter: fused pass vs the reference's per-camera torch loop on the same GPU ----------------
import math
from types import SimpleNamespace
import numpy as np
from sfgs.filter3d import compute_3D_filter
rng = np.random.default_rng(0)
cams = []
for i in range(100):
    q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    cams.append(SimpleNamespace(R=R, T=rng.normal(size=3) * 3, cx=0.0, cy=0.0, image_width=1024, image_height=1024,
                                focal_x=1024 / (2 * math.tan(0.3)), focal_y=1024 / (2 * math.tan(0.3))))
xyz = (torch.randn(N, 3, generator=g) * 8).to(dev)


def reference_loop():  # scene/gaussian_model.py:255-308 as written, on the GPU
    p = xyz.double()
    distance = torch.ones(N, device=dev, dtype=torch.float64) * 1e8
    valid_points = torch.zeros(N, device=dev, dtype=torch.bool)
    focal = 0.0
    for c in cams:
        R = torch.tensor(c.R, device=dev, dtype=torch.float64); T = torch.tensor(c.T, device=dev, dtype=torch.float64)
        pc = p @ R + T[None, :]
        vd = pc[:, 2] > 0.2
        x_, y_, z_ = pc[:, 0], pc[:, 1], torch.clamp(pc[:, 2], min=0.001)
        x_ = x_ / z_ * c.focal_x + c.image_width / 2; y_ = y_ / z_ * c.focal_y + c.image_height / 2
        ins = (x_ >= -0.15 * c.image_width) & (x_ <= c.image_width * 1.15) & (y_ >= -0.15 * c.image_height) & (y_ <= 1.15 * c.image_height)
        v = vd & ins
        distance[v] = torch.min(distance[v], z_[v]); valid_points |= v
        focal = max(focal, c.focal_x)
    distance[~valid_points] = distance[valid_points].max()
    return (distance / focal * (0.2 ** 0.5))[..., None]


def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it):
        r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3, r
t_loop, r1 = t(reference_loop); t_f, r2 = t(lambda: compute_3D_filter(xyz, cams))
print(json.dumps({"filter3d_N": N, "cameras": len(cams), "torch_loop_ms": round(t_loop, 2), "fused_ms": round(t_f, 3),
                  "speedup": round(t_loop / t_f, 1), "max_rel_diff": float(((r1 - r2).abs() / r1.abs()).max())}))

# ---- add_densification_stats: fused kernel vs the reference's boolean-mask torch ops on the same GPU ---------
from sfgs import densify_stats
m = SimpleNamespace(**{k: torch.zeros(N, 1, device=dev) for k in ("xyz_gradient_accum", "xyz_gradient_accum_abs", "xyz_gradient_accum_abs_max", "denom")})
vs = SimpleNamespace(grad=torch.randn(N, 3, device=dev))
filt = torch.rand(N, device=dev) > 0.1


def ref_stats():  # scene/gaussian_model.py:744-749 as written
    m.xyz_gradient_accum[filt] += torch.norm(vs.grad[filt, :2], dim=-1, keepdim=True)
    m.xyz_gradient_accum_abs[filt] += torch.norm(vs.grad[filt, 2:], dim=-1, keepdim=True)
    m.xyz_gradient_accum_abs_max[filt] = torch.max(m.xyz_gradient_accum_abs_max[filt], torch.norm(vs.grad[filt, 2:], dim=-1, keepdim=True))
    m.denom[filt] += 1
t_ref, _ = t(ref_stats, 10); t_f, _ = t(lambda: densify_stats.add_densification_stats(m, vs, filt), 10)
print(json.dumps({"densify_stats_N": N, "torch_masked_ms": round(t_ref, 3), "fused_ms": round(t_f, 4), "speedup": round(t_ref / t_f, 1)}))
