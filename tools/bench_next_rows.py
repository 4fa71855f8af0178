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
from sfgs import compact, sh  # noqa: E402

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
print(json.dumps(out))
