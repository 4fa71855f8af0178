#!/usr/bin/env python
"""Fused pre-pass vs the reference's torch-eager getters on the same GPU (N = 2 M, float64 filter as in training)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
import torch
from sfgs.prepass import fused_activations
from oracle.prepass_torch import prepass_reference   # here: the thing being compared against, run on the GPU

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
