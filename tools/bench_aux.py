#!/usr/bin/env python
"""Timings of the two small operators of the path: fused_ssim forward+backward at the
training image size and simple_knn.distCUDA2 on SfM-sized clouds. Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
import torch  # noqa: E402
from fused_ssim import fused_ssim  # noqa: E402
from simple_knn._C import distCUDA2  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


out = {}
for (H, W) in ((1080, 1920), (1024, 1024), (1440, 2560)):
    a = torch.rand(1, 3, H, W, device=dev, requires_grad=True)
    b = torch.rand(1, 3, H, W, device=dev)

    def step():
        a.grad = None
        (1.0 - fused_ssim(a, b)).backward()
    ms = timeit(step)
    P = 3 * H * W
    out[f"ssim_fwd_bwd_{W}x{H}_ms"] = round(ms, 4)
    # algorithmic bytes: fwd reads 2 planes + writes 3 maps, bwd reads 3 maps + 2 planes, writes 1
    out[f"ssim_fwd_bwd_{W}x{H}_GBps"] = round(11 * P * 4 / (ms * 1e-3) / 1e9, 1)
    with torch.no_grad():
        out[f"ssim_fwd_only_{W}x{H}_ms"] = round(timeit(lambda: fused_ssim(a, b, train=False)), 4)
for n in (10_000, 100_000, 1_000_000):
    pts = torch.randn(n, 3, device=dev)
    out[f"knn_{n}_ms"] = round(timeit(lambda: distCUDA2(pts), iters=3 if n >= 1_000_000 else 10, warm=1), 3)
print(json.dumps(out))
