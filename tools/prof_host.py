"""Host-side cost of one training step through diff_gauss (round 4): wall time per step without any
extra synchronisation, the split forward call / backward call as the host sees it, and a cProfile of the step loop.
usage: python tools/prof_host.py [N ...]   (default 1000 500000 2000000)"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from sfgs.synth import scene, upstream_grads  # noqa: E402

dev = torch.device("cuda:0")
W, H = 1920, 1080


class _NullRaster(torch.autograd.Function):
    """An autograd op with the rasterizer's signature that launches NOTHING: what torch itself (dispatch, autograd graph, the
    engine's hand-over to its device thread and back, output allocation) costs per forward + backward (round 6: VERDICT r5
    item 4 asks how much of the small-scene step floor is this library's)."""

    @staticmethod
    def forward(ctx, means3D, means2D, colors, opac, scales, rots):
        ctx.save_for_backward(means3D, colors, opac, scales, rots)
        n = means3D.shape[0]
        outs = torch.empty(5, H, W, device=means3D.device)
        radii = torch.empty(n, dtype=torch.int32, device=means3D.device)
        ctx.mark_non_differentiable(radii)
        c, d, a = outs.split_with_sizes((3, 1, 1))
        return c, d, a, radii

    @staticmethod
    def backward(ctx, gc_, gd_, ga_, gr_):
        m, c, o, s_, r = ctx.saved_tensors
        return (torch.empty_like(m), torch.empty(m.shape[0], 3, device=m.device), torch.empty_like(c), torch.empty_like(o),
                torch.empty_like(s_), torch.empty_like(r))
gc, gd = (t.to(dev) for t in upstream_grads(W, H, 0))
for n in [int(a) for a in sys.argv[1:]] or [1000, 500000, 2000000]:
    frame, g = scene(n, W, H, seed=0)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"], kernel_size=frame["kernel_size"],
        subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0, viewmatrix=frame["view"].to(dev),
        projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros(n, 3, device=dev, requires_grad=True)
    params = list(t.values()) + [m2]
    tf = tb = 0.0

    def step(timed=False):
        global tf, tb
        for v in params:
            v.grad = None
        t0 = time.perf_counter()
        c, d, *_ = rast(means3D=t["means3D"], means2D=m2, colors_precomp=t["colors_precomp"], opacities=t["opacities"],
                        scales=t["scales"], rotations=t["rotations"])
        t1 = time.perf_counter()
        torch.autograd.backward([c, d], [gc, gd])
        if timed:
            tf += t1 - t0
            tb += time.perf_counter() - t1

    for _ in range(60):
        step()
    torch.cuda.synchronize()
    K = 300
    t0 = time.perf_counter()
    for _ in range(K):
        step(True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    def null_step():
        for v in params:
            v.grad = None
        c, d, _a, _r = _NullRaster.apply(t["means3D"], m2, t["colors_precomp"], t["opacities"], t["scales"], t["rotations"])
        torch.autograd.backward([c, d], [gc, gd])
    for _ in range(60):
        null_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        null_step()
    torch.cuda.synchronize()
    dn = (time.perf_counter() - t0) / K * 1e3
    print(f"N={n}: {dt:.4f} ms/step wall; host inside rast() {tf / K * 1e3:.4f} ms (includes the wait for the plan), "
          f"inside autograd.backward {tb / K * 1e3:.4f} ms; the same loop around an autograd op that launches NOTHING: "
          f"{dn:.4f} ms/step (torch's own floor)", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print("\n".join(ln for ln in s.getvalue().splitlines() if ln.strip())[:6000], flush=True)
