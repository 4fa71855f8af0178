import os, sys, time, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer
from sfgs.synth import scene, upstream_grads
dev = torch.device("cuda:0")
for n in (1000, 100000):
    frame, g = scene(n, 1920, 1080, seed=0)
    gc, gd = (t.to(dev) for t in upstream_grads(1920, 1080, 0))
    settings = GaussianRasterizationSettings(image_height=1080, image_width=1920, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(dev), scale_modifier=1.0,
        viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=0, campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    t = {k: v.to(dev).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros(n, 3, device=dev, requires_grad=True)
    def fwd():
        return rast(means3D=t["means3D"], means2D=m2, colors_precomp=t["colors_precomp"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    for _ in range(5):
        c, d, *_ = fwd(); torch.autograd.backward([c, torch.nan_to_num(d)], [gc, gd])
    torch.cuda.synchronize()
    tf = tb = tn = 0
    for _ in range(20):
        for v in list(t.values()) + [m2]: v.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c, d, *_ = fwd(); torch.cuda.synchronize(); t1 = time.perf_counter()
        dn = torch.nan_to_num(d); torch.cuda.synchronize(); t2 = time.perf_counter()
        torch.autograd.backward([c, dn], [gc, gd]); torch.cuda.synchronize(); t3 = time.perf_counter()
        tf += t1 - t0; tn += t2 - t1; tb += t3 - t2
    print(n, "fwd ms", tf / 20 * 1e3, "nan_to_num ms", tn / 20 * 1e3, "bwd ms", tb / 20 * 1e3, flush=True)
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fwd()
        torch.cuda.synchronize(); print(n, "nograd fwd ms", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
