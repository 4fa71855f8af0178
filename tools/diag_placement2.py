#!/usr/bin/env python
"""Diagnostic 2: WHICH scratch blob makes preprocess slow where it lands? The plan stage is driven through the C ABI with
geom / tiles / bins as separate allocations; one of them at a time is moved to fresh device memory (old copies kept)."""
import json, os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "skyfall-gs_amd"))
import diff_gauss as dg
from sfgs import _lib as L
from sfgs.synth import scene
dev = torch.device("cuda:0")
N, W, H = 2_000_000, 1920, 1080
frame_d, g = scene(N, W, H, seed=0)
settings = dg.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=frame_d["tanfovx"], tanfovy=frame_d["tanfovy"],
    kernel_size=frame_d["kernel_size"], subpixel_offset=None, bg=frame_d["bg"].to(dev), scale_modifier=1.0,
    viewmatrix=frame_d["view"].to(dev), projmatrix=frame_d["proj"].to(dev), sh_degree=0, campos=frame_d["campos"].to(dev), prefiltered=False, debug=False)
t = {k: v.to(dev) for k, v in g.items() if v is not None}
lib = L.load()
keep = []
frame = dg._frame(settings, dev, 0, keep, 0, None)
gs = dg._gaussians(N, t["means3D"], t["scales"], t["rotations"], t["opacities"], t["colors_precomp"], None, None)
cap = int(lib.sfgs_raster_slot_capacity(W, H, 9_000_000)); ccap = 8192
sizes = L.SfgsRasterSizes(dg.C_sizeof(L.SfgsRasterSizes))
L.check(lib.sfgs_raster_sizes(N, W, H, cap, ccap, L.C.byref(sizes)))
radii = torch.empty(N, dtype=torch.int32, device=dev)
pin = torch.empty(16, dtype=torch.int64).pin_memory()
stream = dg._stream(dev)
def alloc(nbytes):
    return torch.empty(int(nbytes) + (3 << 20), dtype=torch.uint8, device=dev)   # odd sizes: never the cached block of another
blobs = dict(geom=alloc(sizes.geom_bytes), tiles=alloc(sizes.tiles_bytes), bins=alloc(sizes.bins_bytes))
print({k: v.numel() >> 10 for k, v in blobs.items()}, "KiB")

def measure(n=30):
    def go():
        L.check(lib.sfgs_raster_forward_plan(L.C.byref(frame), L.C.byref(gs), L.ptr(radii), L.ptr(blobs["geom"]), int(sizes.geom_bytes),
                                             L.ptr(blobs["tiles"]), int(sizes.tiles_bytes), L.ptr(blobs["bins"]), int(sizes.bins_bytes),
                                             cap, ccap, L.C.c_void_p(pin.data_ptr()), stream))
    for _ in range(5): go()
    torch.cuda.synchronize()
    L.profile_enable(True)
    for _ in range(n): go()
    torch.cuda.synchronize()
    p = L.profile_collect(); L.profile_enable(False)
    return round(p["preprocess"][0] / n, 4)

if len(sys.argv) > 1 and sys.argv[1] == "tiles":   # tiles blob only, many placements
    ts = [measure(20)]
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
        keep.append(blobs["tiles"])
        blobs["tiles"] = alloc(sizes.tiles_bytes + (i << 16))
        ts.append(measure(20))
    print(os.environ.get("SFGS_LIB", "default").split("/")[-1], "tiles placements:", ts, "min", min(ts), "max", max(ts))
    sys.exit(0)
print("baseline", measure())
for rnd in range(3):
    for which in ("geom", "tiles", "bins"):
        keep.append(blobs[which])
        blobs[which] = alloc(getattr(sizes, which + "_bytes"))
        print(f"round {rnd}: {which} moved ->", measure())
