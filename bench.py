#!/usr/bin/env python
"""bench.py -- train-step Gaussians/s (forward + backward rasterization) on the BASELINE.json headline
workload: synthetic 2 M-Gaussian scene, 1920x1080 (SURVEY 8d cfg 2), float32, inputs resident in HBM.

One step = one forward + one backward of the rasterizer op through the drop-in boundary
(diff_gauss.GaussianRasterizer -> ctypes -> C ABI of libsfgs.so), exactly what
gaussian_renderer.render() + loss.backward() exercise. N > 1: one process per GPU
(torch.distributed, backend nccl == RCCL), one independent scene per rank (seed = rank; the path
shards by scene) plus the only exchange the sharded training has: an all-reduce of the shared
appearance-MLP gradient bucket (24 966 floats). Weak scaling: value = sum of Gaussians over ranks / max time.

Prints ONE JSON line (rank 0). `roofline` describes the dominant kernel (HIP events recorded by the
library on its launch stream during the timed region); `roofline_step` the whole step with SURVEY 8d's
algorithmic byte count. `cpu_baseline` times the CPU oracle (a port: the reference has no CPU path) on a
bounded sample. Only that leg touches oracle/.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "skyfall-gs_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
VALU_PEAK_TFLOPS = 157.3    # MI355X FP32 vector peak (MI355X_MICROARCH.md): 256 CUs x 128 lanes x 2 FLOP x 2.4 GHz
FP32_PEAK_TFLOPS = 157.3
APPEARANCE_MLP_FLOATS = 24966  # shared parameters all-reduced per step (scene/gaussian_model.py:52-58)


def kernel_bytes(N, Nvis, D, P):
    """SURVEY 8(d) algorithmic bytes split per kernel (sums to 128 N + 184 Nvis + 124 D + 64 P)."""
    return {
        "preprocess": 60 * N + 32 * Nvis,
        "scatter": 12 * D,
        "sort_tiles": 24 * D,
        "composite_fwd": 44 * D + 36 * P,
        "composite_bwd": 28 * P + 44 * D + 48 * Nvis,
        "preprocess_bwd": 104 * Nvis + 68 * N,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="Gaussians in the CPU-baseline sample (-1 = the whole workload, ~10-30 s; 0 = skip)")
    ap.add_argument("--forward-only", action="store_true", help="report render FPS instead of train-step Gaussians/s")
    ap.add_argument("--d2h-async", action="store_true", help="with --forward-only: download every frame through sfgs.video.FrameDownloader (pinned ring, side stream; informational)")
    ap.add_argument("--d2h-copy", action="store_true", help="with --forward-only: copy every frame to the host like render_video.py:181 (informational; never the headline value)")
    ap.add_argument("--sh-degree", type=int, default=-1, help=">= 0: colour path B (in-kernel SH of this degree) instead of colors_precomp")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from sfgs import _lib as L
    from sfgs.synth import scene, upstream_grads
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer, collect_full_counters, last_counters
    L.load()

    W, H, N = args.width, args.height, args.n
    sh = args.sh_degree
    frame, g = scene(N, W, H, seed=rank) if sh < 0 else scene(N, W, H, seed=rank, mode="sh", sh_degree=sh)
    gc, gd = upstream_grads(W, H, rank)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"], subpixel_offset=None, bg=frame["bg"].to(dev),
        scale_modifier=1.0, viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=max(sh, 0),
        campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    t = {k: (v.to(dev).requires_grad_(not args.forward_only) if v is not None else None) for k, v in g.items()}
    means2D = torch.zeros(N, 3, device=dev, requires_grad=not args.forward_only)
    gc, gd = gc.to(dev), gd.to(dev)
    shared_grad = torch.zeros(APPEARANCE_MLP_FLOATS, device=dev)

    downloader = None
    if args.d2h_async:
        from sfgs.video import FrameDownloader
        downloader = FrameDownloader(depth=3, device=dev)

    def step():
        if args.forward_only:
            with torch.no_grad():
                out = rast(means3D=t["means3D"], means2D=means2D, shs=t["shs"], colors_precomp=t["colors_precomp"],
                           opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
                if args.d2h_copy:
                    out[0].cpu()   # rendering.cpu() of the reference's video loop: a synchronous 24.9 MB PCIe copy at 1080p
                elif args.d2h_async:
                    downloader.submit(out[0])
            return
        for v in list(t.values()) + [means2D]:
            if v is not None:
                v.grad = None
        color, depth, _, _, _, _ = rast(means3D=t["means3D"], means2D=means2D, shs=t["shs"],
                                        colors_precomp=t["colors_precomp"], opacities=t["opacities"],
                                        scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([color, depth], [gc, gd])
        if dist is not None:
            dist.all_reduce(shared_grad)

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Warm-up with every kernel timed (HIP events on the launch stream): gives the per-kernel split and names the
    # dominant kernel. The timed region then brackets ONLY that kernel (each event pair costs the GPU ~4 us of
    # bubble; 9 kernels x 2 events per step would inflate the step by ~5 %).
    # one untimed frame with the diagnostic counter read (full stream sync) for max_tile_list; the timed steps use
    # the default mid-frame read
    collect_full_counters(True)
    step()
    max_tile_list = last_counters()["max_tile_list"]
    collect_full_counters(False)
    L.profile_select(None)
    n_prof = max(args.warmup - 1, 1) if args.warmup else 0   # the first step sizes the scratch (may re-plan): not timed
    for i in range(args.warmup):
        if i == args.warmup - n_prof:
            fence()
            L.profile_enable(True)
        step()
    fence()
    warm_prof = L.profile_collect() if args.warmup else {}
    L.profile_enable(True)

    def group(prof, nsteps):
        out = {}
        for name, (ms, launches) in prof.items():
            key = "sort_tiles" if name.startswith("sort_tiles") else name
            e = out.setdefault(key, {"ms_per_step": 0.0, "launches": 0})
            e["ms_per_step"] += ms / max(nsteps, 1)
            e["launches"] += launches
        return out
    warm = group(warm_prof, n_prof)
    dom = max(warm, key=lambda k: warm[k]["ms_per_step"]) if warm else None
    dom_names = [n for n in warm_prof if (n.startswith("sort_tiles") and dom == "sort_tiles") or n == dom]
    L.profile_select(dom_names if dom_names else None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = L.profile_collect()
    L.profile_enable(False)
    L.profile_select(None)

    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    cnt = last_counters()
    Nvis, D_ref, D_eff, P = cnt["num_visible"], cnt["num_duplicates_ref"], cnt["num_duplicates"], W * H
    ms_step = elapsed / args.steps * 1e3
    if args.forward_only:
        B_step = 60 * N + 32 * Nvis + 80 * D_ref + 36 * P
        metric = "render FPS (forward raster, %s)" % ("with the per-frame D2H copy of render_video.py:181" if args.d2h_copy else
                                                      "frames downloaded through a pinned ring on a side stream" if args.d2h_async else "no D2H copy")
        unit, value = "frames/s", world / (ms_step * 1e-3)
    else:
        B_step = 128 * N + 184 * Nvis + 124 * D_ref + 64 * P
        metric, unit = "train-step Gaussians/s (fwd+bwd raster) @1080p", "Gaussians/s"
        value = world * N / (ms_step * 1e-3)

    # per-kernel view (rank 0's launches): dominant kernel from the timed region, the split from the warm-up steps
    kb = kernel_bytes(N, Nvis, D_ref, P)
    per_kernel = dict(warm)
    timed = group(prof, args.steps)
    if dom in timed:
        per_kernel[dom] = timed[dom]
    # HBM traffic of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs,
    # gfx950 FETCH_SIZE x2 correction per MI355X_MICROARCH.md) of THIS workload, committed under profiles/ by
    # tools/collect_profiles.sh. bench.py cannot collect PMCs on itself; null when the workload differs.
    traffic = load_traffic(N, W, H, args.forward_only)
    roofline = None
    if dom is not None:
        dur = per_kernel[dom]["ms_per_step"] * 1e-3
        ach = kb.get(dom, 0) / dur / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic.get(dom),
                    "avg_launch_ms": round(per_kernel[dom]["ms_per_step"], 4),
                    "algorithmic_bytes_per_launch": int(kb.get(dom, 0))}
        if dom.startswith("composite"):
            # SURVEY 8(d)'s second line for the compositing kernels, which are VALU- not HBM-bound: every binned
            # (Gaussian, 8x8 tile) pair is evaluated on the wave's 64 pixel lanes; 22 FLOP per (pixel, splat)
            # evaluation is SURVEY's count for the forward blend (the backward does ~2.5x that, not credited)
            pairs = 64 * D_eff
            tf = pairs * 22 / dur / 1e12
            roofline["valu"] = {"pair_evaluations_per_launch": int(pairs), "flop_per_pair": 22,
                                "achieved": round(tf, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / VALU_PEAK_TFLOPS, 4)}
    step_ach = B_step / (ms_step * 1e-3) / 1e9
    roofline_step = {"bound": "hbm", "achieved": round(step_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(step_ach / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": int(B_step),
                     "kernel_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(per_kernel.items())},
                     "kernel_ms_source": "dominant kernel: timed region; others: warm-up steps",
                     "gpu_busy_ms_per_step": round(sum(v["ms_per_step"] for v in per_kernel.values()), 4)}

    cpu_baseline = None
    if rank == 0 and args.gpus == 1 and args.cpu_sample != 0 and not args.forward_only:
        cpu_baseline = run_cpu_baseline(N if args.cpu_sample < 0 else args.cpu_sample, W, H)

    if rank == 0:
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic cfg-2 scene, N={N} Gaussians/GPU, {W}x{H}, "
                                   f"{'colors_precomp' if sh < 0 else 'SH degree %d in-kernel' % sh}, "
                                   f"kernel_size=0.1, seed=rank, one scene per GPU",
                       "N": N, "width": W, "height": H, "N_vis": Nvis, "D_ref_16x16": D_ref, "D_binned_8x8": D_eff,
                       "max_tile_list": max_tile_list, "parallelism": f"scene-per-gpu x{world}"},
            "roofline": roofline, "roofline_step": roofline_step, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def load_traffic(N, W, H, forward_only):
    """{kernel: HBM bytes per launch} measured by tools/collect_profiles.sh for the default workload."""
    out = {}
    if (N, W, H) != (2_000_000, 1920, 1080):
        return out
    try:
        import glob
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
        for k, v in json.load(open(path)).items():
            key = k.replace("_kernel", "").split("<")[0]
            key = "sort_tiles" if key.startswith("sort_tiles") else key
            out[key] = out.get(key, 0) + int(v["hbm_bytes_per_launch"])
    except Exception:
        pass
    return out


def run_cpu_baseline(n, W, H):
    """The reference has no CPU render path; this times OUR C/OpenMP restatement (oracle/) of the same
    forward + backward on a bounded sample of the same workload."""
    try:
        from oracle import oracle as orc
        from sfgs.synth import scene, upstream_grads
        frame, g = scene(n, W, H, seed=0)
        gc, gd = upstream_grads(W, H, 0)
        cores = os.cpu_count() or 1
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        orc.lib()
        t0 = time.perf_counter()
        R = orc.OracleRender(frame, **g)
        gdm = gd.clone()
        gdm[torch.isnan(torch.from_numpy(R.depth))] = 0
        R.backward(gc, gdm)
        dt = time.perf_counter() - t0
        R.close()
        return {"value": n / dt, "unit": "Gaussians/s", "cores": cores, "kind": "port",
                "sample": f"one fwd+bwd of the C/OpenMP oracle on N={n} Gaussians of the same scene generator at "
                          f"{W}x{H} ({dt:.1f} s)"}
    except Exception as e:  # the baseline must never take the bench line down
        return {"value": None, "unit": "Gaussians/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}


if __name__ == "__main__":
    main()
