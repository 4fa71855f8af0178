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
algorithmic byte count. `cpu_baseline` times the PyTorch CPU restatement of the render path (the
reference has no CPU path) on a bounded sample, next to the C/OpenMP oracle. Only that leg touches oracle/.
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


def pick_backend(world, device_count, env):
    """Collective backend of a multi-rank run. "nccl" (= RCCL on ROCm) always, unless SFGS_BENCH_BACKEND=gloo is set -- a TEST
    hook for boxes with fewer GPUs than ranks (ranks then share GPUs, collectives run on host tensors). It never happens silently
    and never where RCCL could have run: with a GPU per rank the hook is refused, so a driver-run
    `bench.py --gpus 8` on an 8-GPU node can only ever report rccl.backend == "nccl"."""
    backend = env.get("SFGS_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"bench.py: SFGS_BENCH_BACKEND={backend!r}: only 'nccl' (RCCL) or the test hook 'gloo'")
    if backend == "gloo" and world > 1 and device_count >= world:
        raise SystemExit(f"bench.py: SFGS_BENCH_BACKEND=gloo refused: {device_count} GPUs are visible for {world} ranks, so the "
                         "collectives must run over RCCL (backend 'nccl'); the gloo hook is for boxes with fewer GPUs than ranks")
    if backend == "gloo":
        print(f"bench.py: TEST HOOK: collectives over gloo on host tensors ({device_count} GPU(s) for {world} rank(s)); "
              "the JSON line says collective_backend=gloo", file=sys.stderr, flush=True)
    return backend


def box_probe(dev, L):
    """What THIS box sustains, measured in ~0.1 s before the timed region (boxes of the pool run the same
    binary 3-8 % apart, so a line from one round cannot be compared with another round's without it): a 256 MB + 256 MB ->
    256 MB torch add (HBM) and the library's fixed FP32 multiply-add loop (sfgs_box_probe: VALU rate and the shader clock it
    implies at one wave64 FMA per two cycles per SIMD)."""
    x = torch.rand(64 * 1024 * 1024, device=dev)
    y = torch.rand_like(x)
    z = torch.empty_like(x)
    for _ in range(3):
        torch.add(x, y, out=z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        torch.add(x, y, out=z)
    e1.record()
    torch.cuda.synchronize(dev)
    dt = e0.elapsed_time(e1) * 1e-3 / 20
    del x, y, z
    torch.cuda.empty_cache()
    tf, mhz = L.box_probe(None)
    return {"hbm_tbs": round(3 * 64 * 1024 * 1024 * 4 / dt / 1e12, 3), "valu_tflops": round(tf, 2), "sclk_mhz": round(mhz),
            "what": "torch add of 2 x 256 MB -> 256 MB (bytes moved / time); sfgs_box_probe: v_fma_f32 loop at 8 waves per SIMD, "
                    "sclk = the clock that rate implies at 2 cycles per wave64 FMA per SIMD; both before the timed region"}


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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--settle-steps", type=int, default=40,
                    help="untimed steps between the warm-up's bookkeeping (which idles the GPU for a few ms) and the timed "
                         "region: the timed steps start from the sustained clock state a training loop runs at")
    ap.add_argument("--prewarm-steps", type=int, default=50,
                    help="untimed steps BEFORE the W warm-up steps (the same count on every rank): the device needs ~30 ms of "
                         "sustained load to reach its sustained clock state (tools/diag_ramp.py: 1.35 -> 1.25 ms/step over "
                         "the first 25 steps of a process, again after 2 s of idle); 0 disables")
    ap.add_argument("--n", type=int, default=0)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="0 = skip the CPU-baseline leg; otherwise it runs on the workload's own N (a bounded tile sample, ~30-60 s)")
    ap.add_argument("--forward-only", action="store_true", help="report render FPS instead of train-step Gaussians/s")
    ap.add_argument("--d2h-async", action="store_true", help="with --forward-only: download every frame through sfgs.video.FrameDownloader (pinned ring, side stream; informational)")
    ap.add_argument("--d2h-copy", action="store_true", help="with --forward-only: copy every frame to the host like render_video.py:181 (informational; never the headline value)")
    ap.add_argument("--sh-degree", type=int, default=-1, help=">= 0: colour path B (in-kernel SH of this degree) instead of colors_precomp")
    ap.add_argument("--order", choices=["random", "morton"], default="random",
                    help="storage order of the Gaussians: 'random' (the headline: worst case for the id -> record gathers) "
                         "or 'morton' (sorted along a Z-curve of the ground position; informational)")
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4"], default="cfg2",
                    help="SURVEY 8d workloads with their exact geometry. cfg2 (the headline, BASELINE.json configs[1]): 2 M, "
                         "1920x1080, z ~ U(250, 350). cfg3 (configs[2], the IDU loop's rasterizer share): 108 forward-only "
                         "renders at 1024x1024 (train.py:360-525) followed by the timed fwd+bwd steps at 1024x1024, both rates "
                         "in the one JSON line. cfg4 (configs[3]): 5 M, 2560x1440, z ~ U(500, 700), dL/ddepth != 0. "
                         "--n / --width / --height override the config's size")
    ap.add_argument("--zrange", type=float, nargs=2, default=None, metavar=("ZMIN", "ZMAX"),
                    help="view depth range of the synthetic scene (default: the config's)")
    ap.add_argument("--subpixel-offset", choices=["none", "zeros"], default="zeros",
                    help="'zeros' (the headline since round 5): hand the rasterizer an all-zero [H,W,2] subpixel_offset tensor, "
                         "which is what the reference's render() allocates on every call when ray jitter is off "
                         "(gaussian_renderer/__init__.py:37-38) -- the shape the reference actually calls; 'none' (the "
                         "headline of rounds 1-4): no tensor")
    ap.add_argument("--cameras", type=int, default=8,
                    help="after the timed region (never inside it): K seeded cameras around the headline view, visited in turn "
                         "with a FRESH settings tuple per step whose viewmatrix is a transposed (non-contiguous) view and whose "
                         "subpixel_offset is a newly allocated zeros tensor -- what gaussian_renderer.render() builds on every "
                         "call (gaussian_renderer/__init__.py:37-55, scene/cameras.py:62); reported as `train_shaped`. 0 = skip")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: initialise torch.distributed over RCCL (backend nccl, world size 1) and run the "
                         "24 966-float device all-reduce every step, as the N > 1 runs do")
    ap.add_argument("--pin-cores", type=int, default=4,
                    help="sfgs.affinity.auto(cores=K), what tools/launch_scenes.py does per rank: the process is confined to K CPUs of "
                         "one L3 domain while the scene has fewer than 500 k Gaussians (host-bound sizes run 35-45 %% faster and "
                         "reproducibly that way) and left alone above (the 2 M headline is GPU-bound and is NEVER pinned: confinement "
                         "costs it 1-5 %%); reported as host{} in the JSON line. 0 = off")
    ap.add_argument("--pin-below", type=int, default=500_000, help="the size below which --pin-cores applies (released above 1.6 x this)")
    ap.add_argument("--cpu-leg", default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-only", action="store_true", help="run only the CPU-baseline legs and print them")
    argv = sys.argv[1:]
    if not argv and "RANK" in os.environ and os.environ.get("SFGS_BENCH_ARGV"):
        argv = json.loads(os.environ["SFGS_BENCH_ARGV"])   # a rank spawned by `python bench.py --gpus N ...` (see below)
    args = ap.parse_args(argv)
    CFG = {"cfg2": dict(n=2_000_000, width=1920, height=1080, zrange=(250.0, 350.0), configs_index=1),
           "cfg3": dict(n=2_000_000, width=1024, height=1024, zrange=(250.0, 350.0), configs_index=2),
           "cfg4": dict(n=5_000_000, width=2560, height=1440, zrange=(500.0, 700.0), configs_index=3)}[args.config]
    args.n, args.width, args.height = args.n or CFG["n"], args.width or CFG["width"], args.height or CFG["height"]
    zrange = tuple(args.zrange) if args.zrange else CFG["zrange"]
    if args.cpu_leg:
        return cpu_leg(args.cpu_leg, args.n, args.width, args.height)
    if args.cpu_only:
        print(json.dumps(run_cpu_baseline(args.n, args.width, args.height)))
        return

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU
        # (the driver's own multi-GPU invocation already comes through torchrun and skips this)
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # this process's arguments travel in the environment: torchrun's own parser rejects script options that are
        # prefixes of its options ("--n" is ambiguous between --nnodes, --nproc-per-node, ...) even behind the script name
        os.environ["SFGS_BENCH_ARGV"] = json.dumps(sys.argv[1:])
        os.execvp(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                   f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port",
                                   str(port), os.path.abspath(__file__)])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    from sfgs import affinity
    host = {"cpus_allowed": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    affinity.auto(local_rank=local_rank, cores=args.pin_cores, below=args.pin_below, above=int(args.pin_below * 1.6))   # (cores 0: off)
    # SFGS_BENCH_BACKEND=gloo: test hook that exercises the multi-rank control flow (spawn, barriers, per-rank gather,
    # JSON) on a box with fewer GPUs than ranks -- ranks then share GPUs and the collectives run on host tensors.
    backend = pick_backend(world, torch.cuda.device_count(), os.environ)   # "nccl" is RCCL on ROCm
    local_dev = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s_.getsockname()[1]))
            s_.close()
        kw = {"device_id": dev} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    from sfgs import _lib as L
    from sfgs.synth import scene, upstream_grads
    from diff_gauss import GaussianRasterizationSettings, GaussianRasterizer, collect_full_counters, last_counters
    L.load()

    W, H, N = args.width, args.height, args.n
    sh = args.sh_degree
    frame, g = (scene(N, W, H, seed=rank, zrange=zrange) if sh < 0 else
                scene(N, W, H, seed=rank, zrange=zrange, mode="sh", sh_degree=sh))
    if args.order == "morton":
        from sfgs.synth import morton_order
        perm = morton_order(g["means3D"])
        g = {k: (v[perm].contiguous() if v is not None else None) for k, v in g.items()}
    gc, gd = upstream_grads(W, H, rank)
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
        kernel_size=frame["kernel_size"],
        subpixel_offset=torch.zeros(H, W, 2, dtype=torch.float32, device=dev) if args.subpixel_offset == "zeros" else None,
        bg=frame["bg"].to(dev),
        scale_modifier=1.0, viewmatrix=frame["view"].to(dev), projmatrix=frame["proj"].to(dev), sh_degree=max(sh, 0),
        campos=frame["campos"].to(dev), prefiltered=False, debug=False)
    rast = GaussianRasterizer(settings)
    t = {k: (v.to(dev).requires_grad_(not args.forward_only) if v is not None else None) for k, v in g.items()}
    means2D = torch.zeros(N, 3, device=dev, requires_grad=not args.forward_only)
    gc, gd = gc.to(dev), gd.to(dev)
    shared_grad = torch.zeros(APPEARANCE_MLP_FLOATS, device=coll_dev)

    downloader = None
    if args.d2h_async:
        from sfgs.video import FrameDownloader
        downloader = FrameDownloader(depth=3, device=dev)

    def step(forward_only=args.forward_only):
        if forward_only:
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
            if ar_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(shared_grad)
                e1.record()
                ar_events.append((e0, e1))
            else:
                dist.all_reduce(shared_grad)

    ar_events = None   # (start, end) events around the all-reduce, filled during the profiled warm-up steps only

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
    box = box_probe(dev, L) if rank == 0 else None
    # device clock ramp (power management, not this code: the same ramp follows every idle period): a training loop runs
    # at the sustained state, so bring the device there before the W warm-up and K timed steps
    for _ in range(max(args.prewarm_steps, 0)):   # a fixed count: with N > 1 every step holds a collective
        step()
    torch.cuda.synchronize(dev)
    idu = None
    if args.config == "cfg3" and not args.forward_only:
        # the IDU episode's render phase (train.py:360-525: 108 pseudo-camera renders under no_grad), then the training steps
        n_idu = 108
        for _ in range(5):
            step(True)
        fence()
        ti = time.perf_counter()
        for _ in range(n_idu):
            step(True)
        fence()
        dt_idu = time.perf_counter() - ti
        idu = {"renders": n_idu, "ms_per_render": round(dt_idu / n_idu * 1e3, 4), "frames_per_s": round(n_idu / dt_idu, 1),
               "what": "108 forward-only 1024x1024 renders (no_grad, no D2H copy) before the timed fwd+bwd steps"}
        for _ in range(10):
            step()
    L.profile_select(None)
    n_prof = max(args.warmup - 1, 1) if args.warmup else 0   # the first step sizes the scratch (may re-plan): not timed
    for i in range(args.warmup):
        if i == args.warmup - n_prof:
            fence()
            L.profile_enable(True)
            ar_events = [] if dist is not None else None
        step()
    fence()
    allreduce_ms = None
    if ar_events:
        allreduce_ms = sum(a.elapsed_time(b) for a, b in ar_events) / len(ar_events)
    ar_events = None
    # every bracketed launch carries the cost of its own event pair: measure it (empty pairs on the same stream). The
    # per-kernel figures reported below are the RAW event times (they agree with rocprofv3's kernel trace to ~1 %, VERDICT
    # r4: the subtraction over-corrected by 3 %); only `gpu_busy_ms_per_step` takes the pairs' cost out, so that it stays
    # comparable with the un-bracketed step time
    cal = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for a_, b_ in cal:
        a_.record(); b_.record()
    torch.cuda.synchronize(dev)
    event_pair_ms = sorted(a_.elapsed_time(b_) for a_, b_ in cal)[len(cal) // 2]
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
    # The bookkeeping above (event calibration, reading the warm-up's events back) leaves the GPU idle for a few
    # milliseconds, and the device's power management answers idle time with a clock ramp that lasts ~30 steps
    # (tools/diag_transient.py: after 20 ms of idle the next steps take 1.23, 1.14, 1.09, 1.07 ms per ten, 1.065 without
    # the idle). A training loop never idles like that, so run a fixed number of untimed steps again and go from them
    # STRAIGHT into the bracketing barrier + synchronize and the K timed steps.
    L.profile_enable(False)
    for _ in range(max(args.settle_steps, 0)):
        step()
    L.profile_enable(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = L.profile_collect()
    L.profile_enable(False)
    L.profile_select(None)
    # Is the HOST ever the bottleneck? After the timed region (never inside it), SPAN_STEPS more steps with ONE event in front
    # of and one behind every step on the launch stream and no per-kernel bracket: `gpu_span` = first launch .. last kernel of
    # a step, `gpu_gap_between_steps` = end of a step .. start of the next. While the host runs ahead the gap is the cost of
    # the two event records; when the GPU waits for the host it is that wait. (The per-kernel brackets cannot answer this:
    # each carries a few us of event overhead, and their sum is not the step -- rocprofv3's kernel trace of the headline shows
    # the kernels back to back: 940.6 us of kernels in a 942.4 us step, profiles/r5_v1_rocprofv3_100steps.txt.)
    span = None
    if not args.forward_only:
        SPAN_STEPS = 20
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(SPAN_STEPS)]
        for a_, b_ in evs:
            a_.record()
            step()
            b_.record()
        torch.cuda.synchronize(dev)
        spans = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
        gaps = sorted(evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(SPAN_STEPS - 1))
        span = {"steps": SPAN_STEPS, "gpu_span_ms_per_step": round(spans[len(spans) // 2], 4),
                "gpu_gap_between_steps_ms": round(gaps[len(gaps) // 2], 4),
                "what": "median over untimed steps after the timed region: events in front of and behind every step on the launch "
                        "stream; a gap above the cost of two event records (~0.01 ms) means the GPU waited for the host"}

    train_shaped = None
    if args.cameras > 0 and not args.forward_only and dist is None:   # single-process runs only: its steps hold no collective
        train_shaped = run_train_shaped(args, dev, frame, t, means2D, gc, gd, W, H, max(sh, 0), GaussianRasterizationSettings,
                                        GaussianRasterizer, last_counters, step)
    per_rank_ms = [elapsed / args.steps * 1e3]
    cnt = last_counters()
    per_rank_counts = [[cnt["num_visible"], cnt["num_duplicates_ref"], cnt["num_duplicates"]]]
    rccl = None
    if dist is not None:
        # one gather: every rank's own clock and its scene's counters (the driver checks "RCCL saw N ranks" against these)
        mine = torch.tensor([elapsed, cnt["num_visible"], cnt["num_duplicates_ref"], cnt["num_duplicates"]],
                            device=coll_dev, dtype=torch.float64)
        allr = [torch.zeros(4, device=coll_dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [[float(x) for x in t_.tolist()] for t_ in allr]
        per_rank_ms = [r_[0] / args.steps * 1e3 for r_ in rows]
        per_rank_counts = [[int(x) for x in r_[1:]] for r_ in rows]
        elapsed = max(r_[0] for r_ in rows)
        ver = None
        if backend == "nccl":
            try:
                ver = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:
                ver = None
        rccl = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "version": ver,
                "ranks_reporting": len(rows),
                "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "MASTER_ADDR")}}
    Nvis, D_ref, D_eff, P = cnt["num_visible"], cnt["num_duplicates_ref"], cnt["num_duplicates"], W * H
    ms_step = elapsed / args.steps * 1e3
    if args.forward_only:
        B_step = 60 * N + 32 * Nvis + 80 * D_ref + 36 * P
        metric = "render FPS (forward raster, %s)" % ("with the per-frame D2H copy of render_video.py:181" if args.d2h_copy else
                                                      "frames downloaded through a pinned ring on a side stream" if args.d2h_async else "no D2H copy")
        unit, value = "frames/s", world / (ms_step * 1e-3)
    else:
        B_step = 128 * N + 184 * Nvis + 124 * D_ref + 64 * P
        metric = "train-step Gaussians/s (fwd+bwd raster) @%s" % ("1080p" if (W, H) == (1920, 1080) else f"{W}x{H}")
        unit = "Gaussians/s"
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
    traffic, traffic_source = load_traffic(N, W, H, args.forward_only)
    roofline = None
    if dom is not None:
        dur = per_kernel[dom]["ms_per_step"] * 1e-3
        ach = kb.get(dom, 0) / dur / 1e9
        roofline = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic.get(dom),
                    "traffic_source": traffic_source if traffic.get(dom) is not None else None,
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
    # north_star states its target on the forward + backward COMPOSITE pair: the two kernels' algorithmic bytes over the
    # sum of their launch durations (the forward's from the warm-up steps' bracketing, see kernel_ms_source)
    roofline_pair = None
    if not args.forward_only and "composite_fwd" in per_kernel and "composite_bwd" in per_kernel:
        pair_bytes = kb.get("composite_fwd", 0) + kb.get("composite_bwd", 0)
        pair_ms = per_kernel["composite_fwd"]["ms_per_step"] + per_kernel["composite_bwd"]["ms_per_step"]
        pa = pair_bytes / (pair_ms * 1e-3) / 1e9
        roofline_pair = {"kernels": ["composite_fwd", "composite_bwd"], "bound": "hbm", "achieved": round(pa, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(pa / HBM_PEAK_GBS, 5),
                         "algorithmic_bytes": int(pair_bytes), "ms": round(pair_ms, 4), "target_frac": 0.40}
    step_ach = B_step / (ms_step * 1e-3) / 1e9
    roofline_step = {"bound": "hbm", "achieved": round(step_ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(step_ach / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": int(B_step),
                     "kernel_ms_per_step": {k: round(v["ms_per_step"], 4) for k, v in sorted(per_kernel.items())},
                     "kernel_ms_source": "raw HIP-event brackets on the launch stream (dominant kernel: timed region; others: "
                                         "warm-up steps, where every kernel is bracketed)",
                     "event_pair_us": round(event_pair_ms * 1e3, 2),
                     "gpu_busy_ms_per_step": round(sum(max(v["ms_per_step"] - event_pair_ms * v["launches"] / max(
                         args.steps if k == dom else n_prof, 1), 0.0) for k, v in per_kernel.items()), 4),
                     "gpu_busy_source": "sum of the brackets minus the measured cost of an empty event pair per launch (a LOWER "
                                        "bound: the subtraction over-corrects; see host_bound for whether the GPU ever waits)",
                     "host_bound": span}
    if traffic:
        # the waste next to the contract's `frac`: real HBM traffic of the whole step from the committed PMC
        # passes (same file and provenance as roofline.traffic) against the algorithmic bytes
        # the kernels of a steady-state step (the split route's fine_bin / sort_tiles_reg run in the first, un-hinted frame only)
        names = ("preprocess", "bin_count", "bin_rank", "bin_scatter", "select_sort", "composite_fwd", "composite_bwd",
                 "preprocess_bwd", "subpix_bound", "zero_head")
        tot = sum(v for k, v in traffic.items() if k in names)
        if tot > 0 and not args.forward_only:
            roofline_step["traffic"] = int(tot)
            roofline_step["traffic_over_algorithmic"] = round(tot / B_step, 3)
            roofline_step["traffic_gbs"] = round(tot / (ms_step * 1e-3) / 1e9, 1)
            roofline_step["traffic_source"] = traffic_source

    host.update(affinity.state())   # what the timed region ran under
    affinity.auto(cores=0)
    affinity.unpin()   # (the CPU legs use every host core)
    cpu_baseline = None
    if rank == 0 and args.gpus == 1 and args.cpu_sample != 0 and not args.forward_only and args.config == "cfg2":
        cpu_baseline = run_cpu_baseline(N, W, H)

    if rank == 0:
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[{CFG['configs_index']}]: synthetic {args.config} scene (SURVEY 8d), N={N} Gaussians/GPU, "
                                   f"{W}x{H}, z ~ U({zrange[0]:g}, {zrange[1]:g}), "
                                   f"{'colors_precomp' if sh < 0 else 'SH degree %d in-kernel' % sh}, "
                                   f"kernel_size=0.1, subpixel_offset={'zeros[H,W,2]' if args.subpixel_offset == 'zeros' else 'None'}, "
                                   f"dL/dimage and dL/ddepth ~ N(0,1)/P, seed=rank, one scene per GPU",
                       "name": args.config, "zrange": list(zrange),
                       "N": N, "width": W, "height": H, "N_vis": Nvis, "D_ref_16x16": D_ref, "D_binned_8x8": D_eff,
                       "max_tile_list": max_tile_list, "parallelism": f"scene-per-gpu x{world}",
                       "collective_backend": backend if dist is not None else None,
                       "prewarm_steps": args.prewarm_steps, "settle_steps": args.settle_steps, "order": args.order},
            "subpixel_offset": args.subpixel_offset, "kernel_ms_brackets": "raw",   # not comparable with r1-r4 lines otherwise
            "box": box, "host": host,
            "train_shaped": train_shaped,
            "roofline": roofline, "roofline_composite_pair": roofline_pair, "roofline_step": roofline_step,
            "cpu_baseline": cpu_baseline,
            "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
            "per_rank_counts": {"columns": ["N_vis", "D_ref_16x16", "D_binned_8x8"], "rows": per_rank_counts},
            "rccl": rccl,
            "allreduce_ms_per_step": None if allreduce_ms is None else round(allreduce_ms, 4),
        }
        if idu is not None:
            out["idu_render_phase"] = idu
        # the JSON line is the LAST line of stdout: anything native libraries still hold in C stdio buffers (RCCL prints a
        # version banner at init when NCCL_DEBUG=VERSION is in the environment, as on the GPU boxes) goes out first
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def run_train_shaped(args, dev, frame, t, means2D, gc, gd, W, H, sh_degree, Settings, Rasterizer, last_counters, headline_step):
    """The headline scene the way a TRAINING loop presents it (VERDICT r5 item 4 / "weak" 9): K cameras in turn, and per step
    what gaussian_renderer.render() does before it calls the rasterizer -- a new zeros [H,W,2] subpixel_offset tensor
    (gaussian_renderer/__init__.py:37-38), a new GaussianRasterizationSettings tuple (:40-55) whose viewmatrix is the camera's
    world_view_transform, a TRANSPOSED view (scene/cameras.py:62: non-contiguous, so the wrapper copies it and its per-tuple
    frame cache can never hit), a new GaussianRasterizer module. The cameras are small rotations / shifts of the headline view
    (every one sees the whole scene, so the duplicate count stays within a few per cent and the line is comparable with the
    headline); the launch hints and capacities are relearnt from frame to frame as they are in training."""
    import math
    import numpy as np
    from sfgs.camera import fovy_from_fovx, make_frame
    K = args.cameras
    fovx = 2.0 * math.atan(frame["tanfovx"])
    fovy = fovy_from_fovx(fovx, W, H)
    rng = np.random.default_rng(2024)
    cams = []
    for k in range(K):
        yaw, pitch = (rng.uniform(-0.012, 0.012), rng.uniform(-0.008, 0.008)) if k else (0.0, 0.0)
        Ry = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
        Rx = np.array([[1, 0, 0], [0, math.cos(pitch), -math.sin(pitch)], [0, math.sin(pitch), math.cos(pitch)]])
        shift = rng.uniform(-2.0, 2.0, 3) if k else np.zeros(3)
        f = make_frame(Ry @ Rx, shift, fovx, fovy, W, H, kernel_size=frame["kernel_size"], sh_degree=sh_degree)
        # world_view_transform as the reference's Camera holds it: the transpose VIEW of the stored W2C matrix
        cams.append(dict(w2c=f["view"].t().contiguous().to(dev), proj=f["proj"].to(dev), campos=f["campos"].to(dev)))
    bg = frame["bg"].to(dev)
    attempts = []

    def tstep(i):
        c = cams[i % K]
        for v in list(t.values()) + [means2D]:
            if v is not None:
                v.grad = None
        settings = Settings(image_height=H, image_width=W, tanfovx=frame["tanfovx"], tanfovy=frame["tanfovy"],
                            kernel_size=frame["kernel_size"],
                            subpixel_offset=torch.zeros(H, W, 2, dtype=torch.float32, device=dev), bg=bg, scale_modifier=1.0,
                            viewmatrix=c["w2c"].transpose(0, 1), projmatrix=c["proj"], sh_degree=sh_degree, campos=c["campos"],
                            prefiltered=False, debug=False)
        assert not settings.viewmatrix.is_contiguous()
        color, depth, *_ = Rasterizer(settings)(means3D=t["means3D"], means2D=means2D, shs=t["shs"],
                                                colors_precomp=t["colors_precomp"], opacities=t["opacities"],
                                                scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([color, depth], [gc, gd])
        attempts.append(last_counters()["plan_attempts"])
    for i in range(3 * K):           # every camera seen: capacities / hints settled the way a training loop settles them
        tstep(i)
    torch.cuda.synchronize(dev)
    del attempts[:]
    steps = max(40, 5 * K)
    t0 = time.perf_counter()
    for i in range(steps):
        tstep(i)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    # same-moment reference: the headline step (one camera, one reused tuple) right after, same clocks
    t0 = time.perf_counter()
    for _ in range(steps):
        headline_step()
    torch.cuda.synchronize(dev)
    ms_head = (time.perf_counter() - t0) / steps * 1e3
    return {"cameras": K, "steps": steps, "ms_per_step": round(ms, 4), "plan_attempts": int(sum(attempts)),
            "plan_attempts_per_step": round(sum(attempts) / max(len(attempts), 1), 3),
            "headline_ms_per_step_same_moment": round(ms_head, 4),
            "what": "K cameras in turn; per step a fresh zeros subpixel_offset, a fresh settings tuple with a transposed-view "
                    "viewmatrix and a fresh rasterizer module (what render() builds per call); after the timed region"}


def load_traffic(N, W, H, forward_only):
    """({kernel: HBM bytes per launch}, provenance). bench.py cannot collect PMC counters on itself: the figures are READ
    from the newest committed profiles/r*_traffic.json (separate rocprofv3 --pmc passes of this command on another box,
    tools/collect_profiles.sh) and the JSON line says so: file, the commit the passes were taken at (the file's
    "_meta" entry), and whether raster_*.hip changed since. ({}, None) when the workload is not the profiled one."""
    out = {}
    if (N, W, H) != (2_000_000, 1920, 1080):
        return out, None
    src = None
    try:
        import glob
        import subprocess
        def round_key(p):
            b = os.path.basename(p)   # r<round>_v<pass>_traffic.json
            import re
            m = re.match(r"r(\d+)_v(\d+)", b)
            return (int(m.group(1)), int(m.group(2))) if m else (0, 0)
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), key=round_key)[-1]
        data = json.load(open(path))
        meta = data.pop("_meta", {}) if isinstance(data.get("_meta"), dict) else {}
        for k, v in data.items():
            key = k.replace("void sfgs::", "").replace("_kernel", "").split("<")[0].split("(")[0]
            key = "sort_tiles" if key.startswith("sort_tiles") else key
            out[key] = out.get(key, 0) + int(v["hbm_bytes_per_launch"])
        src = {"kind": "file, not measured in this run", "file": os.path.relpath(path, ROOT),
               "collected_at_commit": meta.get("commit"), "fetch_correction": meta.get("fetch_correction",
               "FETCH_SIZE x2 for every kernel (MI355X_MICROARCH.md; established for wide streaming reads)")}
        try:
            if meta.get("commit"):
                r = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", meta["commit"], "--",
                                    "skyfall-gs_amd/csrc/raster_fwd.hip", "skyfall-gs_amd/csrc/raster_bwd.hip",
                                    "skyfall-gs_amd/csrc/composite_bwd.hip"],
                                   capture_output=True, timeout=10)
                src["kernels_changed_since"] = {0: False, 1: True}.get(r.returncode)   # None: no git here (GPU box)
        except Exception:
            src["kernels_changed_since"] = None
    except Exception:
        pass
    return out, src


def run_cpu_baseline(n, W, H):
    """the reference has no CPU render path (gaussian_renderer/__init__.py:27,38 hard-code "cuda"), so "the
    reference's PyTorch CPU render path" is our PyTorch restatement of the same algorithm (oracle/tiled_torch.py, autograd
    backward), timed with torch.set_num_threads(all host cores): `value` is for THIS workload, from a bounded sample of
    its tiles with the extrapolation stated; cfg 1 (BASELINE.json configs[0]) runs in full. The C/OpenMP oracle's time
    (ours too, a "port") is reported alongside. Reported baselines, not targets. Every leg runs in its own process with a
    time limit, so a slow host can never take the bench line down."""
    import subprocess
    cores = os.cpu_count() or 1
    threads = int(os.environ.get("SFGS_CPU_THREADS", cores))
    out = {"value": None, "unit": "Gaussians/s", "cores": threads, "kind": "pytorch-restatement", "sample": None}

    def leg(name, limit):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="passive", HIP_VISIBLE_DEVICES="")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", name, "--n", str(n), "--width", str(W),
                                "--height", str(H)], env=env, capture_output=True, text=True, timeout=limit)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            return json.loads(lines[-1]) if lines else {"failed": (r.stderr or "no output")[-300:]}
        except subprocess.TimeoutExpired:
            return {"failed": f"did not finish within {limit} s on {threads} threads"}
    lim = int(os.environ.get("SFGS_CPU_LEG_LIMIT", "150"))
    r = leg("torch_sample", lim)
    if "value" in r:
        out.update(value=r["value"], sample=r["sample"])
    else:
        out["sample"] = "pytorch restatement on this workload: " + r.get("failed", "failed")
    out["cfg1_full"] = leg("torch_cfg1", lim)
    out["port"] = leg("c_oracle", lim)
    return out


def cpu_leg(name, n, W, H):
    """One CPU-baseline leg (own process, see run_cpu_baseline); prints one JSON line."""
    from sfgs.synth import cfg1, scene, upstream_grads
    threads = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    if name == "c_oracle":
        from oracle import oracle as orc
        frame, g = scene(n, W, H, seed=0)
        gc, gd = upstream_grads(W, H, 0)
        orc.lib()
        t0 = time.perf_counter()
        R = orc.OracleRender(frame, **g)
        gdm = gd.clone()
        gdm[torch.isnan(torch.from_numpy(R.depth))] = 0
        R.backward(gc, gdm)
        dt = time.perf_counter() - t0
        R.close()
        print(json.dumps({"value": n / dt, "unit": "Gaussians/s", "cores": threads, "kind": "port",
                          "sample": f"one fwd+bwd of the C/OpenMP oracle (oracle/sfgs_oracle.c) on the whole workload ({dt:.1f} s)"}))
        return
    from oracle import tiled_torch
    torch.set_num_threads(threads)

    def fwd_bwd(frame, g, gc, gd, subset):
        t = {k: v.clone().requires_grad_(True) for k, v in g.items() if v is not None}
        m2 = torch.zeros(g["means3D"].shape[0], 3, requires_grad=True)
        t0 = time.perf_counter()
        color, depth, _, _, st = tiled_torch.render_tiled(frame, t["means3D"], t["scales"], t["rotations"], t["opacities"],
                                                          colors_precomp=t["colors_precomp"], means2D=m2, tile_subset=subset)
        if st["tiles"]:
            torch.autograd.backward([color, torch.nan_to_num(depth)], [gc, torch.nan_to_num(gd)])
        return time.perf_counter() - t0, st
    if name == "torch_cfg1":      # cfg 1 in full (50 k Gaussians, 800x800)
        f1, g1 = cfg1()
        gc1, gd1 = upstream_grads(800, 800, 0)
        dt1, _ = fwd_bwd(f1, g1, gc1, gd1, None)
        print(json.dumps({"value": 50_000 / dt1, "unit": "Gaussians/s", "seconds": round(dt1, 2), "cores": threads,
                          "workload": "configs[0]: 50 000 Gaussians, 800x800, one fwd+bwd, all tiles"}))
        return
    # this workload: preprocess + binning of all Gaussians, compositing fwd+bwd on every 64th 16x16 tile
    frame, g = scene(n, W, H, seed=0)
    gc, gd = upstream_grads(W, H, 0)
    TX, TY = (W + 15) // 16, (H + 15) // 16
    sample = list(range(7, TX * TY, 64))
    dt_pre, _ = fwd_bwd(frame, g, gc, gd, [])
    dt_s, st = fwd_bwd(frame, g, gc, gd, sample)
    pairs_total = st["num_duplicates"] * 256
    scale = pairs_total / max(st["pair_evaluations"], 1)
    est = dt_pre + max(dt_s - dt_pre, 0.0) * scale
    print(json.dumps({"value": n / est, "sample": (
        f"PyTorch restatement (oracle/tiled_torch.py, float32, {threads} threads) on N={n} at {W}x{H}: preprocess + "
        f"binning of all Gaussians {dt_pre:.1f} s; compositing fwd+bwd on {len(sample)} of {TX * TY} tiles "
        f"({st['pair_evaluations']:.3g} of {pairs_total:.3g} (entry, pixel) pairs) {max(dt_s - dt_pre, 0):.1f} s, "
        f"EXTRAPOLATED by the pair count to {est:.0f} s per fwd+bwd")}))


if __name__ == "__main__":
    main()
