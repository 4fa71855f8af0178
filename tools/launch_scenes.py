#!/usr/bin/env python
"""launch_scenes.py -- one training process per urban tile, one GPU each, on a single 8 x MI355X node.

Replaces the reference's GPU farm (scripts/run_jax.py:52-87: a thread pool that sets CUDA_VISIBLE_DEVICES per scene
and shells out to `python train.py ...`) WITHOUT editing any reference file:

  * every worker process gets HIP_VISIBLE_DEVICES=<its GPU>, so the reference's hard-coded device 0
    (utils/general_utils.py:133, train.py:60,451,499) is that GPU;
  * the worker puts this repo's drop-in packages (diff_gauss, fused_ssim, simple_knn) ahead of the reference on
    sys.path, installs the fused hooks on the reference's own GaussianModel class and runs the
    reference's train.py IN PROCESS (runpy) with the per-scene argument list;
  * by default the scenes train independently, exactly like the reference's farm (no collective at all);
  * with --shared-mlp the appearance MLP, 24 966 floats (scene/gaussian_model.py:52-58), is ONE model kept in step
    across the processes: GaussianModel.training_setup (scene/gaussian_model.py:350-382) is wrapped to adopt the MLP
    the training ranks hold (at the start: rank 0's initialisation) and to attach a sfgs.shard.SharedGradBucket, and
    optimizer.step() first averages the MLP gradients with ONE flat all-reduce over RCCL / xGMI (backend "nccl").
    Everything per-Gaussian stays local. Every collective is the same all-reduce of the same buffer, so repeated
    training_setup calls (restore, IDU episodes), several scenes per rank and scenes of different length cannot pair
    mismatched collectives (sfgs/shard.py).

usage:
  python tools/launch_scenes.py --reference /path/to/Skyfall-GS --gpus 8 \\
      --scenes JAX_004 JAX_068 JAX_164 JAX_168 JAX_175 JAX_214 JAX_260 JAX_264 -- \\
      train.py -s data/datasets_JAX/{scene} -m outputs/JAX/{scene} --eval --port {port} --kernel_size 0.1 \\
      --resolution 1 --sh_degree 1 --appearance_enabled ...          (the flags of scripts/run_jax.py:22)
`{scene}`, `{gpu}`, `{rank}` and `{port}` (6009 + rank, the reference's GUI port) are substituted per process.
"""
import argparse
import datetime
import os
import runpy
import socket
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "skyfall-gs_amd")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ---- hooks on the reference's GaussianModel ---------------------------------------------------------------------------
def install_hooks(gaussian_model_cls, fused=True, shared_mlp=False, zcurve_order=False):
    """Patch the reference's GaussianModel class in place (idempotent). fused: the HIP drop-ins of INTEGRATION.md
    (pre-pass, 3D filter, densification statistics, Adam, prune compaction, densify_and_prune). shared_mlp (opt-in: the
    reference trains every scene independently, scripts/run_jax.py): keep the appearance MLP in step across the
    processes of the torch.distributed group."""
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    if fused:
        from sfgs import adam, appearance, compact, densify, densify_stats, filter3d, max_radii, prepass
        # max_radii: train.py:314 without its syncs; appearance: the MLP's weight gradients as a split-K reduction (plain torch)
        for mod in (prepass, filter3d, densify_stats, adam, compact, max_radii, appearance):
            mod.install(gaussian_model_cls)
        densify.install(gaussian_model_cls, zcurve_order=zcurve_order)   # optionally keeps the rows in Z-curve order
    if shared_mlp and not getattr(gaussian_model_cls, "_sfgs_shared_mlp", False):
        import torch.distributed as dist
        from sfgs.shard import SharedGradBucket
        orig = gaussian_model_cls.training_setup

        def training_setup(self, *args, **kwargs):
            # The reference calls training_setup repeatedly (train.py:95, restore() at scene/gaussian_model.py:163, every
            # IDU episode at train.py:633) and a rank may train several scenes: every call is ONE round of the bucket's
            # single-collective protocol (sfgs/shard.py), which ranks inside optimizer.step() or draining answer.
            out = orig(self, *args, **kwargs)
            mlp = getattr(self, "appearance_mlp", None)
            if mlp is None or not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
                return out
            bucket = SharedGradBucket(list(mlp.parameters()), self.optimizer)
            step = self.optimizer.step
            bucket.sync_setup(step)                # adopt the MLP + Adam state the training ranks hold (or the lowest rank's)
            self._sfgs_shared_bucket = bucket
            _BUCKETS.append(bucket)

            def step_with_all_reduce(*a, **k):
                bucket.all_reduce_()               # average the MLP gradients over the ranks training in this round
                return step(*a, **k)
            self.optimizer.step = step_with_all_reduce
            return out
        training_setup.__wrapped__ = orig
        gaussian_model_cls.training_setup = training_setup
        gaussian_model_cls._sfgs_shared_mlp = True


_BUCKETS = []


def drain():
    """Call when this process has finished all its scenes: answers the other ranks' all-reduces until they finish."""
    return _BUCKETS[-1].drain() if _BUCKETS else _drain_without_bucket()


def _drain_without_bucket():
    # a rank that never reached training_setup (e.g. no scene assigned) still has to match the collectives
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return 0
    from sfgs.shard import SharedGradBucket
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    n = int(os.environ.get("SFGS_SHARED_FLOATS", "24966"))
    b = SharedGradBucket([torch.nn.Parameter(torch.zeros(n, device=dev))])
    return b.drain()


# ---- worker: one process, one GPU, its scenes one after the other ------------------------------------------------------
def worker(args, cmd):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if PKG not in sys.path:
        sys.path.insert(0, PKG)                  # diff_gauss / fused_ssim / simple_knn resolve to the HIP drop-ins
    if args.pin_cores > 0:
        # this rank's process on a few cores of ONE L3 domain (a domain per rank) while its scene is small -- scenes below
        # ~500 k Gaussians are host-bound and 35 - 45 % faster that way -- and released once it has grown (sfgs/affinity.py)
        from sfgs import affinity
        # (thresholds of a TRAINING iteration, ~100 torch launches around the rasterizer: pinned it is 16 - 19 % faster up to
        # 1 M Gaussians, 9 % at 2 M, even at 4 M: profiles/r6_cpu_affinity_small_scenes.txt section 6)
        affinity.auto(local_rank=rank, cores=args.pin_cores, below=args.pin_below, above=int(args.pin_below * 1.4))
        print(f"[launch_scenes rank {rank}] cpu policy: {affinity.state()['policy']}", flush=True)
    import torch
    import torch.distributed as dist
    sharing = world > 1 and args.shared_mlp     # independent scenes (the default) need no process group at all
    if sharing:
        # "nccl" is RCCL on ROCm; SFGS_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
        backend = os.environ.get("SFGS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
        # long timeout: a rank may sit in its IDU refinement (FlowEdit, minutes) while the others wait in the all-reduce
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(hours=12), **kw)
    ref = os.path.abspath(args.reference)
    os.chdir(ref)
    if ref not in sys.path:
        sys.path.insert(1, ref)
    if not args.no_plyfile_standin:
        try:
            import plyfile  # noqa: F401
        except ImportError:                       # the reference imports plyfile; sfgs.ply provides the same two classes
            import types
            from sfgs import ply as sply
            m = types.ModuleType("plyfile")
            m.PlyData, m.PlyElement = sply.PlyData, sply.PlyElement
            sys.modules["plyfile"] = m
    import importlib
    gm = importlib.import_module(args.model_module)
    install_hooks(getattr(gm, args.model_class), fused=not args.no_fused, shared_mlp=args.shared_mlp,
                  zcurve_order=args.zcurve_order)
    scenes = [s for i, s in enumerate(args.scenes) if i % world == rank]
    rc = 0
    try:
        for scene in scenes:
            sub = dict(scene=scene, gpu=os.environ.get("HIP_VISIBLE_DEVICES", "0"), rank=rank, port=6009 + rank)
            argv = [a.format(**sub) for a in cmd]
            print(f"[launch_scenes rank {rank}] {scene}: {' '.join(argv)}", flush=True)
            sys.argv = argv
            runpy.run_path(argv[0], run_name="__main__")
    except SystemExit as e:
        rc = int(e.code or 0)
    finally:
        if sharing:
            rounds = drain()
            print(f"[launch_scenes rank {rank}] finished; answered {rounds} further all-reduce rounds", flush=True)
            dist.destroy_process_group()
    return rc


def main():
    argv = sys.argv[1:]
    cmd = []
    if "--" in argv:
        i = argv.index("--")
        argv, cmd = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="root of the (unmodified) reference checkout; train.py runs from here")
    ap.add_argument("--scenes", nargs="+", required=True)
    ap.add_argument("--gpus", type=int, default=0, help="processes / GPUs to use (default: one per scene, at most the visible GPUs)")
    ap.add_argument("--gpu-ids", default="", help="comma-separated physical GPU ids (default 0..gpus-1)")
    ap.add_argument("--no-fused", action="store_true", help="do not install the fused HIP hooks on GaussianModel")
    ap.add_argument("--shared-mlp", action="store_true",
                    help="share ONE appearance MLP between the scenes (all-reduce of its 24 966 gradients per step over RCCL). "
                         "Default: independent scenes, as the reference trains them. With sharing, a rank that is not "
                         "stepping (loading its next scene, IDU refinement) holds the other ranks' optimizer steps")
    ap.add_argument("--zcurve-order", action="store_true",
                    help="re-sort the Gaussians along a Z-curve after every densify_and_prune (a relabelling; faster binning)")
    ap.add_argument("--no-plyfile-standin", action="store_true")
    ap.add_argument("--pin-below", type=int, default=2_500_000,
                    help="Gaussian count below which --pin-cores applies (released again above 1.4 x this)")
    ap.add_argument("--pin-cores", type=int, default=4,
                    help="CPUs of one L3 domain each rank's process is confined to WHILE ITS SCENE IS SMALL (sfgs.affinity.auto; 0 = leave the affinity alone)")
    ap.add_argument("--model-module", default="scene.gaussian_model")
    ap.add_argument("--model-class", default="GaussianModel")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args(argv)
    if not cmd:
        ap.error("give the per-scene command after `--`, e.g. -- train.py -s data/{scene} -m out/{scene}")
    if args.worker:
        sys.exit(worker(args, cmd))
    ids = [g for g in args.gpu_ids.split(",") if g]
    world = args.gpus or len(ids)
    if not world:
        try:
            import torch
            world = torch.cuda.device_count() or 1
        except Exception:
            world = 1
    world = max(1, min(world, len(args.scenes)))
    ids = ids or [str(i) for i in range(world)]
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HIP_VISIBLE_DEVICES=ids[rank], HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("CUDA_VISIBLE_DEVICES", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"] + argv + ["--"] + cmd, env=env))
    rcs = [p.wait() for p in procs]
    print("[launch_scenes] exit codes:", rcs, flush=True)
    sys.exit(max(abs(r) for r in rcs))


if __name__ == "__main__":
    main()
