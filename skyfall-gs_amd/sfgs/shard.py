"""Multi-GPU sharding of the rasterizer path: one process per GPU, torch.distributed over RCCL/xGMI.

The path shards two ways:

* TRAINING -- by scene. The reference itself runs one process per urban tile (scripts/run_jax.py:8,22,52-87:
  a thread pool that sets CUDA_VISIBLE_DEVICES per scene and shells out to train.py). Here rank r owns
  scene r: its GaussianModel, cameras and optimiser never leave its GPU, and the rasterizer needs NO
  collective. The only parameters shared between scenes are the appearance MLP's
  (scene/gaussian_model.py:52-58: 24 966 floats); their gradients are averaged with ONE flat all-reduce per
  step (`SharedGradBucket`). 100 KB is latency-bound on xGMI, so the bucket is a single message launched
  right after backward and waited on just before the optimiser step.

* JOINT RENDER of several fused scenes -- by screen band. Every rank holds the merged Gaussian set and
  composites the 8-pixel tile rows `band_rows(...)` assigns to it (SfgsFrame.tile_row_begin/end: binning and
  compositing are restricted to the band, so per-rank work is ~1/world); `gather_bands` assembles the frame
  with one all-gather of image rows. Exact: every pixel's complete sorted list lives on one rank.

On CPU (tests) the same code runs over the gloo backend.
"""
import os

import torch
import torch.distributed as dist

TILE = 8  # rows per tile of the rasterizer's binning


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment. Returns (rank, world, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    dev = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if use_gpu else "gloo")  # "nccl" is RCCL on ROCm
        kw = {"device_id": dev} if (use_gpu and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, dev


def scenes_for_rank(scenes, rank, world):
    """Round-robin scene -> rank map (rank r trains scenes r, r + world, ...), the layout the reference's
    GPU farm produces with one process per scene."""
    return [s for i, s in enumerate(scenes) if i % world == rank]


class SharedGradBucket:
    """Flat bucket of the parameters that are shared between the per-rank scenes (the appearance MLP) -- ONE logical
    model and ONE logical Adam state, replicated on every rank that trains.

    EVERY collective this class issues is the same one -- an all-reduce (sum) of the same flat buffer

        [ gradients (n) | parameters (n) | Adam exp_avg (n) | Adam exp_avg_sq (n) | Adam step | number of ranks that train |
          one flag per rank: "I am in training_setup" (world) ]

    so that ranks in different phases can never pair mismatched collectives: a rank inside
    optimizer.step() (`all_reduce_`), a rank (re)building its optimizer (`sync_setup`: train.py:95, restore() at
    scene/gaussian_model.py:163, every IDU episode at train.py:633, or the next scene of a rank that trains several)
    and a rank that has finished (`drain`) all answer one another's rounds:

      * a training rank contributes its gradients, its parameters and Adam moments AS OF THE START of the round and
        counts itself; it divides the gradient sum by the number of ranks that trained in that round -- a short scene
        never blocks or dilutes the long ones -- and steps;
      * a rank in setup contributes only its flag. It ADOPTS the training ranks' parameters and Adam state (their mean:
        they are identical by construction; exact for 1, 2, 4, 8 contributors) and then applies the round's averaged
        gradient with its own optimizer, i.e. performs the very step the training ranks perform: it leaves the round
        bit-identical to them, although training_setup had just handed it a fresh Adam. If nobody trains yet (the common
        start: all ranks set up together) the ranks of the round run one more all-reduce in which the LOWEST rank in
        setup supplies its parameters -- "start from rank 0's initialisation" -- which is consistent because a round
        without a training rank consists of setup and draining ranks only, and all of them see the same flags;
      * a draining rank contributes zeros until a round has neither a training rank nor a rank in setup.

    No host synchronisation on the training path (the divide happens on the device). A rank that is NOT issuing rounds
    (loading its next scene, minutes of IDU refinement) holds the other ranks' optimizer steps: that is the price of
    one shared model, and why the launcher leaves sharing off by default."""

    def __init__(self, params, optimizer=None, single_rank_collectives=False):
        # single_rank_collectives: issue the collectives even in a group of ONE rank (diagnostics / the RCCL smoke test)
        self._single = bool(single_rank_collectives)
        self.params = [p for p in params if p.requires_grad]
        self.optimizer = optimizer             # whose state[p] = {step, exp_avg, exp_avg_sq} is shared (torch.optim.Adam)
        n = self.n = sum(p.numel() for p in self.params)
        ref = self.params[0] if self.params else torch.zeros(0)
        self.world = dist.get_world_size() if self._distributed() else 1
        self.rank = dist.get_rank() if self._distributed() else 0
        self.flat = torch.zeros(4 * n + 2 + self.world, dtype=ref.dtype, device=ref.device)
        self._work = None
        self.steps = 0

    def numel(self):
        return self.n

    def _distributed(self):
        return (dist.is_available() and dist.is_initialized() and
                (dist.get_world_size() > 1 or getattr(self, "_single", False)))

    @staticmethod
    def _pack(dst, tensors):
        torch.cat([t.reshape(-1) for t in tensors], out=dst)

    def _state(self, p):
        st = self.optimizer.state.get(p) if self.optimizer is not None else None
        return st if st else None

    def launch(self):
        """Pack the gradients (and the shared state) and start the all-reduce (async). Call right after backward()."""
        n = self.n
        with torch.no_grad():
            self.flat[2 * n:].zero_()
            if all(p.grad is not None for p in self.params):
                self._pack(self.flat[:n], [p.grad for p in self.params])
            else:
                off = 0
                for p in self.params:
                    k = p.numel()
                    if p.grad is None:
                        self.flat[off:off + k].zero_()
                    else:
                        self.flat[off:off + k].copy_(p.grad.reshape(-1))
                    off += k
            self._pack(self.flat[n:2 * n], [p.data for p in self.params])
            states = [self._state(p) for p in self.params]
            if states and all(st is not None and "exp_avg" in st for st in states):
                self._pack(self.flat[2 * n:3 * n], [st["exp_avg"] for st in states])
                self._pack(self.flat[3 * n:4 * n], [st["exp_avg_sq"] for st in states])
                self.flat[4 * n] = float(states[0]["step"])
            self.flat[4 * n + 1] = 1.0
        if self._distributed():
            self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        return self

    def _unpack_grads(self, scale=None):
        n = self.n
        with torch.no_grad():
            if scale is None:
                self.flat[:n].div_(self.flat[4 * n + 1].clamp(min=1.0))   # mean over the ranks that trained this round
            off = 0
            for p in self.params:
                k = p.numel()
                if p.grad is None:
                    p.grad = self.flat[off:off + k].reshape(p.shape).clone()
                else:
                    p.grad.copy_(self.flat[off:off + k].reshape(p.shape))
                off += k

    def wait(self):
        """Finish the all-reduce and write the averaged gradients back. Call before optimizer.step()."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._distributed():
            self._unpack_grads()
        self.steps += 1

    def all_reduce_(self):
        self.launch().wait()

    def _passive_round(self, in_setup):
        """One round as a rank that does not train. Returns (training ranks, Adam step sum, per-rank setup flags)."""
        n = self.n
        self.flat.zero_()
        if in_setup:
            self.flat[4 * n + 2 + self.rank] = 1.0
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        tail = self.flat[4 * n:].tolist()
        return tail[1], tail[0], tail[2:]

    def _seed_round(self, flags, mine):
        """No rank trains: the lowest rank in setup supplies the parameters (all ranks of the round take part)."""
        n = self.n
        src = next(r for r, f in enumerate(flags) if f > 0)
        buf = self.flat[:n]
        buf.zero_()
        if mine and self.rank == src:
            with torch.no_grad():
                self._pack(buf, [p.data for p in self.params])
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return buf

    def _adopt(self, buf, into):
        off = 0
        with torch.no_grad():
            for p, t in zip(self.params, into):
                k = p.numel()
                t.copy_(buf[off:off + k].reshape(p.shape))
                off += k

    def sync_setup(self, step=None):
        """Call from training_setup (any number of times, at any point of the other ranks' training) with the freshly
        built optimizer's UNWRAPPED step function: the shared parameters and Adam state are replaced by the ones the
        training ranks hold and the round's step is performed -- or, when nobody trains yet, the parameters become those
        of the lowest rank that is setting up in the same round."""
        if not self._distributed():
            return
        n = self.n
        active, step_sum, flags = self._passive_round(in_setup=True)
        if active == 0:
            self._adopt(self._seed_round(flags, mine=True), [p.data for p in self.params])
            return
        inv = 1.0 / active
        self._adopt(self.flat[n:2 * n] * inv, [p.data for p in self.params])
        opt = self.optimizer
        if opt is None or step is None:
            return
        nstep = round(step_sum * inv)
        if nstep > 0:   # the training ranks have stepped before: take over their moments
            for p in self.params:
                st = opt.state[p]
                if "exp_avg" not in st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if torch.is_tensor(st["step"]):
                    st["step"].fill_(float(nstep))
                else:
                    st["step"] = float(nstep)
            self._adopt(self.flat[2 * n:3 * n] * inv, [opt.state[p]["exp_avg"] for p in self.params])
            self._adopt(self.flat[3 * n:4 * n] * inv, [opt.state[p]["exp_avg_sq"] for p in self.params])
        # the step the training ranks take in this round: only the shared parameters carry a gradient here
        mine = set(id(p) for p in self.params)
        stash = []
        for g in opt.param_groups:
            for p in g["params"]:
                if id(p) not in mine and p.grad is not None:
                    stash.append((p, p.grad))
                    p.grad = None
        self.flat[:n].mul_(inv)
        self._unpack_grads(scale=inv)
        step()
        for p in self.params:
            p.grad = None
        for p, g in stash:
            p.grad = g

    def drain(self):
        """This rank's training is over: keep answering the other ranks' rounds with zeros until nobody trains or
        sets up. Returns the number of rounds answered."""
        if not self._distributed():
            return 0
        rounds = 0
        while True:
            active, _, flags = self._passive_round(in_setup=False)
            rounds += 1
            if active == 0 and not any(f > 0 for f in flags):
                return rounds                     # every rank is draining: all leave in the same round
            if active == 0:
                self._seed_round(flags, mine=False)


def band_rows(height, world, rank):
    """(tile_row_begin, tile_row_end, pixel_row_begin, pixel_row_end) of rank's screen band. Tile rows are split
    as evenly as possible; bands are contiguous, disjoint and cover the image."""
    ty = (height + TILE - 1) // TILE
    q, r = divmod(ty, world)
    t0 = rank * q + min(rank, r)
    t1 = t0 + q + (1 if rank < r else 0)
    return t0, t1, min(t0 * TILE, height), min(t1 * TILE, height)


def gather_bands(image, height, rank=None, world=None):
    """image [C,H,W] of which only this rank's band rows are valid -> full [C,H,W] on every rank.
    Bands have (at most two) different heights, so rows are padded to the tallest band for one all_gather."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return image
    world = world or dist.get_world_size()
    rank = dist.get_rank() if rank is None else rank
    spans = [band_rows(height, world, r)[2:] for r in range(world)]
    hmax = max(b - a for a, b in spans)
    Cc, _, W = image.shape
    a, b = spans[rank]
    # RCCL moves device memory directly; a host-only backend (gloo: tests, or a node without xGMI) gets the band
    # staged through host memory and the assembled frame copied back
    staged = image.is_cuda and dist.get_backend() == "gloo"
    src = image.cpu() if staged else image
    mine = torch.zeros(Cc, hmax, W, dtype=src.dtype, device=src.device)
    mine[:, :b - a] = src[:, a:b]
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = torch.empty_like(src)
    for (a, b), part in zip(spans, parts):
        out[:, a:b] = part[:, :b - a]
    return out.to(image.device) if staged else out


def render_joint(rasterizer_cls, settings, inputs, height, shard_gaussians=False):
    """Band-sharded joint render; returns the full (color, depth, alpha) on every rank. `settings` is a
    diff_gauss.GaussianRasterizationSettings.
      shard_gaussians=False: every rank calls this with the SAME merged Gaussian set (the fused PLY replicated once at
        load) and projects / bins all of it for its band.
      shard_gaussians=True: `inputs` are THIS RANK'S Gaussians only (rank r holds scene r; the merged set is the
        concatenation in rank order). Every rank projects and coarse-bins its own Gaussians, the ranks all-gather the
        48-byte compositing records and the coarse items (sfgs_raster_plan_export), merge them (sfgs_raster_plan_merge) and
        bin / sort / composite their band. Same pixels, bit for bit. Per frame and rank this moves 48 B x N_total +
        16 B x (coarse bins x fullest bin) x world over the interconnect in exchange for 1/world of the projection work:
        worth it when the Gaussians are NOT replicated (each rank only has its own scene in memory)."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    t0, t1, _, _ = band_rows(height, world, rank)
    s = settings._replace(tile_rows=(t0, t1))
    with torch.no_grad():
        if shard_gaussians:
            color, depth, alpha = render_merged_parts(_gather_parts(plan_export(settings, inputs)), s)
        else:
            color, depth, _, alpha, _, _ = rasterizer_cls(raster_settings=s)(**inputs)
    packed = torch.cat([color, depth, alpha], 0)
    full = gather_bands(packed, height)
    return full[:3], full[3:4], full[4:5]


# ---- Gaussian-sharded joint render: plan locally, exchange, merge, render the band --------------------------------------
def plan_export(settings, inputs, export_capacity=None):
    """Plan THIS rank's Gaussians for the whole frame and export the plan (device tensors with fixed strides):
    dict(N, D, rec[N,12] f32, count[NCB] i32, items[NCB, C, 4] i32, C, max_coarse)."""
    import diff_gauss as dg
    from sfgs import _lib as L
    lib = L.load()
    means3D = dg._f32c(inputs["means3D"], "means3D")
    dev = means3D.device
    N = int(means3D.shape[0])
    scales = dg._f32c(inputs["scales"], "scales", (3,))
    rotations = dg._f32c(inputs["rotations"], "rotations", (4,))
    opacities = dg._f32c(inputs["opacities"], "opacities").reshape(N, 1)
    colors = dg._f32c(inputs.get("colors_precomp"), "colors_precomp", (3,))
    shs = dg._f32c(inputs.get("shs"), "shs")
    if (shs is None) == (colors is None):
        raise ValueError("Please provide exactly one of either SHs or precomputed colors!")
    H, W = int(settings.image_height), int(settings.image_width)
    full = settings._replace(tile_rows=None)      # the plan covers the whole frame: every band needs these items
    keep = []
    with torch.cuda.device(dev):
        frame = dg._frame(full, dev, 0 if shs is None else int(shs.shape[1]), keep)
        stream = dg._stream(dev)
        gs = L.SfgsGaussians(dg.C_sizeof(L.SfgsGaussians), N, L.ptr(means3D), L.ptr(scales), L.ptr(rotations),
                             L.ptr(opacities), L.ptr(colors), L.ptr(shs))
        sizes = L.SfgsRasterSizes(dg.C_sizeof(L.SfgsRasterSizes))
        L.check(lib.sfgs_raster_sizes(N, W, H, 0, 0, L.C.byref(sizes)))
        ncb = max(int(sizes.coarse_bins), 1)
        cap = int(lib.sfgs_raster_slot_capacity(W, H, 4 * N))
        # (coarse_capacity: room of a coarse bin's SLAB, i.e. for the huge splats' items only since ABI 14 -- every item with
        # one-pass binning)
        ccap = max(8 * N // ncb if (ncb > 65536 or dg._binning_direct()) else 0, 256)
        radii = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
        while True:
            L.check(lib.sfgs_raster_sizes(N, W, H, cap, ccap, L.C.byref(sizes)))
            geom = torch.empty(max(int(sizes.geom_bytes), 256), dtype=torch.uint8, device=dev)
            tiles = torch.empty(int(sizes.tiles_bytes), dtype=torch.uint8, device=dev)
            bins = torch.empty(max(int(sizes.bins_bytes), 256), dtype=torch.uint8, device=dev)
            L.check(lib.sfgs_raster_forward_plan(L.C.byref(frame), L.C.byref(gs), L.ptr(radii), L.ptr(geom), geom.numel(),
                                                 L.ptr(tiles), tiles.numel(), L.ptr(bins), bins.numel(), cap, ccap, None,
                                                 stream))
            cnt = L.SfgsRasterCounters()
            L.check(lib.sfgs_raster_read_counters(L.ptr(tiles), L.C.byref(cnt), stream))
            D, cmax = int(cnt.num_duplicates), int(cnt.max_coarse_bin)
            if not cnt.overflow and int(lib.sfgs_raster_slot_capacity(W, H, D)) <= cap and cmax <= ccap:
                break
            new_cap = max(cap, int(lib.sfgs_raster_slot_capacity(W, H, int(D * 1.25) + 1024)))
            new_ccap = max(ccap, int(cmax * 1.25) + 256)
            if (new_cap, new_ccap) == (cap, ccap):
                new_cap = cap * 2        # a duplicate-index pool ran over although the total fits: more room per pool
            cap, ccap = new_cap, new_ccap
        cmax = max(cmax, int(cnt.max_bin_items))      # the export carries ALL items of a bin (slab + bin-sorted run)
        if export_capacity is None:
            export_capacity = _agree_max(max(cmax, 1), dev)
        C = int(export_capacity)
        if C < cmax:
            raise ValueError(f"export_capacity {C} is below this rank's fullest coarse bin ({cmax})")
        rec = torch.empty(N, 12, dtype=torch.float32, device=dev)
        count = torch.empty(ncb, dtype=torch.int32, device=dev)
        items = torch.empty(ncb, C, 4, dtype=torch.int32, device=dev)
        L.check(lib.sfgs_raster_plan_export(L.C.byref(frame), N, L.ptr(geom), L.ptr(tiles), L.ptr(bins), cap, ccap, C,
                                            L.ptr(rec), L.ptr(count), L.ptr(items), stream))
    return dict(N=N, D=D, rec=rec, count=count, items=items, C=C, max_coarse=cmax)


def _agree_max(v, dev):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64, device=dev if dist.get_backend() != "gloo" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def _gather_parts(part):
    """All-gather the exported plans (rank order). RCCL moves device tensors directly; gloo stages through the host."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [part]
    world = dist.get_world_size()
    dev = part["rec"].device
    staged = dist.get_backend() == "gloo"
    meta = torch.tensor([part["N"], part["D"]], dtype=torch.int64, device="cpu" if staged else dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    Ns, Ds = [int(m[0]) for m in metas], [int(m[1]) for m in metas]
    nmax = max(max(Ns), 1)
    rec = torch.zeros(nmax, 12, dtype=torch.float32, device=dev)
    rec[:part["N"]] = part["rec"]
    out = []
    src = {k: (v.cpu() if staged else v) for k, v in dict(rec=rec, count=part["count"], items=part["items"]).items()}
    gathered = {}
    for k, v in src.items():
        bufs = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(bufs, v.contiguous())
        gathered[k] = [b.to(dev) for b in bufs] if staged else bufs
    for r in range(world):
        out.append(dict(N=Ns[r], D=Ds[r], rec=gathered["rec"][r][:Ns[r]].contiguous(), count=gathered["count"][r],
                        items=gathered["items"][r], C=part["C"]))
    return out


def render_merged_parts(parts, settings):
    """Merge exported plans (part order = Gaussian order of the concatenated set) and render `settings`' frame / band.
    Returns (color, depth, alpha)."""
    import diff_gauss as dg
    from sfgs import _lib as L
    lib = L.load()
    dev = parts[0]["count"].device
    H, W = int(settings.image_height), int(settings.image_width)
    C = int(parts[0]["C"])
    N = sum(p["N"] for p in parts)
    D = sum(p["D"] for p in parts)
    P = len(parts)
    keep = []
    with torch.cuda.device(dev):
        frame = dg._frame(settings._replace(sh_degree=0), dev, 0, keep)   # colours are already in the records
        stream = dg._stream(dev)
        sizes = L.SfgsRasterSizes(dg.C_sizeof(L.SfgsRasterSizes))
        cap = int(lib.sfgs_raster_slot_capacity(W, H, D))
        ccap = P * C
        L.check(lib.sfgs_raster_sizes(N, W, H, cap, ccap, L.C.byref(sizes)))
        geom = torch.empty(max(int(sizes.geom_bytes), 256), dtype=torch.uint8, device=dev)
        tiles = torch.empty(int(sizes.tiles_bytes), dtype=torch.uint8, device=dev)
        bins = torch.empty(max(int(sizes.bins_bytes), 256), dtype=torch.uint8, device=dev)
        arr = lambda ptrs: (L.C.c_void_p * P)(*ptrs)
        L.check(lib.sfgs_raster_plan_merge(L.C.byref(frame), P, (L.C.c_int32 * P)(*[p["N"] for p in parts]),
                                           arr([p["rec"].data_ptr() for p in parts]),
                                           arr([p["count"].data_ptr() for p in parts]),
                                           arr([p["items"].data_ptr() for p in parts]), C, L.ptr(geom), geom.numel(),
                                           L.ptr(tiles), tiles.numel(), L.ptr(bins), bins.numel(), cap, ccap, stream))
        outs = (torch.zeros if getattr(settings, "tile_rows", None) else torch.empty)(5, H, W, dtype=torch.float32, device=dev)
        L.check(lib.sfgs_raster_forward_render(L.C.byref(frame), N, L.ptr(geom), L.ptr(tiles), L.ptr(bins), bins.numel(), cap,
                                               ccap, -1, L.ptr(outs[0:3]), L.ptr(outs[3:4]), L.ptr(outs[4:5]), None, 0, stream))
        cnt = L.SfgsRasterCounters()
        L.check(lib.sfgs_raster_read_counters(L.ptr(tiles), L.C.byref(cnt), stream))
        if cnt.overflow:
            raise RuntimeError("merged plan overflowed its blobs (export_capacity below a rank's fullest coarse bin?)")
    return outs[0:3], outs[3:4], outs[4:5]
