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
    """Flat gradient bucket of the parameters that are shared between the per-rank scenes (the appearance
    MLP). all_reduce_() averages their .grad across ranks with a single collective.

    Scenes differ in length (iterations, early stops): the bucket carries one extra element, the number of ranks
    still training. A rank whose training has ended keeps answering the collective with zeros (`drain()`, called by
    the launcher at exit) until every rank has ended, and the training ranks average over the ACTIVE ranks only -- so
    a short scene never blocks or dilutes the long ones."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0] if self.params else torch.zeros(0)
        self.flat = torch.zeros(n + 1, dtype=ref.dtype, device=ref.device)   # [..., active-rank count]
        self._work = None
        self.steps = 0

    def numel(self):
        return self.flat.numel() - 1

    @staticmethod
    def _distributed():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def launch(self):
        """Pack the grads and start the all-reduce (async). Call right after backward()."""
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        self.flat[off] = 1.0
        if self._distributed():
            self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        return self

    def wait(self):
        """Finish the all-reduce and write the averaged grads back. Call before optimizer.step()."""
        if self._work is not None:
            self._work.wait()
            self._work = None
        n_total = self.flat.numel() - 1
        if self._distributed():
            self.flat[:n_total].div_(self.flat[n_total].clamp(min=1.0))   # mean over the ranks still training
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].reshape(p.shape).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].reshape(p.shape))
            off += n
        self.steps += 1

    def all_reduce_(self):
        self.launch().wait()

    def drain(self):
        """This rank's training is over: keep matching the other ranks' collectives with zeros until nobody trains.
        Returns the number of rounds answered."""
        if not self._distributed():
            return 0
        rounds = 0
        while True:
            self.flat.zero_()
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            rounds += 1
            if float(self.flat[-1]) == 0.0:       # every rank is draining: all leave in the same round
                return rounds


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


def render_joint(rasterizer_cls, settings, inputs, height):
    """Band-sharded joint render: every rank calls this with the same merged Gaussian set; returns the full
    (color, depth, alpha). `settings` is a diff_gauss.GaussianRasterizationSettings."""
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank() if world > 1 else 0
    t0, t1, _, _ = band_rows(height, world, rank)
    s = settings._replace(tile_rows=(t0, t1))
    with torch.no_grad():
        color, depth, _, alpha, _, _ = rasterizer_cls(raster_settings=s)(**inputs)
    packed = torch.cat([color, depth, alpha], 0)
    full = gather_bands(packed, height)
    return full[:3], full[3:4], full[4:5]
