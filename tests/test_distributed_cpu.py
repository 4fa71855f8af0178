"""world_size-2 tests of the multi-GPU host logic over gloo (CPU): scene sharding, the shared-gradient
all-reduce and the band-sharded frame assembly. The collectives are the ones bench.py / a real
run issue over RCCL; only the backend differs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sfgs import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, dev = shard.init_distributed(backend="gloo")
    try:
        ret[rank] = fn(r, w)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return [ret[r] for r in range(world)]


def _bucket_job(rank, world):
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(59, 128), torch.nn.ReLU(), torch.nn.Linear(128, 128), torch.nn.ReLU(),
                              torch.nn.Linear(128, 6))  # scene/gaussian_model.py:52-58
    x = torch.full((4, 59), float(rank + 1))
    mlp(x).sum().backward()
    local = torch.cat([p.grad.reshape(-1) for p in mlp.parameters()]).clone()
    b = shard.SharedGradBucket(mlp.parameters())
    b.launch()
    b.wait()
    merged = torch.cat([p.grad.reshape(-1) for p in mlp.parameters()])
    return b.numel(), local.numpy(), merged.numpy()


def test_shared_grad_bucket_averages_over_ranks():
    out = _run(_bucket_job)
    assert out[0][0] == 24966  # the appearance MLP of the reference
    mean = (out[0][1] + out[1][1]) / 2
    for r in range(2):
        np.testing.assert_allclose(out[r][2], mean, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(out[0][2], out[1][2])  # ranks agree bit for bit


def _band_job(rank, world):
    H, W = 77, 40
    full = torch.arange(5 * H * W, dtype=torch.float32).reshape(5, H, W)
    t0, t1, a, b = shard.band_rows(H, world, rank)
    mine = torch.full_like(full, float("nan"))
    mine[:, a:b] = full[:, a:b]          # a band render leaves the other rows untouched
    got = shard.gather_bands(mine, H)
    return (t0, t1, a, b), bool(torch.equal(got, full))


def test_band_partition_and_gather():
    out = _run(_band_job)
    assert all(ok for _, ok in out)
    (t0, t1, a, b), (u0, u1, c, d) = out[0][0], out[1][0]
    assert t0 == 0 and t1 == u0 and u1 == (77 + 7) // 8 and a == 0 and b == c and d == 77


@pytest.mark.parametrize("height,world", [(1080, 8), (77, 3), (8, 4), (1440, 8)])
def test_band_rows_cover_image_exactly_once(height, world):
    rows = np.zeros(height, int)
    prev = 0
    for r in range(world):
        t0, t1, a, b = shard.band_rows(height, world, r)
        assert t0 == prev and a % 8 == 0
        prev = t1
        rows[a:b] += 1
    assert prev == (height + 7) // 8 and (rows == 1).all()


def test_scene_sharding_matches_one_process_per_scene():
    scenes = ["JAX_004", "JAX_068", "JAX_164", "JAX_168", "JAX_175", "JAX_214", "JAX_260", "JAX_264"]
    seen = []
    for r in range(8):
        mine = shard.scenes_for_rank(scenes, r, 8)
        assert len(mine) == 1
        seen += mine
    assert seen == scenes
    assert shard.scenes_for_rank(scenes, 1, 3) == ["JAX_068", "JAX_175", "JAX_264"]
