"""RCCL executes: torch.distributed backend "nccl" (= RCCL on ROCm) on DEVICE tensors of an MI355X.
A 1-GPU box can only form a group of one rank, which still goes through RCCL's communicator set-up and launches the
collective kernels on the device. Runs in a subprocess (a process group is process-global state).
  * sfgs.shard.SharedGradBucket (the launcher's appearance-MLP bucket, scene/gaussian_model.py:52-58: 24 966 floats):
    launch / wait around an all-reduce, sync_setup and drain rounds;
  * bench.py --gpus 1 --force-dist: the bench's per-step all-reduce with collective_backend "nccl" in the JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

JOB = r'''
import os, socket, sys
sys.path.insert(0, os.path.join({root!r}, "skyfall-gs_amd"))
import torch, torch.distributed as dist
from sfgs import shard
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(59, 128), torch.nn.ReLU(), torch.nn.Linear(128, 128), torch.nn.ReLU(),
                          torch.nn.Linear(128, 6)).to(dev)
opt = torch.optim.Adam(mlp.parameters(), lr=5e-4, eps=1e-15)
b = shard.SharedGradBucket(mlp.parameters(), opt, single_rank_collectives=True)
assert b.numel() == 24966 and b.flat.is_cuda
for k in range(3):
    mlp(torch.full((4, 59), float(k + 1), device=dev)).sum().backward()
    before = torch.cat([p.grad.reshape(-1) for p in mlp.parameters()]).clone()
    b.launch(); b.wait()                       # RCCL all-reduce of the flat bucket on the device
    after = torch.cat([p.grad.reshape(-1) for p in mlp.parameters()])
    assert torch.equal(before, after)          # one rank: the mean over the ranks that train is the rank's own gradient
    opt.step(); opt.zero_grad()
t = torch.arange(8, device=dev, dtype=torch.float32)
dist.all_reduce(t); dist.barrier()
assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
rounds = b.drain()                             # one passive round: nobody trains, nobody sets up -> leave
assert rounds == 1
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK steps", b.steps)
'''


def _env():
    return dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")


def test_shared_grad_bucket_all_reduces_over_rccl_on_the_device():
    r = subprocess.run([sys.executable, "-c", JOB.format(root=ROOT)], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK steps 3" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_force_dist_runs_the_collective_every_step():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--n", "200000",
                        "--width", "640", "--height", "360", "--steps", "5", "--warmup", "3", "--prewarm-steps", "2",
                        "--cpu-sample", "0"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["collective_backend"] == "nccl" and line["n_gpus"] == 1
    assert line["allreduce_ms_per_step"] is not None and line["allreduce_ms_per_step"] > 0


def test_bench_two_ranks_emit_the_line_the_scale_parser_reads():
    """`bench.py --gpus 2` end to end on the one GPU (ranks share it; collectives over gloo through SFGS_BENCH_BACKEND):
    the spawn, the barrier-bracketed timing, the per-rank gather and the ONE JSON line whose fields the driver's
    SCALE_rNN parser reads (BASELINE.json metric / unit, n_gpus, value = the two scenes' Gaussians over the slower
    rank's time, the rccl block with the world size the collective layer saw)."""
    env = dict(_env(), SFGS_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "200000", "--width", "640",
                        "--height", "360", "--steps", "5", "--warmup", "3", "--prewarm-steps", "2", "--settle-steps", "2",
                        "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines   # rank 0 only
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["warmup"] == 3 and line["scaling"] == "weak"
    assert line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["rccl"]["world_size"] == 2 and line["rccl"]["ranks_reporting"] == 2 and line["rccl"]["backend"] == "gloo"
    assert len(line["per_rank_ms_per_step"]) == 2 and len(line["per_rank_counts"]["rows"]) == 2
    # value = whole-job Gaussians per second: both ranks' scenes over the slower rank's step time
    assert abs(line["value"] - 2 * 200000 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"]
    assert abs(line["ms_per_step"] - max(line["per_rank_ms_per_step"])) < 1e-3
    # each rank rendered its own scene (seed = rank): the visible counts differ
    rows = line["per_rank_counts"]["rows"]
    assert rows[0] != rows[1] and all(v > 0 for v in rows[0] + rows[1])
    assert line["allreduce_ms_per_step"] is not None


def test_bench_eight_ranks_on_the_one_gpu():
    """N = 8 made boring before a node exists: `bench.py --gpus 8` -- the invocation the driver's
    scaling run makes -- with the eight ranks sharing this box's one GPU and the collectives over gloo. What is checked
    is everything but the interconnect: eight processes spawned, eight scenes (seed = rank), the all-reduce in every step
    paired across eight ranks, one gather of eight rows, ONE JSON line whose value is the eight scenes' Gaussians over the
    slowest rank's step."""
    env = dict(_env(), SFGS_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--n", "50000", "--width", "480",
                        "--height", "270", "--steps", "4", "--warmup", "2", "--prewarm-steps", "2", "--settle-steps", "2",
                        "--cpu-sample", "0"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["parallelism"] == "scene-per-gpu x8"
    assert line["rccl"]["world_size"] == 8 and line["rccl"]["ranks_reporting"] == 8 and line["rccl"]["backend"] == "gloo"
    assert len(line["per_rank_ms_per_step"]) == 8 and len(line["per_rank_counts"]["rows"]) == 8
    assert len({tuple(r_) for r_ in line["per_rank_counts"]["rows"]}) == 8          # eight different scenes
    assert abs(line["value"] - 8 * 50000 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"]
    assert abs(line["ms_per_step"] - max(line["per_rank_ms_per_step"])) < 1e-3
    assert line["allreduce_ms_per_step"] is not None
