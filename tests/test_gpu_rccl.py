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
