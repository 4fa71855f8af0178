#!/bin/bash
# round 5 (gpurun): the round's final measurement set on the final tree, with a box-speed probe in front (boxes of the pool differ by
# several per cent, one in this round was ~4 % slow across every kernel incl. torch's own): set r5_v5
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c37; mkdir -p $O gpurun_out/meas_r5_v5
python - <<'PY' | tee gpurun_out/meas_r5_v5/box_probe.txt
import torch, time
x = torch.rand(64 * 1024 * 1024, device="cuda"); y = torch.rand_like(x)
for _ in range(5): z = x + y
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): z = x + y
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("torch add of 2 x 256 MB -> 256 MB:", round(dt * 1e3, 4), "ms =", round(3 * x.numel() * 4 / dt / 1e12, 3), "TB/s")
PY
timeout 1500 tools/measure_set.sh r5_v5 > $O/measure.log 2>&1; tail -45 $O/measure.log | cut -c1-300
timeout 1500 tools/collect_profiles.sh r5_v5 > $O/prof.log 2>&1; tail -30 $O/prof.log | cut -c1-200
BENCH_EXTRA="--steps 100 --warmup 30" timeout 600 tools/collect_profiles.sh r5_v5_100steps "kt" > $O/prof100.log 2>&1; tail -12 $O/prof100.log | cut -c1-200
timeout 1200 bash tools/train_iter_breakdown.sh r5_v5 > $O/train_iter.log 2>&1; tail -3 $O/train_iter.log | cut -c1-300
