#!/bin/bash
# round 5 (gpurun): rocprofv3 kernel trace + PMC passes + the 100-step trace of bench.py on the final tree (set r5_v8), box probe in front
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5_final_profiles; mkdir -p $O gpurun_out/meas_r5_v8
python - <<'PY' | tee gpurun_out/meas_r5_v8/box_probe.txt
import torch, time
x = torch.rand(64 * 1024 * 1024, device="cuda"); y = torch.rand_like(x)
for _ in range(5): z = x + y
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): z = x + y
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("torch add of 2 x 256 MB -> 256 MB:", round(dt * 1e3, 4), "ms =", round(3 * x.numel() * 4 / dt / 1e12, 3), "TB/s")
PY
timeout 400 python bench.py > gpurun_out/meas_r5_v8/bench_default.json 2>/dev/null; tail -1 gpurun_out/meas_r5_v8/bench_default.json | cut -c1-160
timeout 300 python tools/bench_train_iter.py > gpurun_out/meas_r5_v8/train_iteration.json 2>/dev/null; cat gpurun_out/meas_r5_v8/train_iteration.json
timeout 1500 tools/collect_profiles.sh r5_v8 > $O/prof.log 2>&1; tail -24 $O/prof.log | cut -c1-160
BENCH_EXTRA="--steps 100 --warmup 30" timeout 600 tools/collect_profiles.sh r5_v8_100steps "kt" > $O/prof100.log 2>&1; tail -10 $O/prof100.log | cut -c1-160
