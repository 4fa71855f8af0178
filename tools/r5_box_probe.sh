#!/bin/bash
# round 5 (gpurun): the headline line, the training iteration and a fixed torch kernel on whatever box this call gets -- boxes of this
# pool differ by several per cent (same binary), so a set of numbers is only comparable within one call
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; O=gpurun_out/box_probe; mkdir -p $O
python - <<'PY' | tee $O/torch_probe.txt
import torch, time
x = torch.rand(64 * 1024 * 1024, device="cuda"); y = torch.rand_like(x)
for _ in range(5): z = x + y
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): z = x + y
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("torch add of 2 x 256 MB -> 256 MB:", round(dt * 1e3, 4), "ms =", round(3 * x.numel() * 4 / dt / 1e12, 3), "TB/s")
PY
timeout 600 python bench.py > $O/bench_default.json 2>/dev/null; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/box_probe/bench_default.json") if l.startswith("{")][-1])
print("bench:", round(d["ms_per_step"], 4), f'{d["value"]:.4g}', d["roofline"]["frac"], d["roofline_step"]["kernel_ms_per_step"])
PY
timeout 300 python tools/bench_train_iter.py 2>/dev/null | tail -1 | tee $O/train_iteration.json
