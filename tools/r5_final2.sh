#!/bin/bash
# round 5 (gpurun), after the fused_ssim rework: smoke(), pytest -m gpu, bench.py (driver-like), the training-iteration breakdown
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c25; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-200
timeout 1800 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-300
timeout 900 python bench.py > $O/bench_default.json 2>/dev/null; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5c25/bench_default.json") if l.startswith("{")][-1])
print("bench default:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"])
PY
timeout 1200 bash tools/train_iter_breakdown.sh r5_v2 > $O/train_iter.log 2>&1; tail -5 $O/train_iter.log | cut -c1-400
