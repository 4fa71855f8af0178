#!/bin/bash
# round 5 (gpurun): what the driver runs at round end, on the final tree: smoke(), pytest -m gpu, python bench.py; plus a soak
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5_final_check; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt | cut -c1-200
timeout 1800 python -m pytest tests -m gpu -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-300
timeout 1500 python tools/soak.py 3100 3300 > $O/soak.txt 2>&1; tail -4 $O/soak.txt | cut -c1-400
timeout 900 python bench.py --cpu-sample 0 --steps 20 --warmup 5 > $O/bench_driverlike.json 2>/dev/null; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5_final_check/bench_driverlike.json") if l.startswith("{")][-1])
print("steps 20 warmup 5:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_step"]["host_bound"]["gpu_gap_between_steps_ms"])
PY
timeout 600 python bench.py > $O/bench_default.json 2>/dev/null; tail -1 $O/bench_default.json | cut -c1-200
