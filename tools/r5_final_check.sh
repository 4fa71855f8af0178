#!/bin/bash
# round 5 (gpurun): what the driver runs at round end, on the final tree: smoke(), pytest -m gpu, python bench.py
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c9; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-300
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5c9/bench.json") if l.startswith("{")][-1])
print(d["metric"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_composite_pair"]["frac"], d["roofline_step"]["host_bound"], d["cpu_baseline"]["value"], d["roofline"]["traffic_source"]["file"])
PY
