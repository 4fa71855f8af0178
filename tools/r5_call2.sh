#!/bin/bash
# round 5, GPU call 2: the whole -m gpu suite on the cleaned sources (ABI 16) incl. the reference-real tests, then the bench lines
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 300 python bench.py --cpu-sample 0 > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json
timeout 300 python bench.py --cpu-sample 0 --n 500000 > $O/bench_500000.json 2>/dev/null
timeout 300 python bench.py --cpu-sample 0 --subpixel-offset none > $O/bench_subpix_none.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default","bench_500000","bench_subpix_none"):
    try:
        d=json.loads([l for l in open(f"gpurun_out/r5c2/{f}.json") if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"],4), d["roofline_step"]["gpu_busy_ms_per_step"], d["roofline_step"]["kernel_ms_per_step"], d["roofline"]["frac"])
    except Exception as e: print(f, "failed", e)
PY
