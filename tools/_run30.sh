set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3x; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout=120 2>&1 | tail -8 ) > $O/all.log 2>&1; tail -8 $O/all.log
timeout 100 python bench.py --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})" | tee $O/step.log
