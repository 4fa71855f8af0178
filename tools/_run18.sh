set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3w; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_prepass_fold.py tests/test_prepass.py -m gpu -q -x --timeout=120 2>&1 | tail -25 ) > $O/fold.log 2>&1; tail -25 $O/fold.log
( timeout 600 python -m pytest tests -m gpu -q -x --timeout=120 --deselect tests/test_gpu_prepass_fold.py 2>&1 | tail -6 ) > $O/all.log 2>&1; tail -6 $O/all.log
for r in 1 2; do timeout 100 python bench.py --forward-only --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fwd-only', round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})"; done | tee $O/fwd.log
for f in 1 0 1 0; do SFGS_PREPASS_FOLD=$f ONLY=fused timeout 200 python tools/bench_train_iter.py 2>/dev/null | sed "s/^/fold=$f /"; done | tee $O/train_iter.log
timeout 100 python bench.py --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('step', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})" | tee $O/step.log
