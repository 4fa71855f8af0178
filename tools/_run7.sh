set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3h; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_launch_hints.py -m gpu -q -x 2>&1 | tail -30 ) > $O/hints.log 2>&1; tail -30 $O/hints.log
( timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/all.log 2>&1; tail -12 $O/all.log
for r in 1 2 3; do for h in 0 1; do
  SFGS_HINTS=$h timeout 200 python bench.py --cpu-sample 0 --steps 60 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline_step']['kernel_ms_per_step']
print('hints=$h', round(d['ms_per_step'], 4), 'busy', d['roofline_step']['gpu_busy_ms_per_step'], ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"
done; done 2>&1 | tee $O/ab_hints.log
for h in 0 1; do SFGS_HINTS=$h timeout 200 python bench.py --forward-only --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('hints=$h fwd-only', round(d['ms_per_step'],4), round(d['value'],1))"; done 2>&1 | tee -a $O/ab_hints.log
