set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3x; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_raster.py tests/test_gpu_launch_hints.py tests/test_gpu_joint_render.py -m gpu -q -x --timeout=120 2>&1 | tail -6 ) | tee $O/raster.log
for r in 1 2 3; do for v in direct twopass; do b=""; [ $v = direct ] && b="direct"; SFGS_BINNING=$b timeout 100 python bench.py --forward-only --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v fwd-only', round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})"; done; done | tee $O/ab_tp.log
