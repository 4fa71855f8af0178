set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3x; mkdir -p $O
( timeout 400 python -m pytest tests/test_gpu_raster.py tests/test_gpu_launch_hints.py tests/test_gpu_joint_render.py tests/test_gpu_fullsize_parity.py -m gpu -q -x --timeout=120 2>&1 | tail -4 ) | tee $O/raster3.log
for r in 1 2 3; do timeout 100 python bench.py --forward-only --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('matrix fwd-only', round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})"; done | tee $O/ab_matrix.log
