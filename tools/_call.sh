cd "${GRAFT_REPO_ROOT:-.}"; R=gpurun_out/r3h; mkdir -p $R
timeout 300 python -m pytest tests/test_gpu_raster.py -x -q -k "fused_select_sort or forward_backward_parity" > $R/pytest_fused.log 2>&1; tail -15 $R/pytest_fused.log
p() { python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline_step']['kernel_ms_per_step']
print('$1', round(d['ms_per_step'], 4), ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"; }
for r in 1 2; do
python bench.py --cpu-sample 0 2>/dev/null | p fused
SFGS_SORT=split python bench.py --cpu-sample 0 2>/dev/null | p split
python bench.py --cpu-sample 0 --forward-only 2>/dev/null | p fused_fwd
SFGS_SORT=split python bench.py --cpu-sample 0 --forward-only 2>/dev/null | p split_fwd
done 2>&1 | tee $R/ab_fused.txt
