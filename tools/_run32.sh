set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3x; mkdir -p $O
E=$PWD/skyfall-gs_amd/sfgs/_exp
for r in 1 2; do for v in g32m4 g32m8 g16m4 g16m8 g8m8; do SFGS_LIB=$E/lib_$v.so timeout 100 python bench.py --forward-only --cpu-sample 0 --steps 60 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k=d['roofline_step']['kernel_ms_per_step']; print('$v fwd-only', round(d['ms_per_step'],4), round(d['value'],1), 'count', round(k['bin_count'],4), 'rank', round(k['bin_rank'],4), 'scatter', round(k['bin_scatter'],4), 'pre', round(k['preprocess'],4))"; done; done | tee $O/ab_scatter_groups.log
