set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3x; mkdir -p $O
E=$PWD/skyfall-gs_amd/sfgs/_exp
for r in 1 2; do for v in tps a1 a3 a5 a13 g16 g64; do SFGS_LIB=$E/lib_$v.so timeout 100 python bench.py --forward-only --cpu-sample 0 --steps 60 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v fwd-only', round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})"; done; done | tee $O/ab_scatter_ablate.log
