set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3l; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
( timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 ) > $O/all.log 2>&1; tail -6 $O/all.log
( bash tools/ab.sh $E/lib_k2.so $E/lib_unser.so -- --steps 60 --warmup 20 ) > $O/ab.log 2>&1
cat $O/ab.log
for v in k2 unser; do SFGS_LIB=$PWD/$E/lib_$v.so timeout 200 python bench.py --forward-only --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v fwd-only', round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})"; done | tee -a $O/ab.log
for v in k2 unser; do echo $v; SFGS_LIB=$PWD/$E/lib_$v.so timeout 300 python tools/bench_next_rows.py 2>/dev/null | tail -1; done | tee $O/next_rows.log
