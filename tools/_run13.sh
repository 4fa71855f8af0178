set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3n; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
( timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_densify.py tests/test_compact.py tests/test_gpu_fullsize_parity.py -m gpu -q -x 2>&1 | tail -4 ) > $O/t.log 2>&1; tail -4 $O/t.log
( bash tools/ab.sh $E/lib_head.so $E/lib_preload.so -- --steps 60 --warmup 20 ) > $O/ab.log 2>&1
cat $O/ab.log
for v in head preload; do SFGS_LIB=$PWD/$E/lib_$v.so timeout 200 python bench.py --forward-only --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v fwd-only', round(d['ms_per_step'],4), round(d['value'],1), {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})"; done | tee -a $O/ab.log
