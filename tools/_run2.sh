set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3b; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
for v in rec4 dg4 rec4dg4; do
  ( SFGS_LIB=$PWD/$E/lib_$v.so timeout 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_fullsize_parity.py -m gpu -q -x 2>&1 | tail -3 ) > $O/parity_$v.log 2>&1
  echo "== $v"; tail -2 $O/parity_$v.log
done
( bash tools/ab.sh $E/lib_base.so $E/lib_rec4.so $E/lib_dg4.so $E/lib_rec4dg4.so -- --steps 60 --warmup 20 ) > $O/ab.log 2>&1
cat $O/ab.log
for v in base rec4dg4; do SFGS_LIB=$PWD/$E/lib_$v.so timeout 200 python bench.py --forward-only --cpu-sample 0 --steps 100 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v fwd-only', round(d['ms_per_step'],4), d['value'], {k: round(v,4) for k,v in d['roofline_step']['kernel_ms_per_step'].items()})"; done
