for r in 1 2 3; do for l in memset zk; do
  SFGS_LIB=$PWD/skyfall-gs_amd/sfgs/_exp/lib_$l.so timeout 200 python bench.py --cpu-sample 0 --forward-only 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('fwd-only $l', round(d['ms_per_step'], 4), round(d['value'],1))"
done; done
