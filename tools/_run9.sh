set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3j; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
for v in m0 m1 m2; do
  echo "== $v"
  SFGS_HINTS=0 SFGS_LIB=$PWD/$E/lib_$v.so timeout 900 python tools/bench_regimes.py near_big_splats_200k city_e25_2M_1080p low_elevation_2M_1080p dense_8M_1080p screen_filling_2k 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['regime'], d['ms_per_step'], 'sort_lds', d['kernel_ms'].get('sort_tiles_lds'), 'reg_long', d['kernel_ms'].get('sort_tiles_reg_long'))"
done 2>&1 | tee $O/modes.log
