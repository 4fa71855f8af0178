cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for l in main skyfall-gs_amd/sfgs/_exp/lib_prev.so; do
  if [ $l = main ]; then unset SFGS_LIB; else export SFGS_LIB=$PWD/$l; fi
  python tools/bench_regimes.py low_elevation_2M_1080p city_e25_2M_1080p near_big_splats_200k dense_8M_1080p jittered_2M_1080p uhd_2M_2160p tiny_scene_1k 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$l'[-12:], r['regime'][:24].ljust(24), r['ms_per_step'], 'preprocess', r['kernel_ms'].get('preprocess'))"
done
