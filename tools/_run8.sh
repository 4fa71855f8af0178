set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3i; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_raster.py -m gpu -q -x 2>&1 | tail -5 ) > $O/parity.log 2>&1; tail -5 $O/parity.log
timeout 900 python tools/bench_regimes.py near_big_splats_200k city_e25_2M_1080p city_e45_2M_1080p 2>&1 | grep '^{' | tee $O/regimes2.jsonl
