set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3m; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
( timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_densify.py tests/test_compact.py tests/test_gpu_fullsize_parity.py -m gpu -q -x 2>&1 | tail -4 ) > $O/t.log 2>&1; tail -4 $O/t.log
( bash tools/ab.sh $E/lib_unser.so $E/lib_prolog.so -- --steps 60 --warmup 20 ) > $O/ab.log 2>&1
cat $O/ab.log
