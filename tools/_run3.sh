set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3c; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
( timeout 600 python -m pytest tests/test_gpu_raster.py tests/test_gpu_fullsize_parity.py -m gpu -q -x 2>&1 | tail -3 ) > $O/parity.log 2>&1; tail -2 $O/parity.log
( bash tools/ab.sh $E/lib_base.so $E/lib_dup1.so -- --steps 60 --warmup 20 ) > $O/ab.log 2>&1
cat $O/ab.log
