set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3e; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
( timeout 600 python -m pytest tests/test_gpu_raster.py -m gpu -q -x 2>&1 | tail -3 ) > $O/parity.log 2>&1; tail -2 $O/parity.log
( bash tools/ab.sh $E/lib_p2pipe.so $E/lib_p1pipe.so -- --steps 60 --warmup 20 ) > $O/ab.log 2>&1
cat $O/ab.log
