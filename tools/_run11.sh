set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3k; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
( bash tools/ab.sh $E/lib_k2.so $E/lib_k3.so $E/lib_k4.so -- --steps 60 --warmup 20 ) > $O/ab.log 2>&1
cat $O/ab.log
SFGS_LIB=$PWD/$E/lib_k3.so timeout 600 python -m pytest tests/test_gpu_raster.py -m gpu -q -x 2>&1 | tail -2
