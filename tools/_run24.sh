set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3w; mkdir -p $O
E=$PWD/skyfall-gs_amd/sfgs/_exp
( timeout 300 python -m pytest tests/test_gpu_raster.py tests/test_gpu_launch_hints.py -m gpu -q -x --timeout=90 2>&1 | tail -4 ) > $O/raster2.log 2>&1; tail -4 $O/raster2.log
for r in 1 2; do for v in cc32 eb cc64 eb64; do SFGS_LIB=$E/lib_$v.so timeout 200 python tools/diag_placement2.py tiles 14 2>&1 | tail -1; done; done | tee $O/placement4.log
