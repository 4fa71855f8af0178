set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3w; mkdir -p $O
bash tools/ab.sh skyfall-gs_amd/sfgs/_exp/lib_base.so skyfall-gs_amd/sfgs/_exp/lib_hm.so -- --steps 100 --warmup 30 | tee $O/ab_hm.log
( timeout 300 python -m pytest tests/test_gpu_raster.py tests/test_gpu_fullsize_parity.py -m gpu -q -x --timeout=120 2>&1 | tail -3 ) | tee $O/raster3.log
