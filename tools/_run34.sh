set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3x; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_raster.py -m gpu -q --timeout=200 -k "uhd_two_bin_rounds or mid_splats_many_items" 2>&1 | tail -40 | cut -c1-600 ) | tee $O/raster4.log
