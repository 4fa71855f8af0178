set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_render_trace.py tests/test_gpu_rccl.py tests/test_gpu_knn_spatial.py tests/test_gpu_densify.py tests/test_gpu_densify_masks_fullsize.py -m gpu -q -x -s 2>&1 | tail -60 ) > $O/new_tests.log 2>&1
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/all_tests.log 2>&1
E=skyfall-gs_amd/sfgs/_exp
( bash tools/ab.sh $E/lib_base.so $E/lib_nozero.so $E/lib_nop1.so $E/lib_nop2.so $E/lib_nostore.so $E/lib_nogather.so $E/lib_sparse.so -- --steps 60 --warmup 20 ) > $O/ablate.log 2>&1
tail -5 $O/new_tests.log; tail -8 $O/all_tests.log; cat $O/ablate.log
