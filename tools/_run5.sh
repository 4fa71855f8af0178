set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3f; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_launcher_fused.py -m gpu -q -x 2>&1 | tail -15 ) > $O/launcher.log 2>&1; tail -15 $O/launcher.log
bash tools/calibrate_sq.sh > $O/cal.log 2>&1; tail -5 $O/cal.log; head -70 gpurun_out/sq_cal/micro_summary.txt | tail -62
