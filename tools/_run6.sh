set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3g; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_joint_render.py -m gpu -q -x 2>&1 | tail -30 ) > $O/joint.log 2>&1; tail -30 $O/joint.log
