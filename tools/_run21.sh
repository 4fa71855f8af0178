set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3w; mkdir -p $O
for r in 1 2 3; do echo "== process $r"; timeout 200 python tools/diag_placement2.py 2>&1 | tail -11; done | tee $O/placement2.log
