set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r3w; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -q --timeout=120 2>&1 | tail -8 ) > $O/all2.log 2>&1; tail -8 $O/all2.log
