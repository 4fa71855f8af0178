#!/bin/bash
# round 5 (gpurun): the RAW-mode (hooks on) preprocess / preprocess_bwd with the raw opacity and filter words loaded once, with the
# other inputs -- parity tests of the RAW route, then an alternating A/B of the fused training iteration (SFGS_LIB = previous build)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -x -k "prepass or raw or handle or training_loop or reference_real or features or viewdirs or boundary" 2>&1 | tail -3 | tee gpurun_out/raw_tests.txt
for rnd in 1 2 3; do for l in main "$@"; do
  if [ $l = main ]; then unset SFGS_LIB; else export SFGS_LIB=$PWD/$l; fi
  echo "$l $(ONLY=fused timeout 300 python tools/bench_train_iter.py 2>/dev/null | tail -1)"
done; done | tee gpurun_out/raw_loads_ab.txt
