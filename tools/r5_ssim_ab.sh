#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
for rnd in 1 2 3; do for l in main "$@"; do
  if [ $l = main ]; then unset SFGS_LIB; else export SFGS_LIB=$PWD/$l; fi
  echo "$l $(timeout 300 python tools/bench_aux.py 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print({k: v for k, v in d.items() if '2560' in k or ('1920' in k and 'only' in k)})")"
done; done | tee gpurun_out/ssim_ab.txt
