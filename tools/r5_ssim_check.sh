#!/bin/bash
# round 5 (gpurun): parity of the fused_ssim kernels after a change, A/B of variant libraries (bench_aux.py under SFGS_LIB),
# and the profile of tools/r5_ssim_prof.sh.   usage: tools/r5_ssim_check.sh [variant.so ...]
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -k "ssim or training_loop or reference_real or loss" -x 2>&1 | tail -5 | tee gpurun_out/ssim_tests.txt
for rnd in 1 2; do for l in main "$@"; do
  if [ $l = main ]; then unset SFGS_LIB; else export SFGS_LIB=$PWD/$l; fi
  echo "$l $(timeout 300 python tools/bench_aux.py 2>/dev/null | cut -c1-420)"
done; done | tee gpurun_out/ssim_ab.txt
unset SFGS_LIB
bash tools/r5_ssim_prof.sh
