#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
ROUNDS=2 bash tools/ab.sh skyfall-gs_amd/sfgs/libsfgs.so skyfall-gs_amd/sfgs/_exp/lib_cb_trk.so skyfall-gs_amd/sfgs/_exp/lib_cb_nopost.so skyfall-gs_amd/sfgs/_exp/lib_fwd_trk.so | tee gpurun_out/flags_ab.txt
bash tools/r5_ssim_ab.sh skyfall-gs_amd/sfgs/_exp/lib_ssim_ilp.so
