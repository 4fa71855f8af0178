#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r5c13; timeout 600 python tools/diag_act_ulp.py > gpurun_out/r5c13/act_ulp.json 2>gpurun_out/r5c13/err.txt; cat gpurun_out/r5c13/act_ulp.json; tail -3 gpurun_out/r5c13/err.txt
